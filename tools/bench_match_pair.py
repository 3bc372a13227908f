"""The reference driver (poly_regression.py:96-172) at full tile size on device-resident inputs:
EMIT 1024 x 1024 x 285 float32 cube, Sentinel-2 visual RGB 6144 x 6144 x 3 uint8 (factor 6)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
import s2_emit
from s2_emit.synthetic import device_problem

H = W = int(os.environ.get("HW", "1024"))
prob = device_problem(H, W, 285, deg=3, seed=0, device=torch.device("cuda", 0))
rgb = prob.real.reshape(H, W, -1)[..., [2, 1, 0]].clamp(0, 1)            # some RGB-like planes at 60 m
s2 = (rgb.repeat_interleave(6, 0).repeat_interleave(6, 1) ** 0.8 * 255 + 4 * torch.randn(H * 6, W * 6, 3, device="cuda")).clamp(0, 255).to(torch.uint8)
for use_ot in (False, True):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = s2_emit.match_pair(prob.cube, prob.emit_w, prob.srf, prob.good_mask, s2, 6, deg=4 if use_ot else 3, use_ot=use_ot, as_numpy=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"match_pair {H}x{W} EMIT / {H*6}x{W*6} S2, use_ot={use_ot}: {dt*1e3:.2f} ms", flush=True)
