"""predict_cube of the degree-3, 10-input ridge model over 1024 x 1024 pixels for a range of target counts (random model)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
import s2_emit
rng = np.random.default_rng(0)
H = W = 1024
X = torch.rand((10, H, W), device="cuda") * 0.5 + 0.1
for T in (16, 32, 33, 48, 64, 65, 96, 97, 192, 285):
    m = s2_emit.PolyRidge.from_params(np.full(10, 0.35), np.full(10, 0.15), rng.normal(0, 0.05, (T, 285)), rng.normal(0, 0.1, T))
    for _ in range(3): out = m.predict_cube(X)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): out = m.predict_cube(X)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"T={T:4d}: {ms:.3f} ms per Mpixel, {2 * 286 * T * H * W / ms / 1e9:.1f} TFLOP/s, checksum {float(out.float().sum()):.6e}", flush=True)
