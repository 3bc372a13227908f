"""K1+K2 launch time as a function of time since the first launch of the process (clock ramp?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "0.5")))       # let the GPU go idle first
N = 1200
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
t0 = torch.cuda.Event(enable_timing=True); t0.record()
for e in evs:
    plan.step(prob.cube, prob.real, k1_events=e)
torch.cuda.synchronize()
ts = [a.elapsed_time(b) for a, b in evs]
at = [t0.elapsed_time(a) for a, _ in evs]
for lo in (0, 5, 10, 20, 40, 80, 160, 320, 640, 1000):
    hi = min(N, lo * 2 if lo else 5)
    seg = sorted(ts[lo:hi])
    print("launches %4d..%4d (t = %6.1f ms): median K1 %.4f ms" % (lo, hi, at[lo], seg[len(seg) // 2]), flush=True)
