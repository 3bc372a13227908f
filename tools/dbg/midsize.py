"""step() on mid-size tiles (where the slot rule changes the number of workgroups of a single-tile launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for H in (100, 160, 200, 256, 300, 362, 512):
    p = device_problem(H, H, 285, deg=3, seed=0)
    plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0)
    for _ in range(50): plan.step(p.cube, p.real)
    ts, ks = [], []
    for _ in range(5):
        e0.record()
        for _ in range(50): plan.step(p.cube, p.real)
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1000)
    for _ in range(20):
        plan.step(p.cube, p.real, k1_events=(e0, e1)); e1.synchronize(); ks.append(e0.elapsed_time(e1) * 1000)
    ks.sort()
    print(os.path.basename(os.environ.get("HSR_LIBRARY", "prod")), f"{H}x{H}: slots {plan.ws.slots}  step {min(ts):.1f} us  K1 median {ks[10]:.1f} us")
