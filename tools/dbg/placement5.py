import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0)
cubes = [prob.cube] + [prob.cube.clone() for _ in range(7)]
def k1(c, real, n=9):
    ev=[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ts=[]
    for _ in range(2): plan.step(c, real)
    for _ in range(n):
        plan.step(c, real, k1_events=ev); ev[1].synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    ts.sort(); return ts[len(ts)//2]
rows = [1024, 1023, 1016, 1008, 1000, 992, 960, 896, 768, 512]
print("rows      " + " ".join(f"{r:>6d}" for r in rows) + "   (us per 1024 rows equivalent)")
for i, c in enumerate(cubes):
    out = []
    for r in rows:
        n = 1024 * r
        t = k1(c.reshape(-1, 285)[:n], prob.real.reshape(-1, 12)[:n])
        out.append(t * 1024 / r)
    print(f"cube {i}    " + " ".join(f"{v*1000:6.1f}" for v in out), flush=True)
