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
cubes = [prob.cube] + [prob.cube.clone() for _ in range(5)]
N = 1024 * 1000
def k1(c, real, n=9):
    ev=[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ts=[]
    for _ in range(2): plan.step(c, real)
    for _ in range(n):
        plan.step(c, real, k1_events=ev); ev[1].synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    ts.sort(); return ts[len(ts)//2]
shifts = [0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 128, 192, 256, 512, 1024, 4096]
print("shift(px) " + " ".join(f"{s:>6d}" for s in shifts))
for i, c in enumerate(cubes):
    row = []
    for s in shifts:
        sh = c.reshape(-1, 285)[s:s + N]
        rs = prob.real.reshape(-1, 12)[s:s + N]
        row.append(k1(sh, rs))
    print(f"cube {i}    " + " ".join(f"{v*1000:6.1f}" for v in row), flush=True)
