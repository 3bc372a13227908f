import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
T = 256
p = device_problem(1600, 1600, 285, deg=3, seed=0)
plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0)
cubes = [p.cube[i * 100:(i + 1) * 100, j * 100:(j + 1) * 100].contiguous() for i in range(16) for j in range(16)]
reals = [p.real[i * 100:(i + 1) * 100, j * 100:(j + 1) * 100].contiguous() for i in range(16) for j in range(16)]
plan.step_batch(cubes, reals)
tb = next(iter(plan._batches.values()))
for _ in range(30):
    eng.batch_reduce_solve(tb, 50)
torch.cuda.synchronize()
