"""Back-to-back time of the slot reduction + solve (single tile, 512 slots) and of the batched form (256 tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
p = device_problem(512, 512, 285, deg=3, seed=0)
plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0)
plan.step(p.cube, p.real)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1000)
    return " ".join(f"{t:.1f}" for t in ts)
print("slots", plan.ws.slots, "reduce_solve us:", timeit(lambda: eng.moments_reduce_solve(plan.ws, 50)))
print("reduce only us:", timeit(lambda: eng.moments_reduce(plan.ws)))
T = 256
g = torch.Generator(device="cuda"); g.manual_seed(1)
cubes = torch.rand((T, 100, 100, 285), generator=g, device="cuda") * 0.6
reals = torch.rand((T, 100, 100, 12), generator=g, device="cuda")
out = plan.step_batch(cubes, reals)
tb = next(iter(plan._batches.values()))
print("batched reduce_solve (256 tiles) us:", timeit(lambda: eng.batch_reduce_solve(tb, 50)))
