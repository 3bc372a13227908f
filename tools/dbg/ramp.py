"""Is a 20-step timed region slower per step than a 100-step one because of a ramp at its start?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
p = device_problem(1024, 1024, 285, deg=3, seed=0)
plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3)
cube, real, log = plan.place_inputs(p.cube, p.real)
print("placement", log)
def region(n, idle_ms=0.0):
    torch.cuda.synchronize()
    if idle_ms: time.sleep(idle_ms * 1e-3)
    t0 = time.perf_counter()
    for _ in range(n): plan.step(cube, real)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n
for _ in range(5): plan.step(cube, real)
for n in (20, 20, 20, 100, 100, 20, 500, 20):
    print(n, "steps:", f"{region(n):.4f} ms/step")
for idle in (1, 10, 100, 1000):
    print("after", idle, "ms idle, 20 steps:", f"{region(20, idle):.4f}", " then 20 more:", f"{region(20):.4f}")
# per-step timeline inside one region
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
torch.cuda.synchronize(); time.sleep(0.05)
ev[0].record()
for i in range(40):
    plan.step(cube, real); ev[i + 1].record()
torch.cuda.synchronize()
print("per-step ms after 50 ms idle:", " ".join(f"{ev[i].elapsed_time(ev[i+1]):.3f}" for i in range(40)))
