"""Per-process K1 speed under different ways of allocating the cube (statistics over several processes)."""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
mode = os.environ.get("PLACEMENT", "0")
pre = None
if mode == "2":      # one big allocation first, the cube is copied into it at the end
    pre = torch.empty((1024, 1024, 285), dtype=torch.float32, device=dev)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
cube = prob.cube
if mode == "1":      # re-allocate after everything else was freed
    host = cube.cpu()
    prob.cube = None
    del cube
    gc.collect(); torch.cuda.empty_cache()
    cube = host.to(dev)
elif mode == "2":
    pre.copy_(cube)
    prob.cube = None
    del cube
    gc.collect(); torch.cuda.empty_cache()
    cube = pre
elif mode == "3":    # clone while the original is alive, then drop the original
    c2 = cube.clone()
    prob.cube = None
    del cube
    cube = c2
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
for _ in range(150):
    plan.step(cube, prob.real)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for e in evs:
    plan.step(cube, prob.real, k1_events=e)
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs)
from s2_emit import _native as nat
from s2_emit._engine import _ptr, _stream
lib = nat.load()
sink = torch.zeros(64, dtype=torch.float32, device=dev)
nbytes = cube.numel() * 4
for _ in range(3):
    nat.check(lib.hsr_probe_read(_ptr(cube), nbytes, _ptr(sink), _stream(torch)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    nat.check(lib.hsr_probe_read(_ptr(cube), nbytes, _ptr(sink), _stream(torch)))
e1.record(); e1.synchronize()
print("mode %s: cube %x  K1 %.4f ms   plain read of the cube %.0f GB/s" % (mode, cube.data_ptr(), ts[20], nbytes * 10 / e0.elapsed_time(e1) / 1e6), flush=True)
