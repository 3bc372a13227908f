#!/usr/bin/env python3
"""The two speeds of K1 (0.195 / 0.216 ms at 1024 x 1024 x 285) are a property of WHERE THE CUBE LIES, not of a clock
state: one process, one stream, eight allocations of the same cube kept alive together -> the fused K1+K2 launch is fast
on some and slow on others, stable per allocation, while a pure LDS-DMA read of the very same buffers (hsr_probe_read
mode 0) runs at 6.7-6.9 TB/s on all of them.  tools/state_probe.py shows that the stream / hardware queue and an idle gap
do not matter.  (Replaces round 1's placement_probe / placement_variants / realloc_probe* / ramp_probe.)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion, _engine as eng, _native as nat
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0)
lib = nat.load()
sink = torch.zeros(64, device="cuda")
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    ts=[]
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
table = plan.table
st = C.c_void_p(0)
cubes = [prob.cube] + [prob.cube.clone() for _ in range(7)]
out = torch.empty((1024*1024, 12), device="cuda")
for i, c in enumerate(cubes):
    nbytes = c.numel()*4
    pr0 = timed(lambda: lib.hsr_probe_read(C.c_void_p(c.data_ptr()), nbytes, 0, C.c_void_p(sink.data_ptr()), st))
    pr1 = timed(lambda: lib.hsr_probe_read(C.c_void_p(c.data_ptr()), nbytes, 1, C.c_void_p(sink.data_ptr()), st))
    k1 = timed(lambda: eng.srf_integrate(c, table, out=out, layout="pixmajor"))
    ev=[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ts=[]
    for _ in range(15):
        plan.step(c, prob.real, k1_events=ev); ev[1].synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    ts.sort()
    print(f"cube {i} @ {c.data_ptr():#x}: LDS-DMA probe {pr0:.4f} ms ({nbytes/pr0/1e9:.2f} TB/s)  plain probe {pr1:.4f}  K1 {k1:.4f}  K1+K2 {ts[len(ts)//2]:.4f}")
