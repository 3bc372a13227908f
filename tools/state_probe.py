#!/usr/bin/env python3
"""Which knob moves K1 between its two speeds (0.2015 / 0.2185 ms alternate between fresh processes on one box)?
One process: the same fused K1+K2 launch timed on different torch streams (= hardware queues), on several cube
allocations (= physical placements), and before / after a device-wide idle gap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem

torch.cuda.set_device(0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0)


def k1_ms(cube, real, stream=None, n=30):
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ts = []
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5):
            plan.step(cube, real)
        for _ in range(n):
            plan.step(cube, real, k1_events=ev)
            ev[1].synchronize()
            ts.append(ev[0].elapsed_time(ev[1]))
    ts.sort()
    return ts[len(ts) // 2]


print("default stream        ", round(k1_ms(prob.cube, prob.real), 4))
streams = {"prio 0 stream A": torch.cuda.Stream(priority=0), "prio 0 stream B": torch.cuda.Stream(priority=0),
           "high prio stream": torch.cuda.Stream(priority=-1), "another high prio": torch.cuda.Stream(priority=-1)}
for rnd in range(2):
    for name, st in streams.items():
        print(f"{name:22s}", round(k1_ms(prob.cube, prob.real, st), 4))
    print("default stream        ", round(k1_ms(prob.cube, prob.real), 4))
# physical placement: five more cube allocations kept alive together
cubes = [prob.cube] + [prob.cube.clone() for _ in range(5)]
for i, c in enumerate(cubes):
    print(f"cube allocation {i} @ {c.data_ptr():#x}", round(k1_ms(c, prob.real), 4))
time.sleep(2.0)
print("after 2 s idle        ", round(k1_ms(prob.cube, prob.real), 4))
