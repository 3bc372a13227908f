import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion, _engine as eng, _native as nat
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0)
plans = {r: SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, reserved_cus=r) for r in (0, 8, 32, 64)}
cubes = [prob.cube] + [prob.cube.clone() for _ in range(7)]
def k1(plan, c, n=15, real=None):
    real = prob.real if real is None else real
    ev=[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ts=[]
    for _ in range(3): plan.step(c, real)
    for _ in range(n):
        plan.step(c, real, k1_events=ev); ev[1].synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    ts.sort(); return ts[len(ts)//2]
u16 = None
for i, c in enumerate(cubes):
    row = [f"{k1(plans[r], c):.4f}" for r in plans]
    # a shifted view of the same allocation: drop the first 16 pixels (18240 bytes) -> tiles start elsewhere
    sh = c.reshape(-1, 285)[16:16 + 1024 * 1023]
    rs = prob.real.reshape(-1, 12)[16:16 + 1024 * 1023]
    row.append(f"shift16px {k1(plans[0], sh, 8, rs):.4f}")
    sh = c.reshape(-1, 285)[64 * 7:64 * 7 + 1024 * 1023]
    rs = prob.real.reshape(-1, 12)[64 * 7:64 * 7 + 1024 * 1023]
    row.append(f"shift7tiles {k1(plans[0], sh, 8, rs):.4f}")
    print(f"cube {i}: reserved 0/8/32/64: {' '.join(row)}", flush=True)
