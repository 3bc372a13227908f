import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0)
table = eng.build_srf_table(prob.emit_w, prob.srf, prob.good_mask)
npix = 1024 * 1024
def k1(cube, out, n=9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for it in range(n + 2):
        e0.record(); eng.srf_integrate(cube, table, out=out, layout="pixmajor"); e1.record(); e1.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
cubes = [prob.cube] + [prob.cube.clone() for _ in range(3)]
outs = [torch.empty((npix, 12), device="cuda") for _ in range(10)]
spacer = []
print("out alloc   " + " ".join(f"{i:>6d}" for i in range(len(outs))))
for ci, c in enumerate(cubes):
    print(f"cube {ci} us   " + " ".join(f"{k1(c, o)*1000:6.1f}" for o in outs), flush=True)
# the weights table and the code are the only other operands: a second table object (new device copy of the weights)
t2 = eng.build_srf_table(prob.emit_w, prob.srf, prob.good_mask)
def k1t(cube, out, tb, n=9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for it in range(n + 2):
        e0.record(); eng.srf_integrate(cube, tb, out=out, layout="pixmajor"); e1.record(); e1.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
print("second weights copy: " + " ".join(f"{k1t(c, outs[0], t2)*1000:6.1f}" for c in cubes))
