"""Does the virtual address of the cube predict K1's speed?  N clones (no spacers between most), address and time each."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
p = device_problem(1024, 1024, 285, deg=3, seed=0)
plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0)
npix = 1024 * 1024
out = eng.alloc_image(torch, plan.table.nb, npix, plan.layout, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(c):
    rr, rl = plan._real_image(p.real, npix)
    def k1():
        eng.srf_integrate_moments(c, plan.table, rr, 3, plan.ws, None, 0.0, 0.0, out=out, reduce=False, layout=plan.layout, real_layout=rl, opts=plan.opts)
    k1(); best = 9
    for _ in range(3):
        e0.record(); k1(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
cubes = [p.cube] + [p.cube.clone() for _ in range(N - 1)]
for c in cubes:
    a = c.data_ptr()
    print(f"{a:#016x}  mod2MB {a % (1<<21):#08x}  mod1GB {(a % (1<<30)) >> 20:5d} MB  GB {a >> 30:6d}   {t(c):.4f}")
