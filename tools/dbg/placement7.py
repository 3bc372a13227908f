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
out = torch.empty((npix, 12), device="cuda")
dummy = torch.zeros(1 << 20, device="cuda")
def k1(cube, k, n=9):
    """K1 (deg 0) preceded by a dummy elementwise kernel over k * 1024 elements (k workgroups of 256 x 4)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for it in range(n + 2):
        if k > 0: dummy[:k * 1024].add_(1.0)
        e0.record(); eng.srf_integrate(cube, table, out=out, layout="pixmajor"); e1.record(); e1.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
cubes = [prob.cube] + [prob.cube.clone() for _ in range(3)]
ks = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17]
print("dummy WGs  " + " ".join(f"{k:>6d}" for k in ks))
for ci, c in enumerate(cubes):
    print(f"cube {ci} us  " + " ".join(f"{k1(c, k)*1000:6.1f}" for k in ks), flush=True)
