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
arena = torch.empty(npix * 12 + (64 << 20), dtype=torch.float32, device="cuda")      # 48 MB + 256 MB of slack
def k1(cube, out, n=9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(2): eng.srf_integrate(cube, table, out=out, layout="pixmajor")
    for _ in range(n):
        e0.record(); eng.srf_integrate(cube, table, out=out, layout="pixmajor"); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
offs = [0, 4096, 65536, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22, 6 << 20, 1 << 23, 12 << 20, 1 << 24, 24 << 20, 1 << 25, 1 << 26, 1 << 27]
print("cube @, arena @", hex(prob.cube.data_ptr()), hex(arena.data_ptr()))
cubes = [prob.cube, prob.cube.clone(), prob.cube.clone()]
print("out offset (KB) " + " ".join(f"{o//1024:>6d}" for o in offs))
for ci, c in enumerate(cubes):
    row = []
    for o in offs:
        out = arena[o // 4: o // 4 + npix * 12].view(npix, 12)
        row.append(k1(c, out))
    print(f"cube {ci} K1 us      " + " ".join(f"{v*1000:6.1f}" for v in row), flush=True)
