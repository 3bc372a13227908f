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
arena = torch.empty((3 << 30) // 4, dtype=torch.float32, device="cuda")      # 3 GB
def k1(cube, out, n=7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for it in range(n + 2):
        e0.record(); eng.srf_integrate(cube, table, out=out, layout="pixmajor"); e1.record(); e1.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
c = prob.cube
GB = 1 << 30
print("cube @", hex(c.data_ptr()), "arena @", hex(arena.data_ptr()))
for off_mb in range(0, 2900, 64):
    o = off_mb << 20
    out = arena[o // 4: o // 4 + npix * 12].view(npix, 12)
    d = (out.data_ptr() - c.data_ptr()) % GB
    print(f"out offset {off_mb:5d} MB  (out - cube) mod 1GB = {d >> 20:5d} MB   out mod 1GB = {(out.data_ptr() % GB) >> 20:5d} MB   K1 {k1(c, out)*1000:6.1f} us", flush=True)
