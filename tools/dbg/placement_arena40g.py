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
GBs = 40
arena = torch.empty(GBs << 28, dtype=torch.float32, device="cuda")      # 40 GB
def k1(cube, out, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for it in range(n + 2):
        e0.record(); eng.srf_integrate(cube, table, out=out, layout="pixmajor"); e1.record(); e1.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
cubes = [prob.cube, prob.cube.clone()]
print("cubes @", [hex(c.data_ptr()) for c in cubes], "arena @", hex(arena.data_ptr()))
line = []
for off_mb in range(0, GBs * 1024 - 64, 256):
    o = off_mb << 20
    out = arena[o // 4: o // 4 + npix * 12].view(npix, 12)
    ts = [k1(c, out) for c in cubes]
    line.append((off_mb, ts))
for c in range(2):
    print(f"cube {c}: " + "".join("F" if t[c] < 0.208 else ("s" if t[c] > 0.213 else "?") for _, t in line))
print("(one character per 256 MB of arena offset; F < 0.208 ms, s > 0.213 ms)")
print(" ".join(f"{t[0]*1000:.0f}" for _, t in line[:64]))
