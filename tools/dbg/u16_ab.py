"""K1 on a uint16 cube with the library named by HSR_LIBRARY (A/B of diagnostic builds on one box):
HSR_LIBRARY=tools/dbg/libhsr_<variant>.so python tools/dbg/u16_ab.py [fast]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
torch.cuda.set_device(0)
p = device_problem(1024, 1024, 285, deg=3, seed=0)
cube = eng.tile_encode_u16(p.cube)
fast = len(sys.argv) > 1 and sys.argv[1] == "fast"
plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0, u16_fast=fast)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(300): plan.step(cube, p.real)          # settle
ts = []
for i in range(60):
    plan.step(cube, p.real, k1_events=(e0, e1))
    e1.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
ref = plan.step(cube, p.real)
print(os.path.basename(os.environ.get("HSR_LIBRARY", "prod")), "fast" if fast else "exact", f"K1 median {ts[len(ts)//2]:.4f} min {ts[0]:.4f} ms",
      "checksum", float(ref.coeffs.sum()), float(ref.matched[:, :12].double().sum()))
