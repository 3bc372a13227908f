#!/usr/bin/env python3
"""hsr_percentile_limits on large images through ctypes (preallocated workspace): noise (torch.rand, what tools/bench_aux.py times) and
a smooth image (a coarse random field upsampled 6 x, what the driver's 10 m phase produces); planes and band-last rows; checked
against torch.quantile-free NumPy percentiles on a subsample of sizes."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np
import torch
from s2_emit import _native as nat
lib = nat.load()
torch.manual_seed(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sides = [int(a) for a in sys.argv[1:]] or [1024, 2048, 6144]
for side in sides:
    n = side * side
    for kind in ("noise", "smooth"):
        if kind == "noise":
            base = torch.rand((3, side, side), device="cuda")
        else:
            c = torch.rand((1, 3, side // 6 + 2, side // 6 + 2), device="cuda") ** 2
            base = torch.nn.functional.interpolate(c, scale_factor=6, mode="bilinear", align_corners=False)[0, :, :side, :side].contiguous()
        m = (torch.rand(n, device="cuda") > 0.1).to(torch.uint8)
        for layout in ("planes", "rows4"):
            if layout == "planes":
                x = base.reshape(3, n).contiguous()
                bs, ps = n, 1
            else:
                x = torch.zeros((n, 4), device="cuda")
                x[:, :3] = base.reshape(3, n).t()
                bs, ps = 1, 4
            work = torch.empty(lib.hsr_percentile_work_bytes(3) // 8 + 1, dtype=torch.int64, device="cuda")
            lohi = torch.empty((3, 2), dtype=torch.float64, device="cuda")
            def call():
                nat.check(lib.hsr_percentile_limits(x.data_ptr(), bs, ps, m.data_ptr(), n, 3, 2.0, 98.0, work.data_ptr(), lohi.data_ptr(), st))
            for _ in range(5): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            reps = 50
            for _ in range(reps): call()
            e1.record(); torch.cuda.synchronize()
            first = lohi.clone()
            same = True
            for _ in range(30):
                call()
                same = same and bool(torch.equal(first, lohi))
            ok = "" if same else " UNSTABLE"
            if side <= 2048:
                mm = m.cpu().numpy().astype(bool)
                xs = (x if layout == "planes" else x[:, :3].t()).cpu().numpy()
                ref = np.stack([np.percentile(xs[c][mm], [2.0, 98.0]) for c in range(3)])
                ok += " exact" if np.array_equal(ref, lohi.cpu().numpy()) else " MISMATCH"
            print(f"{side} x {side} x 3 {kind:6s} {layout:6s} | {e0.elapsed_time(e1) / reps * 1e3:7.1f} us{ok}", flush=True)
