#!/usr/bin/env python3
"""percentile_limits at the reference's tile sizes (100 x 100 ... 1024 x 1024; planes and band-last rows of 4): time per call and
equality with np.percentile (exact order statistics + NumPy's lerp)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import _engine as eng
torch.manual_seed(0)
def timed(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("| image | layout | us per call | == np.percentile |\n|---|---|---|---|")
for side in (100, 600, 1024, 1448):
    n = side * side
    x = torch.rand((3, n), device="cuda") ** 2
    m = (torch.rand(n, device="cuda") > 0.1).to(torch.uint8)
    got = eng.percentile_limits(x, m, 2, 98).cpu().numpy()
    xm = x.cpu().numpy()[:, m.cpu().numpy() != 0]
    want = np.stack([np.percentile(xm[c], [2, 98]) for c in range(3)])
    us = timed(lambda: eng.percentile_limits(x, m, 2, 98))
    print(f"| {side} x {side} x 3 | planes | {us:.1f} | {bool(np.array_equal(got, want))} |", flush=True)
    xr = torch.rand((n, 4), device="cuda") ** 2
    got = eng.percentile_limits(xr, m, 2, 98, "pixmajor", nb=3).cpu().numpy()
    xm = xr.cpu().numpy()[m.cpu().numpy() != 0]
    want = np.stack([np.percentile(xm[:, c], [2, 98]) for c in range(3)])
    us = timed(lambda: eng.percentile_limits(xr, m, 2, 98, "pixmajor", nb=3))
    print(f"| {side} x {side} x 3 | band-last rows of 4 | {us:.1f} | {bool(np.array_equal(got, want))} |", flush=True)
