"""Sinkhorn barycentric targets at the reference's size (5000 x 5000, reg 0.05, 300 iterations)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import _ot
rng = np.random.default_rng(0)
X = rng.random((5000, 3)); Y = np.clip(X[rng.integers(0, 5000, 5000)] ** 0.8 + 0.02 * rng.standard_normal((5000, 3)), 0, 1)
Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
for thr in (1e-6, 0.0):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        yb, info = _ot.barycentric_targets_device(Xd, Yd, 0.05, 300, thr, return_info=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        it = (info["conv_iter"] + 1) if info["conv_iter"] is not None else 300
        print(f"stopThr={thr:g}: {dt*1e3:.2f} ms wall, stopped after {it} iterations, {info}", flush=True)
        if thr != 0.0:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _ot.barycentric_targets_device(Xd, Yd, 0.05, 300, thr, poll_every=50)
            torch.cuda.synchronize()
            print(f"   with poll_every=50: {(time.perf_counter()-t0)*1e3:.2f} ms wall", flush=True)
        if thr == 0.0:
            passes = 2 * 300 + 30
            print(f"   {passes} passes over the 200 MB kernel matrix -> {passes*200e6/dt/1e12:.2f} TB/s effective", flush=True)
