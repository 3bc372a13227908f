#!/usr/bin/env python3
"""PCIe-inclusive rate of the NumPy-facing drop-in call (host cube in, host planes out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np
import torch
import s2_emit
from s2_emit.synthetic import gaussian_srf, emit_wavelengths

H = W = 1024
w, good = emit_wavelengths()
srf = gaussian_srf()
R = np.random.default_rng(0).random((H, W, 285), dtype=np.float32)
s2_emit.pseudo_s2_srf_integral(R[:64], w, srf, good)          # warm up (library load, allocator)
for rep in range(3):
    t0 = time.perf_counter()
    out = s2_emit.pseudo_s2_srf_integral(R, w, srf, good)
    dt = time.perf_counter() - t0
    print(f"numpy in/out 1024x1024x285: {dt*1e3:.1f} ms  -> {H*W*285/dt/1e6:.0f} Mpix*bands/s PCIe-inclusive "
          f"({R.nbytes/dt/1e9:.1f} GB/s of cube bytes)")
Rp = torch.from_numpy(R).pin_memory()
torch.cuda.synchronize()
t0 = time.perf_counter(); Rd = Rp.cuda(non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"pinned H2D of the cube alone: {dt*1e3:.1f} ms ({R.nbytes/dt/1e9:.1f} GB/s)")
t0 = time.perf_counter(); o = s2_emit.pseudo_s2_srf_integral(Rd, w, srf, good); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"device-resident call (torch in/out): {dt*1e3:.2f} ms")
