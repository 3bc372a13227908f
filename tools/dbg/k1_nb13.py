#!/usr/bin/env python3
"""K1+K2 on a 1024 x 1024 x 285 tile with the 12-band table (B10 masked out: rows of 12 floats) and with all 13 Sentinel-2 bands (rows
of 16 floats): ADVICE r3 - does the 13-band launch still keep two workgroups per CU?  Prints ms per launch (HIP events, 30 launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from oracle import oracle_np as onp
from s2_emit import SpectralFusion, _engine as eng

w, good = onp.synthetic_wavelengths()
srf = onp.synthetic_srf()
H = W = 1024
g = torch.Generator(device="cuda")
g.manual_seed(1)
cube = torch.rand((H, W, 285), generator=g, device="cuda") * 0.6
for label, gm in (("12 bands (good_mask), rows of 12", good), ("13 bands (no good_mask), rows of 16", None)):
    plan = SpectralFusion(w, srf, gm, deg=3, min_valid=0.0)
    nb = plan.table.nb
    real = torch.rand((H, W, eng.padded_row(nb)), generator=g, device="cuda")
    for fused in (False, True):
        p = SpectralFusion(w, srf, gm, deg=3, min_valid=0.0, fuse_apply=fused)
        run = (lambda: p.submit(cube, real)) if fused else (lambda: p.step(cube, real))
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{label}: {'fused pipeline' if fused else 'step()'} {e0.elapsed_time(e1) / 30:.4f} ms per tile", flush=True)
        p.close()
