"""Randomised mosaics through fuse_mosaic (per-tile launches and the resident / batched form): both forms must agree bit for
bit with each other and with moments summed in tile order from per-tile step() calls.  python tools/dbg/stress_mosaic.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda")
g.manual_seed(seed)
prob = device_problem(8, 8, 285, deg=3, seed=seed, device=torch.device("cuda", 0))
bad = 0
for k in range(cases):
    T = int(rng.integers(1, 9))
    deg = int(rng.integers(1, 5))
    u16 = rng.random() < 0.3
    kw = dict(deg=deg, min_valid=0.0, min_count=5, apply_mask=bool(rng.random() < 0.5), clip=True)
    plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, **kw)
    nb = plan.table.nb
    tiles, masks = [], []
    for i in range(T):
        H, W = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        c = torch.rand((H, W, 285), generator=g, device="cuda") * 0.6
        tiles.append((eng.tile_encode_u16(c) if u16 else c, torch.rand((H, W, eng.padded_row(nb)), generator=g, device="cuda")))
        masks.append(None if rng.random() < 0.5 else (torch.rand(H * W, generator=g, device="cuda") > 0.3).to(torch.uint8))
    c1, m1, o1 = plan.fuse_mosaic(tiles, masks)
    c1, m1 = c1.clone(), m1.clone()
    o1 = [(o.pseudo.clone(), o.matched.clone()) for o in o1]
    c2, m2, o2 = plan.fuse_mosaic(tiles, masks, resident=True)
    why = []
    if not torch.equal(m1.view(torch.int64), m2.view(torch.int64)): why.append("moments resident vs per-tile (max rel %.2e)" % float(((m1 - m2).abs() / m1.abs().clamp_min(1e-300)).max()))
    if not torch.equal(c1.view(torch.int64), c2.view(torch.int64)): why.append("coeffs resident vs per-tile")
    for i, ((p1, q1), b) in enumerate(zip(o1, o2)):
        if not torch.equal(p1.view(torch.int32), b.pseudo.view(torch.int32)): why.append("pseudo %d" % i)
        if not torch.equal(q1.view(torch.int32), b.matched.view(torch.int32)): why.append("matched %d" % i)
    ok = not why
    # independent: per-tile moments from the operator path, summed in float64 in tile order - equal to the library's fixed-order
    # reduction up to the association of the sum (a few ulp)
    tot = None
    for (c, r), m in zip(tiles, masks):
        mo = plan.step(c, r, m, reuse_buffers=False).moments
        tot = mo.clone() if tot is None else tot + mo
    rel = float(((tot - m1).abs() / m1.abs().clamp_min(1e-300)).max())
    if not rel < 1e-14: why.append("moments vs sum of step() moments (max rel %.2e)" % rel)
    if why:
        bad += 1
        print("MISMATCH", T, nb, deg, u16, kw, why[:4], flush=True)
    plan.close()
print("mosaic stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
