"""Randomised batches of ragged tiles through step_batch (float32 / uint16): every tile must carry the bits of its own
operator-by-operator step().  python tools/dbg/stress_batch.py [seed] [cases]"""
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
names = list(prob.srf.keys())
bad = 0
for k in range(cases):
    T = int(rng.integers(1, 30))
    sel = [None, ("B4", "B3", "B2"), tuple(names[:7])][int(rng.integers(0, 3))]
    srf = prob.srf if sel is None else {n: prob.srf[n] for n in sel}
    deg = int(rng.integers(1, 5))
    u16 = rng.random() < 0.4
    kw = dict(deg=deg, min_valid=0.0, min_count=int(rng.choice([0, 5, 50])), apply_mask=bool(rng.random() < 0.5), clip=bool(rng.random() < 0.7),
              u16_fast=bool(u16 and rng.random() < 0.5))
    plan = SpectralFusion(prob.emit_w, srf, prob.good_mask, **kw)
    nb = plan.table.nb
    cubes, reals, masks = [], [], []
    for i in range(T):
        npix_shape = [(1, 1), (1, int(rng.integers(1, 200))), (int(rng.integers(1, 120)), int(rng.integers(1, 120))), (100, 100), (int(rng.integers(120, 260)), 300)][int(rng.integers(0, 5))]
        H, W = npix_shape
        c = torch.rand((H, W, 285), generator=g, device="cuda") * 0.6
        cubes.append(eng.tile_encode_u16(c) if u16 else c)
        reals.append(torch.rand((H, W, eng.padded_row(nb)), generator=g, device="cuda"))
        masks.append(None if rng.random() < 0.5 else (torch.rand(H * W, generator=g, device="cuda") > float(rng.random())).to(torch.uint8))
    out = plan.step_batch(cubes, reals, masks)
    torch.cuda.synchronize()
    ok = True
    for i in range(T):
        o = plan.step(cubes[i], reals[i], masks[i], reuse_buffers=False)
        ti = out.tile(i)
        ok = ok and torch.equal(o.coeffs.view(torch.int64), ti.coeffs.view(torch.int64)) and torch.equal(o.moments.view(torch.int64), ti.moments.view(torch.int64)) \
            and torch.equal(o.pseudo.view(torch.int32), ti.pseudo.view(torch.int32)) and torch.equal(o.matched.view(torch.int32), ti.matched.view(torch.int32))
    if not ok:
        bad += 1
        print("MISMATCH", T, nb, deg, u16, kw, flush=True)
    plan.close()
print("batch stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
