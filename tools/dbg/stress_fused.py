"""Randomised tiles through the fused one-kernel-per-tile pipeline, the two-slot pipeline and the prepared step() (float32 and
uint16 cubes): every tile must carry the bits of the operator-by-operator path (step(reuse_buffers=False)).
python tools/dbg/stress_fused.py [seed] [cases] [tiny | mid | big]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda")
g.manual_seed(seed)
prob = device_problem(8, 8, 285, deg=3, seed=seed, device=torch.device("cuda", 0))     # wavelengths / SRF table
names = list(prob.srf.keys())
bad = 0
for k in range(cases):
    size = sys.argv[3] if len(sys.argv) > 3 else "mid"       # tiny: 1-40 pixel edges, mid: 1-300, big: 300-1100
    lo, hi = {"tiny": (1, 40), "mid": (1, 300), "big": (300, 1100)}[size]
    H, W = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
    npix = H * W
    sel = [None, ("B4", "B3", "B2"), tuple(names[:7])][int(rng.integers(0, 3))]
    srf = prob.srf if sel is None else {n: prob.srf[n] for n in sel}
    gm = prob.good_mask if rng.random() < 0.7 else None
    deg = int(rng.integers(1, 5))
    u16 = rng.random() < 0.4
    kw = dict(deg=deg, min_valid=0.0, min_count=int(rng.choice([0, 5, 50])), apply_mask=bool(rng.random() < 0.5), clip=bool(rng.random() < 0.7),
              u16_fast=bool(u16 and rng.random() < 0.5))
    ref = SpectralFusion(prob.emit_w, srf, gm, **kw)
    mode = int(rng.integers(0, 3))             # 0: fused pipeline, 1: two-slot pipeline, 2: prepared step()
    pipe = SpectralFusion(prob.emit_w, srf, gm, fuse_apply=(mode == 0), **kw)
    nb = ref.table.nb
    row = eng.padded_row(nb)
    cubes = []
    for i in range(3):
        c = torch.rand((H, W, 285), generator=g, device="cuda") * 0.6
        if rng.random() < 0.3:
            c.view(-1)[int(rng.integers(0, c.numel()))] = float("nan")
        cubes.append(eng.tile_encode_u16(c) if u16 else c)
    reals = [torch.rand((H, W, row), generator=g, device="cuda") for _ in range(3)]
    masks = [None if rng.random() < 0.4 else (torch.rand(npix, generator=g, device="cuda") > float(rng.random())).to(torch.uint8) for _ in range(4)]
    seq = [(cubes[i % 3], reals[(i + 1) % 3], masks[i % 4]) for i in range(int(rng.integers(1, 8)))]
    got = []
    for c, r, m in seq:
        o = pipe.step(c, r, m) if mode == 2 else pipe.submit(c, r, m)
        if o is not None:
            got.append(tuple(t.clone() for t in (o.pseudo, o.matched, o.moments, o.coeffs)))
    if mode != 2:
        got += [tuple(t.clone() for t in (o.pseudo, o.matched, o.moments, o.coeffs)) for o in pipe.drain()]
    ok = len(got) == len(seq)
    for (c, r, m), gt in zip(seq, got):
        w = ref.step(c, r, m, reuse_buffers=False)
        ok = ok and torch.equal(gt[0].view(torch.int32), w.pseudo.view(torch.int32)) and torch.equal(gt[1].view(torch.int32), w.matched.view(torch.int32)) \
            and torch.equal(gt[2].view(torch.int64), w.moments.view(torch.int64)) and torch.equal(gt[3].view(torch.int64), w.coeffs.view(torch.int64))
    if not ok:
        bad += 1
        print("MISMATCH", H, W, nb, deg, u16, kw, len(seq), mode, flush=True)
    pipe.close()
    ref.close()
print("fused pipeline stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
