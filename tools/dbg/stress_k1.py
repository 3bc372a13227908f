"""Randomised spectral sizes, SRF sets, tile shapes and layouts through K1 (float32 and uint16 cubes) against the float64
product with the same weight table; NaN classification for finite-weight NaNs.  python tools/dbg/stress_k1.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import _engine as eng, _native as nat
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad = 0
for k in range(cases):
    B = int(rng.choice([16, 17, 31, 48, 64, 100, 151, 224, 285, 286, 300]))
    H, W = int(rng.integers(1, 150)), int(rng.integers(1, 150))
    npix = H * W
    w = np.sort(rng.random(B) * 2000 + 400).astype(np.float32)
    nbands = int(rng.integers(1, 14))
    srf = {}
    for b in range(nbands):
        c, wd = float(rng.random() * 1800 + 500), float(rng.random() * 150 + 10)
        lam = np.arange(300.0, 2600.0, 1.0)
        r = np.exp(-0.5 * ((lam - c) / wd) ** 2)
        r[r < 1e-3] = 0
        srf[f"B{b}"] = (lam, r)
    good = None if rng.random() < 0.5 else (rng.random(B) > 0.15)
    table = eng.build_srf_table(w, srf, good)
    if table.nb == 0:
        continue
    R = (rng.random((H, W, B)) * 0.6).astype(np.float32)
    u16 = rng.random() < 0.4
    layout = [nat.PLANAR, nat.PIXMAJOR][int(rng.integers(0, 2))]
    Rd = torch.from_numpy(R).cuda()
    if u16:
        cube = eng.tile_encode_u16(Rd)
        Rref = eng.tile_decode_u16(cube).cpu().numpy().astype(np.float64).reshape(npix, B) if hasattr(eng, "tile_decode_u16") else None
        if Rref is None:
            Rref = (cube.cpu().numpy().astype(np.float32) * np.float32(1e-4)).astype(np.float64).reshape(npix, B)
    else:
        cube = Rd
        Rref = R.astype(np.float64).reshape(npix, B)
        if rng.random() < 0.4:                              # a NaN in a band some SRF supports
            kk = int(table.k0[0] + table.klen[0] // 2)
            pp = int(rng.integers(0, npix))
            cube = cube.clone()
            cube.view(npix, B)[pp, kk] = float("nan")
            Rref = Rref.copy()
            Rref[pp, kk] = np.nan
    img = eng.srf_integrate(cube, table, layout=layout)
    got = (img if layout == nat.PLANAR else img[:, :table.nb].t()).cpu().numpy().astype(np.float64)      # (nb, npix)
    ref = table.weights @ Rref.T
    nanref = np.isnan(Rref).any(axis=1)
    ok = np.array_equal(np.isnan(got).all(axis=0), nanref) and not np.isnan(got[:, ~nanref]).any()
    if ok and (~nanref).any():
        den = np.abs(ref[:, ~nanref]).max()
        err = np.abs(got[:, ~nanref] - ref[:, ~nanref]).max() / max(den, 1e-30)
        ok = err < 3e-6
    if not ok:
        bad += 1
        print("K1 MISMATCH", H, W, B, table.nb, u16, layout, flush=True)
print("K1 stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
