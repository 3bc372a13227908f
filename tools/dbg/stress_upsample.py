"""Randomised shapes through the producer upsampler (hsr_bilinear_upsample_mask_hist) against the three separate operators.
python tools/dbg/stress_upsample.py [seed] [cases]"""
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
    Hc, Wc, f, nb = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 9)), int(rng.integers(1, 5))
    row = eng.padded_row(nb)
    c = np.zeros((Hc * Wc, row), np.float32)
    kind = int(rng.integers(0, 3))
    c[:, :nb] = [rng.random((Hc * Wc, nb)), rng.standard_normal((Hc * Wc, nb)) * 100, np.round(rng.random((Hc * Wc, nb)) * 3) / 3][kind]
    for _ in range(int(rng.integers(0, 4))):
        c[int(rng.integers(0, Hc * Wc)), int(rng.integers(0, nb))] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
    pmin, pmax = [(2, 98), (0, 100), (10, 60)][int(rng.integers(0, 3))]
    cd = torch.from_numpy(c).cuda()
    fine, mask, lohi = eng.bilinear_upsample_mask_limits(cd, Hc, Wc, f, pmin, pmax, nb=nb)
    rf = eng.bilinear_upsample(cd, Hc, Wc, f, layout=nat.PIXMAJOR, nb=nb)
    rm = eng.valid_mask(rf, -1, None, None, nat.PIXMAJOR, nbx=nb)
    rl = eng.percentile_limits(rf, rm, pmin, pmax, nat.PIXMAJOR, nb=nb)
    ok = torch.equal(fine.view(torch.int32)[:, :nb], rf.view(torch.int32)[:, :nb]) and torch.equal(mask, rm) and torch.equal(lohi.view(torch.int64), rl.view(torch.int64))
    if not ok:
        bad += 1
        print("MISMATCH", Hc, Wc, f, nb, kind, pmin, pmax, flush=True)
print("upsample stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
