"""Randomised host tiles through SpectralFusion.stream (changing shapes, masks that come and go, float32 / uint16, depths 1-3):
every yielded tile must carry the bits of step() on the same tile.  python tools/dbg/stress_stream.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
prob = device_problem(8, 8, 285, deg=3, seed=seed, device=torch.device("cuda", 0))
bad = 0
for k in range(cases):
    deg = int(rng.integers(1, 5))
    u16 = rng.random() < 0.4
    kw = dict(deg=deg, min_valid=0.0, min_count=5, apply_mask=bool(rng.random() < 0.5), clip=True)
    plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, **kw)
    ref = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, **kw)
    nb = plan.table.nb
    tiles = []
    shape = (int(rng.integers(1, 120)), int(rng.integers(1, 120)))
    for i in range(int(rng.integers(1, 9))):
        if rng.random() < 0.3:
            shape = (int(rng.integers(1, 120)), int(rng.integers(1, 120)))
        H, W = shape
        c = (rng.random((H, W, 285)) * 0.6).astype(np.float32)
        if u16:
            c = np.clip(np.rint(c * 10000), 0, 65534).astype(np.uint16)
        r = rng.random((H, W, eng.padded_row(nb))).astype(np.float32)
        m = None if rng.random() < 0.5 else (rng.random(H * W) > 0.3).astype(np.uint8)
        tiles.append((c, r, m))
    depth = int(rng.integers(1, 4))
    n = 0
    ok = True
    for idx, coeffs, matched, out in plan.stream(tiles, depth=depth, to_host=True):
        c, r, m = tiles[idx]
        want = ref.step(torch.from_numpy(c).cuda(), torch.from_numpy(r).cuda(), None if m is None else torch.from_numpy(m).cuda(), reuse_buffers=False)
        ok = ok and np.array_equal(np.asarray(coeffs).view(np.int64), want.coeffs.cpu().numpy().view(np.int64)) \
            and np.array_equal(np.asarray(matched).view(np.int32).reshape(-1), want.matched.cpu().numpy().view(np.int32).reshape(-1))
        n += 1
    if not (ok and n == len(tiles)):
        bad += 1
        print("STREAM MISMATCH", len(tiles), depth, deg, u16, kw, n, flush=True)
    plan.close(); ref.close()
print("stream stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
