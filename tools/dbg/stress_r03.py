"""Randomised shapes through the kernels rebuilt in round 3 (Gram, Cholesky solve, percentile select, predict slices):
exactness / tolerance against NumPy.  python tools/dbg/stress_r03.py [seed] [cases]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
import s2_emit
from s2_emit import _native as nat, _engine as eng
from s2_emit._engine import _ptr, _stream
from itertools import combinations_with_replacement as cwr
lib = nat.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
bad = 0

# ---- Gram: small integers are exact in float64
for k in range(cases):
    na = 16 * int(rng.integers(1, 21))
    nb = na + 16 * int(rng.integers(0, 20)) if rng.random() < 0.7 else 16 * int(rng.integers(1, 30))
    n = int(rng.choice([1, 3, 8, 17, 100, 999, 4097, 20011, 60000]))
    sym = nb >= na and rng.random() < 0.7
    if sym:
        Q = rng.integers(-3, 4, (n, nb)).astype(np.float64)
        Ad = Bd = torch.from_numpy(Q).cuda()
        A, B, lda, ldb = Q, Q, nb, nb
    else:
        A = rng.integers(-3, 4, (n, na)).astype(np.float64)
        B = rng.integers(-3, 4, (n, nb)).astype(np.float64)
        Ad, Bd, lda, ldb = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), na, nb
    work = torch.empty(max(1, lib.hsr_gram_work_bytes(na, nb, n) // 8), dtype=torch.float64, device="cuda")
    Cd = torch.full((na, nb), -7.0, dtype=torch.float64, device="cuda")
    nat.check(lib.hsr_gram_f64(_ptr(Ad), lda, na, _ptr(Bd), ldb, nb, n, _ptr(work), _ptr(Cd), nb, _stream(torch)))
    if not np.array_equal(Cd.cpu().numpy(), A[:, :na].T @ B[:, :nb]):
        bad += 1
        print("GRAM MISMATCH", n, na, nb, sym, flush=True)
print("gram done", flush=True)

# ---- Cholesky solve
for k in range(cases // 2):
    n = 32 * int(rng.integers(1, 17))
    T = int(rng.choice([1, 5, 16, 17, 32, 33, 100, 285]))
    M = rng.standard_normal((n, n + 8))
    A = M @ M.T / n + 0.5 * np.eye(n)
    Bm = rng.standard_normal((n, T))
    Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(Bm.copy()).cuda()
    wk = torch.empty(lib.hsr_chol_work_bytes(n) // 8, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    nat.check(lib.hsr_chol_solve_f64(_ptr(Ad), n, n, _ptr(Bd), T, T, _ptr(wk), _ptr(info), _stream(torch)))
    X = Bd.cpu().numpy()
    ref = np.linalg.solve(A, Bm)
    err = np.abs(X - ref).max() / max(1e-30, np.abs(ref).max())
    if int(info.item()) != 0 or not err < 1e-9:
        bad += 1
        print("CHOL MISMATCH", n, T, err, int(info.item()), flush=True)
print("chol done", flush=True)

# ---- percentiles, planar and band-last rows of 4
for k in range(cases):
    npix = int(rng.choice([1, 2, 5, 63, 64, 65, 1000, 4096, 70001, 300007]))
    nbc = int(rng.integers(1, 4))
    kind = int(rng.integers(0, 4))
    x = [rng.random((npix, nbc)), rng.standard_normal((npix, nbc)) * 10.0 ** float(rng.integers(-3, 4)), np.round(rng.random((npix, nbc)) * 4) / 4,
         np.full((npix, nbc), 0.125)][kind].astype(np.float32)
    frac = float(rng.choice([1.0, 0.5, 0.01]))
    mask = rng.random(npix) < frac
    mask[0] = True
    pmin, pmax = [(2, 98), (0, 100), (25, 75), (50, 50), (0.1, 99.9)][int(rng.integers(0, 5))]
    ref = np.array([np.percentile(x[mask, c], [pmin, pmax]) for c in range(nbc)])
    md = torch.from_numpy(mask.view(np.uint8)).cuda()
    got = eng.percentile_limits(torch.from_numpy(np.ascontiguousarray(x.T)).cuda(), md, pmin, pmax).cpu().numpy()
    rows = np.zeros((npix, 4), np.float32)
    rows[:, :nbc] = x
    got4 = eng.percentile_limits(torch.from_numpy(rows).cuda(), md, pmin, pmax, "pixmajor", nb=nbc).cpu().numpy()
    if not (np.array_equal(got, ref) and np.array_equal(got4, ref)):
        bad += 1
        print("PERCENTILE MISMATCH", npix, nbc, kind, frac, pmin, pmax, got, got4, ref, flush=True)
print("percentile done", flush=True)

# ---- predict: random target counts against the float64 oracle
for T in sorted(set(int(t) for t in rng.integers(1, 400, 12)) | {1, 32, 33, 64, 65, 96, 97, 160, 161, 288, 289}):
    npx = int(rng.choice([1, 31, 129, 513, 2049]))
    mean, scale = rng.random(10) * 0.3 + 0.2, rng.random(10) * 0.2 + 0.05
    coef, b = rng.normal(0, 0.05, (T, 285)), rng.normal(0, 0.1, T)
    m = s2_emit.PolyRidge.from_params(mean, scale, coef, b)
    X = (rng.random((npx, 10)) * 0.5 + 0.1).astype(np.float32)
    got = m.predict(X)
    z = (X.astype(np.float64) - mean) / scale
    phi = np.stack([np.prod(z[:, list(c)], axis=1) for d in (1, 2, 3) for c in cwr(range(10), d)], axis=1)   # sklearn's order
    ref = phi @ coef.T + b
    err = np.abs(got - ref).max()
    if not err < 2e-4:
        bad += 1
        print("PREDICT MISMATCH", T, npx, err, flush=True)
print("predict done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
