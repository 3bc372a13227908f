"""Randomised fits of the polynomial-ridge variant (10 inputs, degree 3) against a float64 NumPy solve of the same system.
python tools/dbg/stress_ridge.py [seed] [cases]"""
import os, sys
from itertools import combinations_with_replacement as cwr
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
import s2_emit
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
combos = [c for d in (1, 2, 3) for c in cwr(range(10), d)]
def feats(z):
    return np.stack([np.prod(z[:, list(c)], axis=1) for c in combos], axis=1)
bad = 0
for k in range(cases):
    n = int(rng.choice([300, 1000, 4097, 10000, 29127, 40001]))
    T = int(rng.choice([1, 6, 16, 31, 32, 33, 48, 64, 80, 97, 150, 285]))
    base = rng.random((n, 4))
    X = (600 + 4000 * np.clip(base @ rng.random((4, 10)) / 2 + 0.02 * rng.standard_normal((n, 10)), 0, 1)).astype(np.float32)
    Yr = np.clip(base @ rng.random((4, T)) / 3 + 0.01 * rng.standard_normal((n, T)), 0.001, 0.6)
    Y = np.log(Yr / (1 - Yr))
    m = s2_emit.PolyRidge(degree=3, alpha=1.0).fit(X, Y)
    X64 = X.astype(np.float64)
    mean, scale = X64.mean(0), X64.std(0)
    scale[scale == 0] = 1.0
    P = feats((X64 - mean) / scale)
    pm, ym = P.mean(0), Y.mean(0)
    Pc, Yc = P - pm, Y - ym
    Wt = np.linalg.solve(Pc.T @ Pc + np.eye(285), Pc.T @ Yc)
    b = ym - pm @ Wt
    Xt = X[:: max(1, n // 500)]
    ref = feats((Xt.astype(np.float64) - mean) / scale) @ Wt + b
    got = m.predict(Xt)
    err = float(np.abs(got - ref).max())
    if not err < 3e-4:
        bad += 1
        print("RIDGE MISMATCH", n, T, err, flush=True)
print("ridge stress done; failures:", bad, flush=True)
sys.exit(1 if bad else 0)
