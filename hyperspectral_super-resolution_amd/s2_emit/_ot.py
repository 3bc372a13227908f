"""Entropic OT sub-step of fit_ot_poly_rgb / ot_match_rgb_sinkhorn_pot.

Reference: s2_emit/poly_regression.py:31-56 and s2_emit/color.py:78-104 call POT
(``ot.dist`` + ``ot.sinkhorn``).  POT is an unpinned dependency that is absent offline, so this
step is **parity unpinned**: it follows POT's documented ``sinkhorn_knopp`` (K = exp(-M/reg);
v <- b/(K^T u); u <- a/(K v); error check every 10th iteration on ||v*(K^T u) - b||_2) and is
validated by OT invariants only.  The dense 5000 x 5000 float64 kernel matrix lives in HBM and
the mat-vecs are plain library GEMVs through torch (a "next" row, not a
hand-written kernel yet).  Sampling stays on the host so the PCG64 stream matches the reference.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat


def sample_pairs(src_rgb, ref_rgb, mask, n_samples, seed, min_rows):
    """Rows inside the mask, non-finite rows dropped for X and Y independently, then
    ``default_rng(seed).choice(..., replace=False)`` for X then Y (poly_regression.py:31-47)."""
    rng = np.random.default_rng(seed)
    X_all = np.asarray(src_rgb)[mask].reshape(-1, 3).astype(np.float64)
    Y_all = np.asarray(ref_rgb)[mask].reshape(-1, 3).astype(np.float64)
    X_all = X_all[np.isfinite(X_all).all(axis=1)]
    Y_all = Y_all[np.isfinite(Y_all).all(axis=1)]
    if X_all.shape[0] < min_rows or Y_all.shape[0] < min_rows:
        return None
    ns = min(n_samples, X_all.shape[0])
    nt = min(n_samples, Y_all.shape[0])
    X = X_all[rng.choice(X_all.shape[0], size=ns, replace=False)]
    Y = Y_all[rng.choice(Y_all.shape[0], size=nt, replace=False)]
    return X, Y


def barycentric_targets_device(Xd, Yd, reg=0.05, numItermax=300, stopThr=1e-6):
    """Xd (ns,3), Yd (nt,3) float64 GPU tensors -> Ybar (ns,3) float64 GPU tensor."""
    torch = nat.require_gpu()
    ns, nt = Xd.shape[0], Yd.shape[0]
    a = torch.full((ns,), 1.0 / ns, dtype=torch.float64, device=Xd.device)
    b = torch.full((nt,), 1.0 / nt, dtype=torch.float64, device=Xd.device)
    M = (Xd * Xd).sum(1)[:, None] + (Yd * Yd).sum(1)[None, :] - 2.0 * (Xd @ Yd.T)
    M.clamp_(min=0.0)
    K = torch.exp(M / (-reg))
    u = torch.full((ns,), 1.0 / ns, dtype=torch.float64, device=Xd.device)
    v = torch.full((nt,), 1.0 / nt, dtype=torch.float64, device=Xd.device)
    for ii in range(numItermax):
        uprev, vprev = u, v
        KtU = K.T @ u
        v = b / KtU
        u = a / (K @ v)
        if ii % 10 == 0:     # the only host synchronisations: every 10th iteration, as POT checks
            bad = (KtU == 0).any() | ~torch.isfinite(u).all() | ~torch.isfinite(v).all()
            if bool(bad):
                u, v = uprev, vprev
                break
            err = torch.linalg.vector_norm(v * (K.T @ u) - b)
            if float(err) < stopThr:
                break
    P = u[:, None] * K * v[None, :]
    return (P @ Yd) / (P.sum(dim=1, keepdim=True) + 1e-32)


def barycentric_targets(X, Y, reg=0.05, numItermax=300, stopThr=1e-6) -> np.ndarray:
    torch = nat.require_gpu()
    Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    Yd = torch.from_numpy(np.ascontiguousarray(Y)).cuda()
    return barycentric_targets_device(Xd, Yd, reg, numItermax, stopThr).cpu().numpy()
