"""Entropic OT sub-step of fit_ot_poly_rgb / ot_match_rgb_sinkhorn_pot.

Reference: s2_emit/poly_regression.py:31-56 and s2_emit/color.py:78-104 call POT
(``ot.dist`` + ``ot.sinkhorn``).  POT is an unpinned dependency that is absent offline, so this
step is **parity unpinned**: it follows POT's documented ``sinkhorn_knopp`` (K = exp(-M/reg);
v <- b/(K^T u); u <- a/(K v); error check every 10th iteration on ||v*(K^T u) - b||_2) and is
validated against the oracle's restatement and OT invariants only.  The dense 5000 x 5000 float64 kernel
matrix lives in HBM; the two mat-vecs per iteration, the breakdown / convergence state and the barycentric
projection are hand-written kernels (csrc/hsr_ot.hip), enqueued in one call without host
synchronisation.  Sampling stays on the host so the PCG64 stream matches the reference.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _native as nat


def _finite_rows(image, mask):
    rows = np.asarray(image)[mask].reshape(-1, 3).astype(np.float64)
    return rows[np.isfinite(rows).all(axis=1)]


def sample_pairs(src_rgb, ref_rgb, mask, n_samples, seed, min_rows):
    """Source and reference sample rows for the OT fit, or None when either side has fewer than ``min_rows`` usable rows.
    Contract (poly_regression.py:31-47): rows inside the mask; rows with a non-finite channel dropped for the two images
    independently; ONE ``default_rng(seed)`` generator draws without replacement first the source rows, then the
    reference rows - that order is what makes the draw reproduce the reference's PCG64 stream."""
    pools = [_finite_rows(src_rgb, mask), _finite_rows(ref_rgb, mask)]
    if min(len(p) for p in pools) < min_rows:
        return None
    gen = np.random.default_rng(seed)
    picked = [p[gen.choice(len(p), size=min(n_samples, len(p)), replace=False)] for p in pools]    # source first
    return picked[0], picked[1]


def sample_pairs_device(src, ref, mask, n_samples, seed, min_rows):
    """Device form of sample_pairs: src, ref (H,W,3) GPU tensors, mask (H,W) bool / uint8 GPU tensor.
    The row selection happens on the GPU (row-major order of the True mask positions, non-finite rows dropped
    for X and Y independently - exactly what ``src_rgb[mask]`` + the isfinite filter produce on the host); only
    the two row counts come back, so that ``default_rng(seed).choice`` draws the reference's indices, which are
    then gathered on the device.  Returns (X, Y) float64 GPU tensors or None (fewer than ``min_rows`` rows).
    On a 1024 x 1024 image the host version spends ~60 ms in boolean indexing; this one well under 1 ms."""
    torch = nat.require_gpu()
    s3 = src.reshape(-1, src.shape[-1])[:, :3]
    r3 = ref.reshape(-1, ref.shape[-1])[:, :3]
    m = mask.reshape(-1).to(torch.bool)
    idx_x = torch.nonzero(m & torch.isfinite(s3).all(dim=1)).squeeze(1)
    idx_y = torch.nonzero(m & torch.isfinite(r3).all(dim=1)).squeeze(1)
    nx, ny = int(idx_x.numel()), int(idx_y.numel())          # the synchronisation of this step
    if nx < min_rows or ny < min_rows:
        return None
    rng = np.random.default_rng(seed)
    sel_x = rng.choice(nx, size=min(n_samples, nx), replace=False)
    sel_y = rng.choice(ny, size=min(n_samples, ny), replace=False)
    gx = idx_x[torch.from_numpy(sel_x).to(idx_x.device)]
    gy = idx_y[torch.from_numpy(sel_y).to(idx_y.device)]
    return s3[gx].to(torch.float64).contiguous(), r3[gy].to(torch.float64).contiguous()


def barycentric_targets_device(Xd, Yd, reg=0.05, numItermax=300, stopThr=1e-6, return_info: bool = False,
                               poll_every: Optional[int] = None):
    """Xd (ns,3), Yd (nt,3) float64 GPU tensors -> Ybar (ns,3) float64 GPU tensor.
    By default one C-ABI call enqueues the kernel matrix, every Sinkhorn iteration and the barycentric projection
    (csrc/hsr_ot.hip); nothing synchronises with the host.  With ``poll_every=k`` the iterations are enqueued in
    blocks of k and the device state is read after each block (one synchronisation per block), so that nothing is
    enqueued after convergence - same result, ~2 ms less when the solve converges early.  ``return_info`` adds a
    dict with the iteration the loop stopped at (reads the device state: one synchronisation)."""
    torch = nat.require_gpu()
    lib = nat.load()
    if not (Xd.is_cuda and Yd.is_cuda and Xd.dtype == torch.float64 and Yd.dtype == torch.float64
            and Xd.dim() == 2 and Yd.dim() == 2 and Xd.shape[1] == 3 and Yd.shape[1] == 3):
        raise ValueError("X and Y must be (n, 3) float64 GPU tensors")
    Xd, Yd = Xd.contiguous(), Yd.contiguous()
    ns, nt = int(Xd.shape[0]), int(Yd.shape[0])
    work = torch.empty(int(lib.hsr_ot_work_bytes(ns, nt)), dtype=torch.uint8, device=Xd.device)
    ybar = torch.empty((ns, 3), dtype=torch.float64, device=Xd.device)
    info = torch.empty(6, dtype=torch.int32, device=Xd.device)
    from ._engine import _stream
    stream = _stream(torch, Xd)          # raises if Xd is not on the current device
    never = 0x7FFFFFFF
    if poll_every is None or poll_every <= 0:
        nat.check(lib.hsr_ot_sinkhorn_barycentric(Xd.data_ptr(), ns, Yd.data_ptr(), nt, float(reg), int(numItermax),
                                                  float(stopThr), work.data_ptr(), ybar.data_ptr(), info.data_ptr(),
                                                  stream), "hsr_ot_sinkhorn_barycentric")
    else:
        nat.check(lib.hsr_ot_begin(Xd.data_ptr(), ns, Yd.data_ptr(), nt, float(reg), work.data_ptr(), stream), "hsr_ot_begin")
        done = 0
        while done < int(numItermax):
            cnt = min(int(poll_every), int(numItermax) - done)
            nat.check(lib.hsr_ot_iterate(ns, nt, done, cnt, float(stopThr), work.data_ptr(), info.data_ptr(), stream),
                      "hsr_ot_iterate")
            done += cnt
            h = info[:2].cpu()
            if int(h[0]) != never or int(h[1]) != never:
                break
        nat.check(lib.hsr_ot_finish(Yd.data_ptr(), ns, nt, done, work.data_ptr(), ybar.data_ptr(), info.data_ptr(), stream),
                  "hsr_ot_finish")
    if not return_info:
        return ybar
    h = info.cpu()
    return ybar, {"break_iter": None if int(h[0]) == never else int(h[0]),
                  "conv_iter": None if int(h[1]) == never else int(h[1]),
                  "checks": int(h[2]), "err": float(h[4:6].view(torch.float64)[0])}


def barycentric_targets(X, Y, reg=0.05, numItermax=300, stopThr=1e-6) -> np.ndarray:
    torch = nat.require_gpu()
    Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    Yd = torch.from_numpy(np.ascontiguousarray(Y)).cuda()
    return barycentric_targets_device(Xd, Yd, reg, numItermax, stopThr, poll_every=50).cpu().numpy()
