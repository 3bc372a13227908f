"""Per-channel polynomial regression on MI355X.  Mirrors the two importable functions of reference
``s2_emit/poly_regression.py`` (lines 16-62 and 65-84; the rest of that file is a Colab script).

fit:   masked rows -> host PCG64 sampling (same stream as the reference) -> Sinkhorn barycentric
       targets on the GPU (``_ot``) -> per-channel np.polyfit replaced by Vandermonde moments
       (``hsr_poly_moments_f64``) + scaled normal-equation eigen-solve (``hsr_poly_solve``).
apply: one streaming HIP kernel (``hsr_poly_apply``): float64 Horner, mask select, clip.
"""
from __future__ import annotations

import numpy as np

from . import _engine as eng
from . import _native as nat
from . import _ot


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def polyfit_columns(X, Y, deg: int, min_count: int = 0):
    """np.polyfit(X[:, c], Y[:, c], deg) for every column c, on the GPU.
    X, Y: (n, C) float64 NumPy arrays or GPU tensors.  Returns (C, deg+1) float64 ndarray."""
    torch = nat.require_gpu()
    if not 1 <= deg <= nat.HSR_MAX_DEG:
        raise ValueError(f"deg must be in [1, {nat.HSR_MAX_DEG}] for the device fit, got {deg}")
    Xd = (X if _is_torch(X) else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64))).to("cuda", torch.float64)
    Yd = (Y if _is_torch(Y) else torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float64))).to("cuda", torch.float64)
    xt = Xd.T.contiguous()
    yt = Yd.T.contiguous()
    ws = eng.MomentWorkspace(xt.device, xt.shape[0], deg)
    mom = eng.poly_moments_f64(xt, yt, deg, ws)
    return eng.poly_solve(mom, deg, min_count).cpu().numpy()


def fit_ot_poly_rgb(
    src_rgb, ref_rgb, mask,
    deg=2,
    n_samples=5000,
    reg=0.05,
    numItermax=300,
    stopThr=1e-6,
    seed=0
):
    """
    Fit per-channel polynomial mapping y = poly(x) using OT barycentric targets.
    src_rgb, ref_rgb: (H,W,3) float in [0,1]
    mask: (H,W) boolean
    Returns coeffs: (3, deg+1) poly coefficients (highest power first) for R,G,B.
    """
    def identity():                                 # soft fallback of poly_regression.py:38-41
        coeffs = np.zeros((3, deg + 1), dtype=np.float64)
        coeffs[:, -2] = 1.0
        return coeffs

    if not (_is_torch(src_rgb) or _is_torch(ref_rgb) or _is_torch(mask)):
        # small masks are settled on the host (no device needed for the fallback; a few thousand rows cost nothing)
        m = np.asarray(mask, dtype=bool)
        cnt = int(m.sum())
        if cnt < 200:
            return identity()
        if cnt <= 65536:
            nx = int(np.isfinite(np.asarray(src_rgb)[m].reshape(-1, 3)).all(axis=1).sum())
            ny = int(np.isfinite(np.asarray(ref_rgb)[m].reshape(-1, 3)).all(axis=1).sum())
            if nx < 200 or ny < 200:
                return identity()
    torch = nat.require_gpu()

    def dev(a, dtype=None):
        t = a.detach() if _is_torch(a) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to("cuda") if dtype is None else t.to("cuda", dtype)

    # sampling on the device: only the two row counts and the drawn indices cross PCIe (see _ot.sample_pairs_device)
    s = _ot.sample_pairs_device(dev(src_rgb), dev(ref_rgb), dev(mask, torch.bool), n_samples, seed, min_rows=200)
    if s is None:
        return identity()
    Xd, Yd = s
    Ybar = _ot.barycentric_targets_device(Xd, Yd, reg, numItermax, stopThr, poll_every=50)   # the result is read on the host next
    return polyfit_columns(Xd, Ybar, deg)


def apply_poly_rgb(rgb, coeffs, mask=None):
    """
    Apply per-channel polynomial mapping to RGB image in [0,1].
    coeffs: (C, deg+1) poly coefficients from np.polyfit (highest power first).
    NumPy in -> NumPy float32 out; torch GPU tensor in -> torch GPU float32 tensor out.
    """
    torch = nat.require_gpu()
    as_torch = _is_torch(rgb)
    co = np.ascontiguousarray(coeffs.detach().cpu().numpy() if _is_torch(coeffs) else coeffs, dtype=np.float64)
    if co.ndim != 2:
        raise ValueError(f"coeffs must be (C, deg+1). Got shape {co.shape}")
    C_, n = co.shape
    if rgb.ndim != 3 or rgb.shape[-1] < C_:
        raise IndexError(f"rgb must be (H,W,C>={C_}). Got shape {tuple(rgb.shape)}")
    if n - 1 > nat.HSR_MAX_APPLY_DEG:
        raise ValueError(f"polynomial degree {n - 1} exceeds HSR_MAX_APPLY_DEG={nat.HSR_MAX_APPLY_DEG}")
    H, W, Cc = (int(v) for v in rgb.shape)
    if as_torch:
        x = rgb.to(device="cuda", dtype=torch.float32).contiguous()
    else:
        x = torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.float32)).cuda()
    m = None
    if mask is not None:
        if _is_torch(mask):
            m = mask.to(device=x.device, dtype=torch.uint8).contiguous().reshape(-1)
        else:
            m = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.bool_).view(np.uint8)).cuda().reshape(-1)
        if int(m.numel()) != H * W:
            raise IndexError("boolean index did not match indexed array: mask must be (H,W)")
    if Cc != C_:      # the reference touches only the first len(coeffs)==3 channels; the rest are clipped
        full = np.zeros((Cc, n), dtype=np.float64)
        full[:, -2 if n >= 2 else -1] = 1.0
        full[:C_] = co
        co = full
        # identity rows evaluate to x exactly for finite x; non-finite x would differ from a
        # pass-through (0*inf), so those channels are restored from the input below.
    cd = torch.from_numpy(co).cuda()
    if H * W == 0:
        out = x.clone()
    else:
        out = eng.poly_apply(x.reshape(-1, Cc), cd, m, None, True, nat.PIXMAJOR).reshape(H, W, Cc)
        if Cc != C_:
            out[..., C_:] = torch.clamp(x[..., C_:], 0.0, 1.0)
    return out if as_torch else out.cpu().numpy()
