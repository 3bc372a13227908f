"""Percentile stretches and colour matching.  Mirrors reference ``s2_emit/color.py``.

On the measured path (reference poly_regression.py:126-127,161) is
``apply_shared_percentile_stretch``: exact masked percentiles by a device radix select
(csrc/hsr_select.hip) and the stretch itself fused into K3 (csrc/hsr_poly.hip).  The remaining
names are API surface kept for drop-in use: plain host NumPy, no kernels.
"""
from __future__ import annotations

import numpy as np

from . import _engine as eng
from . import _native as nat


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def robust_norm(x: np.ndarray, pmin: float = 2, pmax: float = 98) -> np.ndarray:
    """Whole-array NaN-aware percentile stretch (color.py:6-8).  Host NumPy (API surface)."""
    lo, hi = np.nanpercentile(x, [pmin, pmax])
    return np.clip((x - lo) / (hi - lo + 1e-12), 0, 1)


def robust_norm_rgb(img: np.ndarray, mask: np.ndarray, pmin: float = 2, pmax: float = 98) -> np.ndarray:
    """Per-channel percentile stretch with limits from the masked pixels; float64 output with NaN outside the
    mask (color.py:10-23).  img (H,W,3), mask (H,W) bool.  Host NumPy (API surface)."""
    inside = np.asarray(mask, dtype=bool)
    limits = np.percentile(img[inside][:, :3], [pmin, pmax], axis=0)          # (2, 3): lo row, hi row
    scaled = (np.asarray(img[..., :3], dtype=float) - limits[0]) / (limits[1] - limits[0] + 1e-12)
    scaled[~inside] = np.nan
    return np.clip(scaled, 0, 1)


def device_percentile_stretch(x, mask=None, pmin=2, pmax=98, layout=nat.PIXMAJOR, lohi=None, nb=None,
                              group=None, distributed=None):
    """Device form: x float32 (npix, C) pixel-major or (C, npix) planar, mask uint8 (npix,).
    Returns (stretched float32 tensor, lohi (C, 2) float64 tensor); no host synchronisation.
    Inside an initialised torch.distributed job of more than one rank the limits are the exact
    percentiles of the union of all ranks' masked samples (histogram all-reduce per select pass,
    see _engine.percentile_limits); pass distributed=False for per-rank limits."""
    if lohi is None:
        lohi = eng.percentile_limits(x, mask, pmin, pmax, layout, nb, group=group, distributed=distributed)
    return eng.poly_apply_stretch_only(x, lohi, layout, nb=nb), lohi


def apply_shared_percentile_stretch(img, mask, pmin: float = 2, pmax: float = 98):
    """
    Per-channel (first 3 channels, as the reference) percentile stretch with limits taken inside
    ``mask`` and applied to every pixel; float32 in [0,1] (color.py:25-34).
    NumPy in -> NumPy out; torch GPU tensor in -> torch GPU tensor out.
    """
    torch = nat.require_gpu()
    as_torch = _is_torch(img)
    if img.ndim != 3:
        raise ValueError(f"img must be (H,W,C). Got shape {tuple(img.shape)}")
    H, W, Cc = (int(s) for s in img.shape)
    if Cc < 3:
        raise IndexError(f"index 2 is out of bounds for axis 2 with size {Cc}")
    if as_torch:
        x = img.to(dtype=torch.float32).contiguous()
        m = mask.to(device=x.device, dtype=torch.uint8).contiguous().reshape(-1)
    else:
        x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda()
        m = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.bool_).view(np.uint8)).cuda().reshape(-1)
    if Cc == 3:
        x3 = x.reshape(-1, 3)
    else:
        x3 = x[..., :3].contiguous().reshape(-1, 3)
    if int(m.numel()) != H * W:
        raise IndexError("boolean index did not match indexed array: mask must be (H,W)")
    y3, _ = device_percentile_stretch(x3, m, pmin, pmax, nat.PIXMAJOR)
    if Cc == 3:
        out = y3.reshape(H, W, 3)
    else:
        out = torch.zeros((H, W, Cc), dtype=torch.float32, device=x.device)
        out[..., :3] = y3.reshape(H, W, 3)
    return out if as_torch else out.cpu().numpy()


def _value_cdf(values: np.ndarray):
    """Distinct sorted values of a 1-D sample and their empirical CDF n(<= v) / (n + 1e-32)."""
    order = np.sort(values, kind="stable")
    last = np.flatnonzero(np.append(order[1:] != order[:-1], True))           # last index of every run
    return order[last], (last + 1).astype(np.float64) / (order.size + 1e-32)


def _hist_match_channel(src: np.ndarray, ref: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """CDF matching of one channel inside the mask (color.py:36-53): every masked source value goes to the
    reference value at the same cumulative frequency (np.interp between the reference's distinct values)."""
    picked = src[mask].ravel()
    s_vals, s_cdf = _value_cdf(picked)
    r_vals, r_cdf = _value_cdf(ref[mask].ravel())
    lut = np.interp(s_cdf, r_cdf, r_vals)
    out = src.copy()
    out[mask] = lut[np.searchsorted(s_vals, picked)]
    return out


def histogram_match_rgb(src_rgb: np.ndarray, ref_rgb: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """
    Histogram-match each channel independently within mask (color.py:55-63).
    Inputs assumed in [0,1].  Host NumPy (API surface).
    """
    out = src_rgb.copy()
    for c in range(3):
        out[..., c] = _hist_match_channel(out[..., c], ref_rgb[..., c], mask)
    return np.clip(out, 0, 1)


def ot_match_rgb_sinkhorn_pot(
    src_rgb: np.ndarray,
    ref_rgb: np.ndarray,
    mask: np.ndarray,
    n_samples: int = 5_000,
    reg: float = 0.05,
    numItermax: int = 300,
    stopThr: float = 1e-6,
    seed: int = 0,
) -> np.ndarray:
    """
    3D colour transfer: Sinkhorn OT on masked RGB samples, barycentric targets, affine least
    squares, applied inside the mask (color.py:65-116).  Sampling follows the reference's PCG64
    stream on the host; the Sinkhorn iterations run on the GPU in float64 (see ._ot).
    """
    from . import _ot
    s = _ot.sample_pairs(src_rgb, ref_rgb, mask, n_samples, seed, min_rows=2)
    if s is None:
        return src_rgb.copy()
    X, Y = s
    Ybar = _ot.barycentric_targets(X, Y, reg, numItermax, stopThr)
    X_aug = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1)
    Wm, *_ = np.linalg.lstsq(X_aug, Ybar, rcond=None)
    A, t = Wm[:3, :], Wm[3, :]
    out = src_rgb.copy().astype(np.float32)
    Xm = out[mask].reshape(-1, 3).astype(np.float64)
    out[mask] = np.clip(Xm @ A + t, 0.0, 1.0).reshape(out[mask].shape).astype(np.float32)
    return out
