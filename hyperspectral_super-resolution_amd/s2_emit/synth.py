"""SRF-weighted band synthesis on MI355X.  Mirrors reference ``s2_emit/synth.py``.

``pseudo_s2_srf_integral`` keeps the reference signature, return type and exceptions
(synth.py:9-45); the 13 full-cube float64 passes of the reference are replaced by one streaming pass
of the HIP kernel ``hsr_srf_integrate`` (csrc/hsr_srf.hip).  There is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from . import _engine as eng
from . import _native as nat


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def pseudo_s2_srf_integral(
    R,
    emit_w: np.ndarray,
    srf_dict: Dict[str, Tuple[np.ndarray, np.ndarray]],
    good_mask: Optional[np.ndarray] = None,
    device=None,
) -> Dict[str, Optional[np.ndarray]]:
    """
    SRF-weighted band synthesis.

    R: (H, W, B) reflectance - NumPy array (any float dtype; computed in float32) or a float32
       torch tensor already resident on the GPU (zero-copy)
    emit_w: (B,) wavelengths (nm)
    good_mask: (B,) boolean, optional

    returns: band -> (H, W) float64 ndarray, or None for a band without SRF support on the
    (masked) EMIT grid - exactly the reference's dict, in srf_dict order.  For a torch input the
    values are float32 device tensors (views of one (nb, H, W) allocation).
    """
    emit_w = np.asarray(emit_w.detach().cpu() if _is_torch(emit_w) else emit_w).astype(float)
    if R.ndim != 3:
        raise ValueError(f"R must be (H,W,B). Got shape {tuple(R.shape)}")
    if emit_w.ndim != 1 or emit_w.shape[0] != R.shape[-1]:
        raise ValueError(f"emit_w must be (B,) matching R bands. Got {emit_w.shape} vs {R.shape[-1]}")
    if good_mask is not None and _is_torch(good_mask):
        good_mask = good_mask.detach().cpu().numpy()

    table = eng.build_srf_table(emit_w, srf_dict, good_mask)
    out: Dict[str, Optional[np.ndarray]] = {band: None for band in table.names}
    if table.nb == 0:
        return out

    torch = nat.require_gpu()
    H, W = int(R.shape[0]), int(R.shape[1])
    as_torch = _is_torch(R)
    if as_torch:
        cube = R if (R.is_cuda and R.dtype == torch.float32 and R.is_contiguous()) else \
            R.to(device=device or "cuda", dtype=torch.float32).contiguous()
    else:
        cube = torch.from_numpy(np.ascontiguousarray(R, dtype=np.float32)).to(device or "cuda")
    if H * W == 0:
        planes = torch.empty((table.nb, 0), dtype=torch.float32, device=cube.device)
    else:
        planes = eng.srf_integrate(cube, table, layout=nat.PLANAR)
    planes = planes.reshape(table.nb, H, W)
    if as_torch:
        for i, band in enumerate(table.supported):
            out[band] = planes[i]
    else:
        host = planes.cpu().numpy().astype(np.float64)
        for i, band in enumerate(table.supported):
            out[band] = host[i]
    return out


def pseudo_s2_rgb(pseudo_s2: Dict[str, Optional[np.ndarray]], order=("B4", "B3", "B2")):
    """
    Builds RGB stack from pseudo_s2 dict.
    Returns (H, W, 3)
    """
    chans = []
    for b in order:
        x = pseudo_s2.get(b, None)
        if x is None:
            raise ValueError(f"Band {b} is None/missing in pseudo_s2.")
        chans.append(x)
    if _is_torch(chans[0]):
        import torch
        return torch.stack(chans, dim=-1)
    return np.stack(chans, axis=-1)


def crop_to_overlap(s2_path, emit_path, out_s2_path, out_emit_path):
    """Raster I/O helper of the reference (synth.py:61-139): out of scope of the accelerated path."""
    raise NotImplementedError(
        "crop_to_overlap is GDAL/rasterio raster I/O (reference s2_emit/synth.py:61-139) and is outside "
        "the MI355X hot path; crop with rasterio/gdal_translate upstream and pass arrays to s2_emit.")
