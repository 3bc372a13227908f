"""Device-level operators of the s2_emit hot path: torch tensors in, torch tensors out.

PyTorch is used only as plumbing (HBM allocations, the current HIP stream, torch.distributed);
every operator here is one or a few stream-ordered calls into libhsr_mi355x.so through its C ABI
(include/hsr.h).  Nothing in this module synchronises the device.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat

_NEG_INF = float("-inf")


def _stream(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---------------------------------------------------------------------------------------------
# host-side SRF weight table (reference s2_emit/synth.py:25,33-43 evaluated once, in float64)
# ---------------------------------------------------------------------------------------------
@dataclass
class SrfTable:
    """Normalised trapezoid weights Wn[b, k] so that pseudo_s2[b] = sum_k R[..., k] * Wn[b, k].

    ``names``      every key of srf_dict in insertion order (synth.py:32 keeps that order)
    ``supported``  keys whose resampled SRF is not identically zero (others -> None, synth.py:37-39)
    ``weights``    (len(supported), B) float64
    ``k0/klen``    support [k0, k0+klen) of each row (first..last non-zero weight)
    """
    names: List[str]
    supported: List[str]
    weights: np.ndarray
    k0: np.ndarray
    klen: np.ndarray
    B: int
    _dev: Dict[str, object] = field(default_factory=dict, repr=False)

    @property
    def nb(self) -> int:
        return len(self.supported)

    def device_weights(self, device):
        import torch
        key = str(device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self.weights.astype(np.float32)).to(device).contiguous()
        return self._dev[key]


def build_srf_table(emit_w, srf_dict, good_mask=None) -> SrfTable:
    """np.interp of every SRF on the EMIT grid, good_mask, all-zero test, trapezoid weights and the
    ``den + 1e-32`` normalisation - the same float64 expressions as synth.py:33-43, with the
    pixel-independent part of np.trapz folded into one weight per wavelength:
    trapz(R*r, w) = sum_k R_k * r_k * (d_{k-1} + d_k)/2,  d_k = w_{k+1} - w_k."""
    w = np.asarray(emit_w).astype(float)
    if w.ndim != 1:
        raise ValueError(f"emit_w must be (B,) matching R bands. Got {w.shape} vs ?")
    B = w.shape[0]
    gm = None if good_mask is None else np.asarray(good_mask).astype(float)
    d = np.diff(w)
    half = (np.concatenate([[0.0], d]) + np.concatenate([d, [0.0]])) / 2.0
    names, supported, rows, k0s, kls = [], [], [], [], []
    for band, (lam, rsp) in srf_dict.items():
        names.append(band)
        r = np.interp(w, lam, rsp, left=0.0, right=0.0)
        if gm is not None:
            r = r * gm
        if np.all(r == 0):
            continue
        den = (d * (r[1:] + r[:-1]) / 2.0).sum() if B > 1 else 0.0      # np.trapz(r, x=w)
        row = (r * half) / (den + 1e-32)
        nz = np.nonzero(row)[0]
        supported.append(band)
        rows.append(row)
        if nz.size:
            k0s.append(int(nz[0]))
            kls.append(int(nz[-1] - nz[0] + 1))
        else:
            k0s.append(0)
            kls.append(0)
    W = np.asarray(rows, dtype=np.float64).reshape(len(rows), B)
    return SrfTable(names, supported, W, np.asarray(k0s, np.int32), np.asarray(kls, np.int32), B)


# ---------------------------------------------------------------------------------------------
# K1 / K1+K2
# ---------------------------------------------------------------------------------------------
def _as_cube2d(cube):
    torch = nat.require_gpu()
    if not (cube.is_cuda and cube.dtype == torch.float32 and cube.is_contiguous()):
        raise ValueError("cube must be a contiguous float32 tensor on the GPU")
    return cube.reshape(-1, cube.shape[-1])


def _i32arr(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def srf_integrate(cube, table: SrfTable, out=None):
    """K1.  cube (..., B) float32 on the GPU -> planes (nb, npix) float32 (band-major)."""
    torch = nat.require_gpu()
    lib = nat.load()
    c2 = _as_cube2d(cube)
    npix, B = c2.shape
    if B != table.B:
        raise ValueError(f"emit_w must be (B,) matching R bands. Got {(table.B,)} vs {B}")
    nb = table.nb
    planes = out if out is not None else torch.empty((nb, npix), dtype=torch.float32, device=cube.device)
    wn = table.device_weights(cube.device)
    for b0 in range(0, nb, nat.HSR_MAX_BANDS):          # >16 bands: one more pass over the cube
        b1 = min(nb, b0 + nat.HSR_MAX_BANDS)
        k0a, k0p = _i32arr(table.k0[b0:b1])      # keep the arrays alive across the call
        kla, klp = _i32arr(table.klen[b0:b1])
        nat.check(lib.hsr_srf_integrate(_ptr(c2), npix, B, _ptr(wn[b0:b1]), k0p, klp, b1 - b0,
                                        _ptr(planes[b0:b1]), planes.stride(0), _stream(torch)),
                  "hsr_srf_integrate")
    return planes


class MomentWorkspace:
    """Per-device scratch for the partial sums and the reduced moments (allocated once)."""

    def __init__(self, device, nb: int, deg: int):
        torch = nat.require_gpu()
        lib = nat.load()
        self.nb, self.deg = nb, deg
        self.M = 3 * deg + 2
        nbytes = lib.hsr_partials_bytes(nb, deg)
        self.partials = torch.empty(nbytes // 8, dtype=torch.float64, device=device)
        self.moments = torch.zeros((nb, self.M), dtype=torch.float64, device=device)
        self.coeffs = torch.zeros((nb, deg + 1), dtype=torch.float64, device=device)


def srf_integrate_moments(cube, table: SrfTable, real_planes, deg: int, ws: MomentWorkspace,
                          mask=None, min_x=_NEG_INF, min_y=_NEG_INF, out=None, events=None):
    """K1+K2 fused: planes and the per-band Vandermonde moments (nb, 3deg+2) in one cube pass.
    ``events``: optional (start, stop) torch.cuda.Event pair recorded on the launch stream right
    around the fused kernel (bench.py's live roofline measurement)."""
    torch = nat.require_gpu()
    lib = nat.load()
    c2 = _as_cube2d(cube)
    npix, B = c2.shape
    nb = table.nb
    if nb > nat.HSR_MAX_BANDS:
        raise ValueError(f"fused SRF+moments handles at most {nat.HSR_MAX_BANDS} bands per call")
    if B != table.B:
        raise ValueError(f"emit_w must be (B,) matching R bands. Got {(table.B,)} vs {B}")
    real2 = real_planes.reshape(nb, -1)
    if not (real2.dtype == torch.float32 and real2.is_cuda and real2.stride(1) == 1 and real2.shape[1] == npix):
        raise ValueError("real_planes must be (nb, npix) float32 on the GPU")
    if mask is not None and not (mask.dtype == torch.uint8 and mask.numel() == npix and mask.is_contiguous()):
        raise ValueError("mask must be a contiguous uint8 tensor with one byte per pixel")
    planes = out if out is not None else torch.empty((nb, npix), dtype=torch.float32, device=cube.device)
    wn = table.device_weights(cube.device)
    k0a, k0p = _i32arr(table.k0)                 # keep the arrays alive across the call
    kla, klp = _i32arr(table.klen)
    slots = C.c_int32(0)
    if events is not None:
        events[0].record()
    nat.check(lib.hsr_srf_integrate_moments(_ptr(c2), npix, B, _ptr(wn), k0p, klp, nb, _ptr(planes),
                                            planes.stride(0), _ptr(real2), real2.stride(0), _ptr(mask),
                                            min_x, min_y, deg, _ptr(ws.partials), C.byref(slots),
                                            _stream(torch)), "hsr_srf_integrate_moments")
    if events is not None:
        events[1].record()
    nat.check(lib.hsr_moments_reduce(_ptr(ws.partials), slots.value, nb, deg, _ptr(ws.moments),
                                     _stream(torch)), "hsr_moments_reduce")
    return planes, ws.moments


def poly_moments(x, y, deg: int, ws: MomentWorkspace, mask=None, min_x=_NEG_INF, min_y=_NEG_INF,
                 lohi_x=None, lohi_y=None):
    """K2 on materialised (nb, npix) float32 planes -> moments (nb, 3deg+2) float64 on the device."""
    torch = nat.require_gpu()
    lib = nat.load()
    nb, npix = x.shape
    for t in (x, y):
        if not (t.dtype == torch.float32 and t.is_cuda and t.stride(1) == 1 and tuple(t.shape) == (nb, npix)):
            raise ValueError("x and y must be (nb, npix) float32 on the GPU")
    slots = C.c_int32(0)
    nat.check(lib.hsr_poly_moments(_ptr(x), x.stride(0), _ptr(y), y.stride(0), _ptr(mask), npix, nb, deg,
                                   min_x, min_y, _ptr(lohi_x), _ptr(lohi_y), _ptr(ws.partials),
                                   C.byref(slots), _stream(torch)), "hsr_poly_moments")
    nat.check(lib.hsr_moments_reduce(_ptr(ws.partials), slots.value, nb, deg, _ptr(ws.moments),
                                     _stream(torch)), "hsr_moments_reduce")
    return ws.moments


def poly_moments_f64(x, y, deg: int, ws: MomentWorkspace):
    """K2 for float64 sample columns: x, y (nb, n) float64 on the GPU -> moments (nb, 3deg+2)."""
    torch = nat.require_gpu()
    lib = nat.load()
    nb, npix = x.shape
    for t in (x, y):
        if not (t.dtype == torch.float64 and t.is_cuda and t.stride(1) == 1 and tuple(t.shape) == (nb, npix)):
            raise ValueError("x and y must be (nb, n) float64 on the GPU")
    slots = C.c_int32(0)
    nat.check(lib.hsr_poly_moments_f64(_ptr(x), x.stride(0), _ptr(y), y.stride(0), npix, nb, deg,
                                       _ptr(ws.partials), C.byref(slots), _stream(torch)), "hsr_poly_moments_f64")
    nat.check(lib.hsr_moments_reduce(_ptr(ws.partials), slots.value, nb, deg, _ptr(ws.moments),
                                     _stream(torch)), "hsr_moments_reduce")
    return ws.moments


def poly_solve(moments, deg: int, min_count: int, out=None):
    """np.polyfit from moments, on the device, stream ordered.  (nb, deg+1) float64, highest first."""
    torch = nat.require_gpu()
    lib = nat.load()
    nb = moments.shape[0]
    coeffs = out if out is not None else torch.empty((nb, deg + 1), dtype=torch.float64, device=moments.device)
    nat.check(lib.hsr_poly_solve(_ptr(moments), nb, deg, int(min_count), _ptr(coeffs), _stream(torch)),
              "hsr_poly_solve")
    return coeffs


def poly_solve_host(moments: np.ndarray, deg: int, min_count: int) -> np.ndarray:
    """Host twin of poly_solve (same C code compiled for the CPU); needs only the library."""
    lib = nat.load()
    m = np.ascontiguousarray(moments, dtype=np.float64)
    nb = m.shape[0]
    out = np.zeros((nb, deg + 1), dtype=np.float64)
    nat.check(lib.hsr_poly_solve_host(m.ctypes.data_as(C.POINTER(C.c_double)), nb, deg, int(min_count),
                                      out.ctypes.data_as(C.POINTER(C.c_double))), "hsr_poly_solve_host")
    return out


def poly_apply(x, coeffs, mask=None, lohi=None, clip=True, layout=nat.LAYOUT_PLANAR, out=None):
    """K3.  planar: x (nb, npix); interleaved: x (npix, nb).  float32 in, float32 out."""
    torch = nat.require_gpu()
    lib = nat.load()
    if layout == nat.LAYOUT_PLANAR:
        nb, npix = x.shape
        if x.stride(1) != 1:
            raise ValueError("planar x must have unit pixel stride")
        xs = x.stride(0)
    else:
        npix, nb = x.shape
        if not x.is_contiguous():
            raise ValueError("interleaved x must be contiguous")
        xs = 0
    deg = coeffs.shape[1] - 1
    if coeffs.shape[0] != nb or coeffs.dtype != torch.float64 or not coeffs.is_contiguous():
        raise ValueError("coeffs must be a contiguous (nb, deg+1) float64 tensor")
    o = out if out is not None else torch.empty_like(x, memory_format=torch.contiguous_format)
    os_ = o.stride(0) if layout == nat.LAYOUT_PLANAR else 0
    nat.check(lib.hsr_poly_apply(_ptr(x), xs, _ptr(mask), _ptr(coeffs), nb, deg, npix, _ptr(lohi),
                                 1 if clip else 0, layout, _ptr(o), os_, _stream(torch)), "hsr_poly_apply")
    return o


def poly_apply_stretch_only(x, lohi, layout=nat.LAYOUT_PLANAR, out=None):
    """float32(clip((x - lo)/(hi - lo + 1e-12), 0, 1)) per channel: K3 without a polynomial."""
    torch = nat.require_gpu()
    lib = nat.load()
    if layout == nat.LAYOUT_PLANAR:
        nb, npix = x.shape
        xs = x.stride(0)
    else:
        npix, nb = x.shape
        xs = 0
    o = out if out is not None else torch.empty_like(x, memory_format=torch.contiguous_format)
    os_ = o.stride(0) if layout == nat.LAYOUT_PLANAR else 0
    nat.check(lib.hsr_poly_apply(_ptr(x), xs, None, None, nb, 0, npix, _ptr(lohi), 1, layout, _ptr(o), os_,
                                 _stream(torch)), "hsr_poly_apply(stretch)")
    return o


def percentile_limits(x, mask=None, pmin=2.0, pmax=98.0, layout=nat.LAYOUT_PLANAR):
    """Exact np.percentile(vals[mask], [pmin, pmax]) per channel -> (nb, 2) float64 on the device."""
    torch = nat.require_gpu()
    lib = nat.load()
    if layout == nat.LAYOUT_PLANAR:
        nb, npix = x.shape
        xs = x.stride(0)
    else:
        npix, nb = x.shape
        xs = 0
    work = torch.empty(lib.hsr_percentile_work_bytes(nb) // 8 + 1, dtype=torch.int64, device=x.device)
    lohi = torch.empty((nb, 2), dtype=torch.float64, device=x.device)
    nat.check(lib.hsr_percentile_limits(_ptr(x), xs, layout, _ptr(mask), npix, nb, float(pmin), float(pmax),
                                        _ptr(work), _ptr(lohi), _stream(torch)), "hsr_percentile_limits")
    return lohi


def valid_mask(x, pos_band: int = -1, y=None, mask_in=None):
    """mask[p] = all x bands finite && x[pos_band] > 0 && all y bands finite (poly_regression.py:106,118)."""
    torch = nat.require_gpu()
    lib = nat.load()
    nbx, npix = x.shape
    out = torch.empty(npix, dtype=torch.uint8, device=x.device)
    nat.check(lib.hsr_valid_mask(_ptr(x), x.stride(0), nbx, pos_band, _ptr(y), y.stride(0) if y is not None else 0,
                                 y.shape[0] if y is not None else 0, _ptr(mask_in), npix, _ptr(out),
                                 _stream(torch)), "hsr_valid_mask")
    return out


def probe_read_bandwidth(nbytes: int = 1 << 30, iters: int = 10, device="cuda:0") -> float:
    """Measured pure-read HBM rate of this box in bytes/s (diagnostic for the roofline report)."""
    torch = nat.require_gpu()
    lib = nat.load()
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    sink = torch.zeros(64, dtype=torch.float32, device=device)
    for _ in range(2):
        nat.check(lib.hsr_probe_read(_ptr(buf), nbytes, _ptr(sink), _stream(torch)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        nat.check(lib.hsr_probe_read(_ptr(buf), nbytes, _ptr(sink), _stream(torch)))
    e1.record()
    e1.synchronize()
    return nbytes * iters / (e0.elapsed_time(e1) * 1e-3)
