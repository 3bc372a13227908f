"""Device-level operators of the s2_emit hot path: torch tensors in, torch tensors out.

PyTorch is used only as plumbing (HBM allocations, the current HIP stream, torch.distributed);
every operator here is one or a few stream-ordered calls into libhsr_mi355x.so through its C ABI
(include/hsr.h).  Nothing in this module synchronises the device.

Image-like tensors come in two layouts (include/hsr.h, "band_stride / pixel_stride"):
  PLANAR    (nb, npix)   band-major planes
  PIXMAJOR  (npix, row)  pixel-major / band-last, row >= nb (rows padded to a multiple of 4 floats
                         take the vectorised paths); the layout of the cube and of (H, W, C) images.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _native as nat
from ._native import PIXMAJOR, PLANAR

_NEG_INF = float("-inf")


def _stream(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def padded_row(nb: int) -> int:
    """Row length (floats) of a pixel-major image with nb channels: next multiple of 4."""
    return (nb + 3) // 4 * 4


def _img(t, layout: str, nb: Optional[int] = None):
    """(band_stride, pixel_stride, nb, npix) of a 2-D float32 image tensor in the given layout."""
    if t.dim() != 2:
        raise ValueError(f"image tensors are 2-D ((nb, npix) or (npix, row)); got shape {tuple(t.shape)}")
    if layout == PLANAR:
        n, npix = t.shape
        if npix > 1 and t.stride(1) != 1:
            raise ValueError("planar image must have unit pixel stride")
        bs = t.stride(0) if n > 1 else max(int(t.stride(0)), int(npix))
        return int(bs), 1, int(nb or n), int(npix)
    if layout == PIXMAJOR:
        npix, row = t.shape
        if row > 1 and t.stride(1) != 1:
            raise ValueError("pixel-major image must have unit band stride")
        ps = t.stride(0) if npix > 1 else max(int(t.stride(0)), int(row))
        n = int(nb or row)
        if n > row:
            raise ValueError(f"pixel-major image has rows of {row} < nb={n}")
        return 1, int(ps), n, int(npix)
    raise ValueError(f"unknown layout {layout!r}")


def alloc_image(torch, nb: int, npix: int, layout: str, device):
    if layout == PLANAR:
        return torch.empty((nb, npix), dtype=torch.float32, device=device)
    return torch.empty((npix, padded_row(nb)), dtype=torch.float32, device=device)


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
    if not (cube.is_cuda and cube.dtype in (torch.float32, torch.uint16) and cube.is_contiguous()):
        raise ValueError("cube must be a contiguous float32 (or uint16 tile) tensor on the GPU")
    return cube.reshape(-1, cube.shape[-1])


# uint16 tiles (reference writer tiles_helpers/utils.py:309-318,362-374)
TILE_SCALE = 10000.0         # emit_scale: reflectance -> uint16
TILE_NODATA = 65535          # emit_nodata_u16


def _decode_scale(scale) -> float:
    """Decode factor as the consumers use it: float32(1/emit_scale) = the notebook's s2_scale=1e-4."""
    return float(np.float32(1.0 / TILE_SCALE)) if scale is None else float(np.float32(scale))


def tile_encode_u16(x, scale: float = TILE_SCALE, src_nodata=None, nodata_u16: int = TILE_NODATA):
    """float32 GPU tensor (any shape) -> uint16 tile samples exactly as the reference's tile writer
    quantises them (tiles_helpers/utils.py:362-374): round-half-even of the float32 product, clip to
    [0, nodata_u16-1], non-finite / source-nodata samples -> nodata_u16."""
    torch = nat.require_gpu()
    lib = nat.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("x must be a contiguous float32 tensor on the GPU")
    out = torch.empty(x.shape, dtype=torch.uint16, device=x.device)
    nat.check(lib.hsr_tile_encode_u16(_ptr(x), x.numel(), float(scale), int(src_nodata is not None),
                                      float(src_nodata) if src_nodata is not None else 0.0, int(nodata_u16),
                                      _ptr(out), _stream(torch)), "hsr_tile_encode_u16")
    return out


def tile_decode_u16(u, scale=None, nodata: Optional[int] = TILE_NODATA):
    """uint16 GPU tensor -> float32: float32(u) * float32(scale) (default 1e-4), nodata -> NaN."""
    torch = nat.require_gpu()
    lib = nat.load()
    if not (u.is_cuda and u.dtype == torch.uint16 and u.is_contiguous()):
        raise ValueError("u must be a contiguous uint16 tensor on the GPU")
    out = torch.empty(u.shape, dtype=torch.float32, device=u.device)
    nat.check(lib.hsr_tile_decode_u16(_ptr(u), u.numel(), _decode_scale(scale), -1 if nodata is None else int(nodata),
                                      _ptr(out), _stream(torch)), "hsr_tile_decode_u16")
    return out


def _i32arr(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def srf_integrate(cube, table: SrfTable, out=None, layout: str = PLANAR, scale=None, nodata: Optional[int] = TILE_NODATA):
    """K1.  cube (..., B) float32 on the GPU -> pseudo-S2 image in ``layout`` (float32).
    A uint16 cube is taken as a tile in the reference's storage format and decoded inside the kernel
    (``scale`` default 1e-4, ``nodata`` default 65535, None = no nodata value)."""
    torch = nat.require_gpu()
    lib = nat.load()
    c2 = _as_cube2d(cube)
    npix, B = c2.shape
    if B != table.B:
        raise ValueError(f"emit_w must be (B,) matching R bands. Got {(table.B,)} vs {B}")
    nb = table.nb
    img = out if out is not None else alloc_image(torch, nb, npix, layout, cube.device)
    bs, ps, _, _ = _img(img, layout, nb)
    wn = table.device_weights(cube.device)
    for b0 in range(0, nb, nat.HSR_MAX_BANDS):          # >16 bands: one more pass over the cube
        b1 = min(nb, b0 + nat.HSR_MAX_BANDS)
        k0a, k0p = _i32arr(table.k0[b0:b1])              # keep the arrays alive across the call
        kla, klp = _i32arr(table.klen[b0:b1])
        dst = img[b0:b1] if layout == PLANAR else img[:, b0:]
        if c2.dtype == torch.uint16:
            nat.check(lib.hsr_srf_integrate_u16(_ptr(c2), npix, B, _decode_scale(scale), -1 if nodata is None else int(nodata),
                                                _ptr(wn[b0:b1]), k0p, klp, b1 - b0, _ptr(dst), bs, ps, _stream(torch)),
                      "hsr_srf_integrate_u16")
        else:
            nat.check(lib.hsr_srf_integrate(_ptr(c2), npix, B, _ptr(wn[b0:b1]), k0p, klp, b1 - b0,
                                            _ptr(dst), bs, ps, _stream(torch)), "hsr_srf_integrate")
    return img


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
        self.slots = 0


def srf_integrate_moments(cube, table: SrfTable, real, deg: int, ws: MomentWorkspace, mask=None,
                          min_x=_NEG_INF, min_y=_NEG_INF, out=None, events=None, reduce=True,
                          layout: str = PIXMAJOR, real_layout: Optional[str] = None, scale=None,
                          nodata: Optional[int] = TILE_NODATA):
    """K1+K2 fused: pseudo-S2 image and the per-band Vandermonde moments in one cube pass.
    A uint16 cube is decoded inside the kernel (see srf_integrate).
    ``real``: real-S2 image in ``real_layout`` (default: same as ``layout``).
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
    rbs, rps, _, rn = _img(real, real_layout or layout, nb)
    if rn != npix or real.dtype != torch.float32 or not real.is_cuda:
        raise ValueError("real must be a float32 GPU image with one value per pixel and band")
    if mask is not None and not (mask.dtype == torch.uint8 and mask.numel() == npix and mask.is_contiguous()):
        raise ValueError("mask must be a contiguous uint8 tensor with one byte per pixel")
    img = out if out is not None else alloc_image(torch, nb, npix, layout, cube.device)
    bs, ps, _, _ = _img(img, layout, nb)
    wn = table.device_weights(cube.device)
    k0a, k0p = _i32arr(table.k0)                 # keep the arrays alive across the call
    kla, klp = _i32arr(table.klen)
    slots = C.c_int32(0)
    if events is not None:
        events[0].record()
    if c2.dtype == torch.uint16:
        nat.check(lib.hsr_srf_integrate_moments_u16(_ptr(c2), npix, B, _decode_scale(scale),
                                                    -1 if nodata is None else int(nodata), _ptr(wn), k0p, klp, nb,
                                                    _ptr(img), bs, ps, _ptr(real), rbs, rps, _ptr(mask), min_x, min_y,
                                                    deg, _ptr(ws.partials), C.byref(slots), _stream(torch)),
                  "hsr_srf_integrate_moments_u16")
    else:
        nat.check(lib.hsr_srf_integrate_moments(_ptr(c2), npix, B, _ptr(wn), k0p, klp, nb, _ptr(img), bs, ps,
                                                _ptr(real), rbs, rps, _ptr(mask), min_x, min_y, deg,
                                                _ptr(ws.partials), C.byref(slots), _stream(torch)),
                  "hsr_srf_integrate_moments")
    if events is not None:
        events[1].record()
    ws.slots = slots.value
    if not reduce:                      # caller continues with moments_reduce_solve / moments_reduce
        return img, None
    return img, moments_reduce(ws)


def moments_reduce(ws: MomentWorkspace):
    """Fixed-order reduction of the partial slots of the last moments launch -> ws.moments."""
    torch = nat.require_gpu()
    lib = nat.load()
    nat.check(lib.hsr_moments_reduce(_ptr(ws.partials), ws.slots, ws.nb, ws.deg, _ptr(ws.moments),
                                     _stream(torch)), "hsr_moments_reduce")
    return ws.moments


def moments_reduce_solve(ws: MomentWorkspace, min_count: int):
    """Reduce + np.polyfit solve in one launch (no exchange in between) -> (ws.moments, ws.coeffs)."""
    torch = nat.require_gpu()
    lib = nat.load()
    nat.check(lib.hsr_moments_reduce_solve(_ptr(ws.partials), ws.slots, ws.nb, ws.deg, int(min_count),
                                           _ptr(ws.moments), _ptr(ws.coeffs), _stream(torch)),
              "hsr_moments_reduce_solve")
    return ws.moments, ws.coeffs


def poly_moments(x, y, deg: int, ws: MomentWorkspace, mask=None, min_x=_NEG_INF, min_y=_NEG_INF,
                 lohi_x=None, lohi_y=None, layout: str = PLANAR, nb: Optional[int] = None):
    """K2 on materialised float32 images -> moments (nb, 3deg+2) float64 on the device."""
    torch = nat.require_gpu()
    lib = nat.load()
    xbs, xps, n, npix = _img(x, layout, nb)
    ybs, yps, _, ny = _img(y, layout, n)
    if ny != npix or x.dtype != torch.float32 or y.dtype != torch.float32:
        raise ValueError("x and y must be float32 images of the same shape")
    slots = C.c_int32(0)
    nat.check(lib.hsr_poly_moments(_ptr(x), xbs, xps, _ptr(y), ybs, yps, _ptr(mask), npix, n, deg,
                                   min_x, min_y, _ptr(lohi_x), _ptr(lohi_y), _ptr(ws.partials),
                                   C.byref(slots), _stream(torch)), "hsr_poly_moments")
    ws.slots = slots.value
    return moments_reduce(ws)


def poly_moments_f64(x, y, deg: int, ws: MomentWorkspace):
    """K2 for float64 sample columns: x, y (nb, n) float64 on the GPU -> moments (nb, 3deg+2)."""
    torch = nat.require_gpu()
    lib = nat.load()
    nb, npix = x.shape
    for t in (x, y):
        if not (t.dtype == torch.float64 and t.is_cuda and (npix == 1 or t.stride(1) == 1) and tuple(t.shape) == (nb, npix)):
            raise ValueError("x and y must be (nb, n) float64 on the GPU")
    xs = x.stride(0) if nb > 1 else max(int(x.stride(0)), npix)
    ys = y.stride(0) if nb > 1 else max(int(y.stride(0)), npix)
    slots = C.c_int32(0)
    nat.check(lib.hsr_poly_moments_f64(_ptr(x), xs, _ptr(y), ys, npix, nb, deg,
                                       _ptr(ws.partials), C.byref(slots), _stream(torch)), "hsr_poly_moments_f64")
    ws.slots = slots.value
    return moments_reduce(ws)


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


def poly_apply(x, coeffs, mask=None, lohi=None, clip=True, layout: str = PLANAR, out=None,
               nb: Optional[int] = None):
    """K3.  float32 image in, float32 image out (same layout).  ``coeffs`` None: stretch/clip only."""
    torch = nat.require_gpu()
    lib = nat.load()
    xbs, xps, n, npix = _img(x, layout, nb)
    deg = 0
    if coeffs is not None:
        deg = coeffs.shape[1] - 1
        if coeffs.shape[0] != n or coeffs.dtype != torch.float64 or not coeffs.is_contiguous():
            raise ValueError("coeffs must be a contiguous (nb, deg+1) float64 tensor")
    o = out if out is not None else torch.empty_like(x, memory_format=torch.contiguous_format)
    obs, ops, _, _ = _img(o, layout, n)
    nat.check(lib.hsr_poly_apply(_ptr(x), xbs, xps, _ptr(mask), _ptr(coeffs), n, deg, npix, _ptr(lohi),
                                 1 if clip else 0, _ptr(o), obs, ops, _stream(torch)), "hsr_poly_apply")
    return o


def poly_apply_stretch_only(x, lohi, layout: str = PLANAR, out=None, nb: Optional[int] = None):
    """float32(clip((x - lo)/(hi - lo + 1e-12), 0, 1)) per channel: K3 without a polynomial."""
    return poly_apply(x, None, None, lohi, True, layout, out, nb)


def percentile_limits(x, mask=None, pmin=2.0, pmax=98.0, layout: str = PLANAR, nb: Optional[int] = None,
                      group=None, distributed: Optional[bool] = None, _reduce=None):
    """Exact np.percentile(vals[mask], [pmin, pmax]) per channel -> (nb, 2) float64 on the device.

    With a torch.distributed group of more than one rank (or distributed=True) the limits are GLOBAL: the
    integer histogram of each of the three radix-select passes is all-reduced (sum) between the ranks, so
    the order statistics are exact over the union of every rank's masked samples - the stretch limits of
    the reference pipeline (color.py:31-32) for a mosaic sharded over GPUs.  ``_reduce`` (tests) replaces
    the all-reduce by a callable acting on the int32 histogram view."""
    torch = nat.require_gpu()
    lib = nat.load()
    xbs, xps, n, npix = _img(x, layout, nb)
    work = torch.empty(lib.hsr_percentile_work_bytes(n) // 8 + 1, dtype=torch.int64, device=x.device)
    lohi = torch.empty((n, 2), dtype=torch.float64, device=x.device)
    if distributed is None:
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not distributed and _reduce is None:
        nat.check(lib.hsr_percentile_limits(_ptr(x), xbs, xps, _ptr(mask), npix, n, float(pmin), float(pmax),
                                            _ptr(work), _ptr(lohi), _stream(torch)), "hsr_percentile_limits")
        return lohi
    w32 = work.view(torch.int32)
    nat.check(lib.hsr_percentile_begin(_ptr(work), n, _stream(torch)), "hsr_percentile_begin")
    for p in (1, 2, 3):
        nat.check(lib.hsr_percentile_hist(p, _ptr(x), xbs, xps, _ptr(mask), npix, n, _ptr(work), _stream(torch)),
                  "hsr_percentile_hist")
        off, cnt = C.c_int64(0), C.c_int64(0)
        nat.check(lib.hsr_percentile_hist_region(p, n, C.byref(off), C.byref(cnt)), "hsr_percentile_hist_region")
        region = w32[off.value // 4: off.value // 4 + cnt.value]
        if _reduce is not None:
            _reduce(p, region)
        else:
            import torch.distributed as dist
            dist.all_reduce(region, op=dist.ReduceOp.SUM, group=group)
        nat.check(lib.hsr_percentile_scan(p, n, float(pmin), float(pmax), _ptr(work), _ptr(lohi), _stream(torch)),
                  "hsr_percentile_scan")
    return lohi


def valid_mask(x, pos_band: int = -1, y=None, mask_in=None, layout: str = PLANAR, nbx: Optional[int] = None,
               nby: Optional[int] = None):
    """mask[p] = all x bands finite && x[pos_band] > 0 && all y bands finite (poly_regression.py:106,118)."""
    torch = nat.require_gpu()
    lib = nat.load()
    xbs, xps, nx, npix = _img(x, layout, nbx)
    ybs = yps = ny = 0
    if y is not None:
        ybs, yps, ny, _ = _img(y, layout, nby)
    out = torch.empty(npix, dtype=torch.uint8, device=x.device)
    nat.check(lib.hsr_valid_mask(_ptr(x), xbs, xps, nx, pos_band, _ptr(y), ybs, yps, ny, _ptr(mask_in), npix,
                                 _ptr(out), _stream(torch)), "hsr_valid_mask")
    return out


def block_mean(fine, Hc: int, Wc: int, factor: int, scale: float = 1.0, layout: str = PLANAR,
               out_layout: Optional[str] = None, nb: Optional[int] = None):
    """'average' downsampling of an exactly aligned fine image (Hc*f x Wc*f pixels) by an integer factor.
    fine: float32 / uint8 / uint16 image tensor in ``layout``; returns float32 image in ``out_layout``."""
    torch = nat.require_gpu()
    lib = nat.load()
    dt = {torch.float32: 0, torch.uint8: 1, torch.uint16: 2}.get(fine.dtype)
    if dt is None:
        raise ValueError(f"block_mean: unsupported dtype {fine.dtype}")
    ibs, ips, n, npf = _img(fine, layout, nb)
    if npf != Hc * factor * Wc * factor:
        raise ValueError(f"block_mean: fine image has {npf} pixels, expected {Hc * factor}x{Wc * factor}")
    ol = out_layout or layout
    out = alloc_image(torch, n, Hc * Wc, ol, fine.device)
    obs, ops, _, _ = _img(out, ol, n)
    nat.check(lib.hsr_block_mean(_ptr(fine), dt, ibs, ips, n, Hc, Wc, factor, float(scale), _ptr(out), obs, ops,
                                 _stream(torch)), "hsr_block_mean")
    return out


def bilinear_upsample(coarse, Hc: int, Wc: int, factor: int, layout: str = PLANAR,
                      out_layout: Optional[str] = None, nb: Optional[int] = None):
    """Pixel-centre aligned bilinear upsampling by an integer factor (edge clamp)."""
    torch = nat.require_gpu()
    lib = nat.load()
    ibs, ips, n, npc = _img(coarse, layout, nb)
    if npc != Hc * Wc or coarse.dtype != torch.float32:
        raise ValueError("bilinear_upsample: coarse must be a float32 image with Hc*Wc pixels")
    ol = out_layout or layout
    out = alloc_image(torch, n, Hc * factor * Wc * factor, ol, coarse.device)
    obs, ops, _, _ = _img(out, ol, n)
    nat.check(lib.hsr_bilinear_upsample(_ptr(coarse), ibs, ips, n, Hc, Wc, factor, _ptr(out), obs, ops,
                                        _stream(torch)), "hsr_bilinear_upsample")
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
