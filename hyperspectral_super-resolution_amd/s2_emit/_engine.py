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


class _launch:
    """Context of one or a few library calls on the device that owns ``t``: makes that device current for the HIP
    launch (kernels go to the CURRENT device, whatever device the pointers belong to) and yields the stream torch
    has current on THAT device.  A plan built for cuda:1 in a process whose current device is cuda:0 therefore
    launches on cuda:1, on cuda:1's stream."""
    __slots__ = ("dev", "guard")

    def __init__(self, t):
        self.dev = t.device if hasattr(t, "device") else t
        self.guard = None

    def __enter__(self):
        import torch
        if self.dev.type != "cuda":
            raise ValueError(f"expected a GPU tensor, got one on {self.dev}")
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        if idx != torch.cuda.current_device():
            self.guard = torch.cuda.device(idx)
            self.guard.__enter__()
        return C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


def _stream(torch, t=None):
    """Stream for callers that do not switch devices themselves (ridge.py, _ot.py): the current stream of the current
    device - and a loud error if ``t`` lives on another device instead of a launch on the wrong GPU."""
    cur = torch.cuda.current_device()
    if t is not None and t.is_cuda and t.device.index not in (None, cur):
        raise ValueError(f"tensor on {t.device} but the current device is cuda:{cur}: wrap the call in "
                         f"torch.cuda.device({t.device.index})")
    return C.c_void_p(torch.cuda.current_stream(cur).cuda_stream)


def srf_options(tile_pixels: int = 0, reserved_cus: int = 0, u16_single_buffer: bool = False, u16_fast: bool = False):
    """hsr_srf_options for one call / one plan (None everywhere means the defaults).  u16_fast: opt-in fast arithmetic
    of the uint16 kernels (decode scale folded into the weights; ~1e-7 relative off the bit-exact path)."""
    return nat.SrfOptions(int(tile_pixels), int(reserved_cus), 1 if u16_single_buffer else 0,
                          nat.HSR_SRF_U16_FAST if u16_fast else 0)


def _opt(opts):
    return C.byref(opts) if opts is not None else None


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
    with _launch(x) as st:
        nat.check(lib.hsr_tile_encode_u16(_ptr(x), x.numel(), float(scale), int(src_nodata is not None),
                                          float(src_nodata) if src_nodata is not None else 0.0, int(nodata_u16),
                                          _ptr(out), st), "hsr_tile_encode_u16")
    return out


def tile_decode_u16(u, scale=None, nodata: Optional[int] = TILE_NODATA):
    """uint16 GPU tensor -> float32: float32(u) * float32(scale) (default 1e-4), nodata -> NaN."""
    torch = nat.require_gpu()
    lib = nat.load()
    if not (u.is_cuda and u.dtype == torch.uint16 and u.is_contiguous()):
        raise ValueError("u must be a contiguous uint16 tensor on the GPU")
    out = torch.empty(u.shape, dtype=torch.float32, device=u.device)
    with _launch(u) as st:
        nat.check(lib.hsr_tile_decode_u16(_ptr(u), u.numel(), _decode_scale(scale), -1 if nodata is None else int(nodata),
                                          _ptr(out), st), "hsr_tile_decode_u16")
    return out


def _i32arr(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def srf_integrate(cube, table: SrfTable, out=None, layout: str = PLANAR, scale=None, nodata: Optional[int] = TILE_NODATA,
                  opts=None):
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
        with _launch(c2) as st:
            if c2.dtype == torch.uint16:
                nat.check(lib.hsr_srf_integrate_u16(_ptr(c2), npix, B, _decode_scale(scale), -1 if nodata is None else int(nodata),
                                                    _ptr(wn[b0:b1]), k0p, klp, b1 - b0, _ptr(dst), bs, ps, _opt(opts), st),
                          "hsr_srf_integrate_u16")
            else:
                nat.check(lib.hsr_srf_integrate(_ptr(c2), npix, B, _ptr(wn[b0:b1]), k0p, klp, b1 - b0,
                                                _ptr(dst), bs, ps, _opt(opts), st), "hsr_srf_integrate")
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
        # fused fit (hsr_srf_integrate_fit): group sums and the arrival tickets (zero between launches)
        self.group_partials = torch.empty(nat.HSR_FIT_GROUPS * nb * self.M, dtype=torch.float64, device=device)
        self.tickets = torch.zeros(nat.HSR_FIT_TICKETS, dtype=torch.int32, device=device)

    def fused_fit(self, min_count: int):
        """The hsr_fused_fit record of this workspace."""
        return nat.FusedFit(self.group_partials.data_ptr(), self.tickets.data_ptr(), self.moments.data_ptr(),
                            self.coeffs.data_ptr(), int(min_count))


def srf_integrate_moments(cube, table: SrfTable, real, deg: int, ws: MomentWorkspace, mask=None,
                          min_x=_NEG_INF, min_y=_NEG_INF, out=None, events=None, reduce=True,
                          layout: str = PIXMAJOR, real_layout: Optional[str] = None, scale=None,
                          nodata: Optional[int] = TILE_NODATA, opts=None, fit_min_count: Optional[int] = None):
    """K1+K2 fused: pseudo-S2 image and the per-band Vandermonde moments in one cube pass.
    A uint16 cube is decoded inside the kernel (see srf_integrate).
    ``fit_min_count``: given, the same launch also reduces the slots and solves (hsr_srf_integrate_fit): returns
    (image, (moments, coeffs)) - the workspace's tensors - and ``reduce`` is ignored.
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
    with _launch(c2) as st:
        if events is not None:
            events[0].record(torch.cuda.current_stream(c2.device))
        if fit_min_count is not None:
            fit = ws.fused_fit(fit_min_count)
            if c2.dtype == torch.uint16:
                nat.check(lib.hsr_srf_integrate_fit_u16(_ptr(c2), npix, B, _decode_scale(scale),
                                                        -1 if nodata is None else int(nodata), _ptr(wn), k0p, klp, nb,
                                                        _ptr(img), bs, ps, _ptr(real), rbs, rps, _ptr(mask), min_x, min_y,
                                                        deg, _ptr(ws.partials), C.byref(slots), C.byref(fit), _opt(opts), st),
                          "hsr_srf_integrate_fit_u16")
            else:
                nat.check(lib.hsr_srf_integrate_fit(_ptr(c2), npix, B, _ptr(wn), k0p, klp, nb, _ptr(img), bs, ps,
                                                    _ptr(real), rbs, rps, _ptr(mask), min_x, min_y, deg,
                                                    _ptr(ws.partials), C.byref(slots), C.byref(fit), _opt(opts), st),
                          "hsr_srf_integrate_fit")
        elif c2.dtype == torch.uint16:
            nat.check(lib.hsr_srf_integrate_moments_u16(_ptr(c2), npix, B, _decode_scale(scale),
                                                        -1 if nodata is None else int(nodata), _ptr(wn), k0p, klp, nb,
                                                        _ptr(img), bs, ps, _ptr(real), rbs, rps, _ptr(mask), min_x, min_y,
                                                        deg, _ptr(ws.partials), C.byref(slots), _opt(opts), st),
                      "hsr_srf_integrate_moments_u16")
        else:
            nat.check(lib.hsr_srf_integrate_moments(_ptr(c2), npix, B, _ptr(wn), k0p, klp, nb, _ptr(img), bs, ps,
                                                    _ptr(real), rbs, rps, _ptr(mask), min_x, min_y, deg,
                                                    _ptr(ws.partials), C.byref(slots), _opt(opts), st),
                      "hsr_srf_integrate_moments")
        if events is not None:
            events[1].record(torch.cuda.current_stream(c2.device))
    ws.slots = slots.value
    if fit_min_count is not None:
        return img, (ws.moments, ws.coeffs)
    if not reduce:                      # caller continues with moments_reduce_solve / moments_reduce
        return img, None
    return img, moments_reduce(ws)


# ---------------------------------------------------------------------------------------------
# batched small tiles (hsr.h "batched small tiles"; the reference's 100 x 100 tile pairs,
# tiles_helpers/utils.py:223-305)
# ---------------------------------------------------------------------------------------------
PLACEMENT_PITCH_GB = 16.0


_placement_logged = False


def placement_search(first, make, probe, trials: int, pitch_gb: float, device, budget_gb: Optional[float] = None,
                     candidate_bytes: int = 0, stats: Optional[dict] = None):
    """Where a buffer lies in device memory changes K1's speed by ~9 % (profiles/r02_two_speeds.md: a map of 28
    candidate sets per process shows stretches of 15-25 GB where K1 runs at 0.192-0.204 ms between stretches of 20-70 GB
    where it runs at 0.208-0.227 ms; the speed is a stable property of the allocation; profiles/r03_placement_mechanism.md:
    not address translation, not the allocator - no allocation recipe avoids the slow class, so the only lever is to look).
    This times ``probe(candidate)`` - one K1 launch - on ``first`` and on up to ``trials - 1`` candidates from
    ``make()``, one every ``pitch_gb`` GB (a spacer allocation between candidates, held until the end so that the next
    one lands in another stretch), one untimed and two timed launches each, and returns (fastest candidate, times in ms).
    Every candidate is timed - the slow class is wide (0.208-0.235 ms) and an early exit on "both speeds seen" stopped
    inside it.  Same bytes whichever candidate is kept, so results do not change.

    MEMORY.  Every extra candidate pins ``pitch_gb`` GB of spacer plus ``candidate_bytes`` until the search ends, and all of
    it stays in torch's caching allocator afterwards (SpectralFusion.release_search_memory()).  The search is therefore
    BOUNDED: it never holds more than ``budget_gb`` GB (default: half of the device memory free when it starts), it asks
    torch.cuda.mem_get_info before every allocation instead of running into an out-of-memory error (whose handling
    makes torch flush and retry its cache), and it stops at the first candidate that does not fit.  ``stats`` (a dict)
    receives what was held.  The first search of a process is logged once on the "s2_emit" logger."""
    global _placement_logged
    torch = nat.require_gpu()
    stream = torch.cuda.current_stream(device)
    cands, spacers, times = [first], [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    free0, _total = torch.cuda.mem_get_info(device)
    budget = 0.5 * free0 if budget_gb is None else min(float(budget_gb) * (1 << 30), float(free0))
    per_trial = int(pitch_gb * (1 << 30)) + int(candidate_bytes)
    held = 0
    for i in range(max(1, trials)):
        if i > 0:
            free, _ = torch.cuda.mem_get_info(device)
            if held + per_trial > budget or per_trial + (1 << 30) > free:        # would exceed the budget / leave < 1 GB free
                break
            try:
                spacers.append(torch.empty(int(pitch_gb * (1 << 30)), dtype=torch.uint8, device=device))
                cands.append(make())
            except RuntimeError:          # out of memory after all (another process allocated meanwhile)
                break
            held += per_trial
        c = cands[-1]
        probe(c)                           # untimed: first touch of the candidate
        t = []
        for _ in range(2):
            e0.record(stream)
            probe(c)
            e1.record(stream)
            e1.synchronize()
            t.append(e0.elapsed_time(e1))
        times.append(min(t))
    # The fastest of many candidates by two launches each is partly the luckiest: the best few are timed again, six launches each,
    # and the median decides (r04: with 96 dense candidates a 20-step run landed on 0.2207-0.2299 ms where the 100-step runs of the
    # same box gave 0.2120-0.2146).
    if len(times) > 3:
        finals = {}
        for i in sorted(range(len(times)), key=times.__getitem__)[:8]:
            t = []
            for _ in range(6):
                e0.record(stream)
                probe(cands[i])
                e1.record(stream)
                e1.synchronize()
                t.append(e0.elapsed_time(e1))
            finals[i] = sorted(t)[len(t) // 2]
        best = min(finals, key=finals.__getitem__)
    else:
        best = min(range(len(times)), key=times.__getitem__)
    keep = cands[best]
    if stats is not None:
        stats["held_gb"] = stats.get("held_gb", 0.0) + held / (1 << 30)
        stats["budget_gb"] = budget / (1 << 30)
    if not _placement_logged and len(times) > 1:
        _placement_logged = True
        import logging
        logging.getLogger("s2_emit").info(
            "placement search: %d candidate allocations timed (%.4f - %.4f ms), %.1f GB of spacers and losing candidates left "
            "in torch's caching allocator (budget %.1f GB; SpectralFusion.release_search_memory() returns them to the driver)",
            len(times), min(times), max(times), held / (1 << 30), budget / (1 << 30))
    # The spacers and the losing candidates go back to torch's caching allocator, NOT to the driver: releasing them
    # (torch.cuda.empty_cache()) right after the search was measured to cost K1 0.2-3 % again (0.2016-0.2077 ms against
    # 0.2000-0.2007 ms over three fresh processes each) - unmapping ~190 GB next to the kept buffers is not neutral.
    del cands, spacers, c
    return keep, [round(t, 4) for t in times]


def placement_rank(make, probe, count: int, pitch_gb: float, device, budget_gb: Optional[float] = None, candidate_bytes: int = 0,
                   stats: Optional[dict] = None, retime: int = 0):
    """placement_search for a set of interchangeable buffers (the resident tiles of a mosaic): up to ``count`` candidates from
    ``make()``, ``pitch_gb`` GB of spacer between consecutive ones (so that they sample different stretches of device memory), each
    timed with ``probe`` (one untimed + two timed launches).  Returns [(candidate, ms)] for every candidate that fitted the budget,
    in allocation order - the caller keeps the fastest ones.  Same memory rules as placement_search."""
    torch = nat.require_gpu()
    stream = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    free0, _total = torch.cuda.mem_get_info(device)
    budget = 0.5 * free0 if budget_gb is None else min(float(budget_gb) * (1 << 30), float(free0))
    per_trial = int(pitch_gb * (1 << 30)) + int(candidate_bytes)
    held, out, spacers = 0, [], []
    for i in range(max(1, count)):
        free, _ = torch.cuda.mem_get_info(device)
        if i > 0 and (held + per_trial > budget or per_trial + (1 << 30) > free):
            break
        try:
            if i > 0 and pitch_gb > 0:
                spacers.append(torch.empty(int(pitch_gb * (1 << 30)), dtype=torch.uint8, device=device))
            c = make()
        except RuntimeError:
            break
        held += per_trial
        probe(c)
        t = []
        for _ in range(2):
            e0.record(stream)
            probe(c)
            e1.record(stream)
            e1.synchronize()
            t.append(e0.elapsed_time(e1))
        out.append((c, min(t)))
    if retime > 0 and len(out) > retime:           # the best `retime` again, six launches each, median (see placement_search)
        for i in sorted(range(len(out)), key=lambda k: out[k][1])[:retime]:
            t = []
            for _ in range(6):
                e0.record(stream)
                probe(out[i][0])
                e1.record(stream)
                e1.synchronize()
                t.append(e0.elapsed_time(e1))
            out[i] = (out[i][0], sorted(t)[len(t) // 2])
    if stats is not None:
        stats["held_gb"] = stats.get("held_gb", 0.0) + held / (1 << 30)
        stats["budget_gb"] = budget / (1 << 30)
    del spacers
    return out


class TileBatch:
    """Device tables and workspaces of one batch of tiles for the three batched launches.

    ``cubes`` / ``reals`` / ``masks``: per-tile GPU tensors - cube (H, W, B) or (npix, B) float32 / uint16,
    real (H, W, C >= nb) or (npix, C) float32 band-last (all tiles with the same C), mask uint8 (npix) or None.
    Output rows are ``padded_row(nb)`` floats.  The tables hold raw pointers: the batch keeps the tensors alive."""

    def __init__(self, cubes, reals, masks, table: SrfTable, deg: int, opts=None, in_place: bool = False):
        torch = nat.require_gpu()
        lib = nat.load()
        T = len(cubes)
        if T < 1:
            raise ValueError("a batch needs at least one tile")
        if reals is not None and len(reals) != T:
            raise ValueError("reals must match cubes")
        masks = list(masks) if masks is not None else [None] * T
        if len(masks) != T:
            raise ValueError("masks must match cubes")
        self.table, self.deg, self.nb, self.T = table, int(deg), table.nb, T
        self.M = 3 * deg + 2 if deg > 0 else 1
        self.row = padded_row(self.nb)
        dev = cubes[0].device
        self.device = dev
        self.cubes = [_as_cube2d(c) for c in cubes]
        dts = {c.dtype for c in self.cubes}
        if len(dts) != 1:
            raise ValueError("all cubes of a batch must have the same dtype")
        self.u16 = self.cubes[0].dtype == torch.uint16
        self.npix = [int(c.shape[0]) for c in self.cubes]
        for c in self.cubes:
            if c.shape[1] != table.B:
                raise ValueError(f"emit_w must be (B,) matching R bands. Got {(table.B,)} vs {c.shape[1]}")
            if c.device != dev:
                raise ValueError("all tiles of a batch must live on one GPU")
        self.reals, self.real_row = None, 0
        if deg > 0:
            if reals is None:
                raise ValueError("deg > 0 needs the real-S2 targets")
            self.reals = []
            for r, n in zip(reals, self.npix):
                r2 = r.reshape(-1, r.shape[-1])
                if not (r2.is_cuda and r2.dtype == torch.float32 and r2.is_contiguous() and r2.shape[0] == n and r2.shape[1] >= self.nb):
                    raise ValueError("real targets must be contiguous float32 band-last GPU tensors (npix, C >= nb)")
                self.reals.append(r2)
            rows = {int(r.shape[1]) for r in self.reals}
            if len(rows) != 1:
                raise ValueError("all real targets of a batch must have the same number of channels")
            self.real_row = rows.pop()
        self.masks = []
        for m, n in zip(masks, self.npix):
            if m is not None:
                m = m.reshape(-1)
                if m.dtype == torch.bool:
                    m = m.view(torch.uint8)
                if not (m.is_cuda and m.dtype == torch.uint8 and m.numel() == n and m.is_contiguous()):
                    raise ValueError("mask must be a contiguous uint8 tensor with one byte per pixel")
            self.masks.append(m)
        total = sum(self.npix)
        self.offsets = np.concatenate([[0], np.cumsum(self.npix)]).astype(np.int64)
        self.opts = opts
        self.in_place = bool(in_place)
        self.placement_log = None
        self._build(torch.empty((total, self.row), dtype=torch.float32, device=dev))
        self.moments = torch.zeros((T, self.nb, self.M), dtype=torch.float64, device=dev)
        self.coeffs = torch.zeros((T, self.nb, max(deg, 0) + 1), dtype=torch.float64, device=dev)
        # a mosaic's global fit (fuse_mosaic(resident=True)): the sum over the tiles and its polynomial live with the batch
        self.total_moments = torch.zeros((self.nb, self.M), dtype=torch.float64, device=dev)
        self.global_coeffs = torch.zeros((self.nb, max(deg, 0) + 1), dtype=torch.float64, device=dev)

    def _build(self, pseudo):
        """Tile and unit tables for the output image ``pseudo`` (the tables hold raw pointers into it)."""
        torch = nat.require_gpu()
        lib = nat.load()
        T, dev = self.T, self.device
        self.pseudo = pseudo
        self.matched = pseudo if self.in_place else torch.empty_like(pseudo)
        tiles = (nat.BatchTile * T)()
        esz = 4 * self.row
        for i in range(T):
            tl = tiles[i]
            tl.cube_dev = self.cubes[i].data_ptr()
            tl.real_dev = self.reals[i].data_ptr() if self.reals is not None else None
            tl.mask_dev = self.masks[i].data_ptr() if self.masks[i] is not None else None
            tl.pseudo_dev = self.pseudo.data_ptr() + int(self.offsets[i]) * esz
            tl.matched_dev = self.matched.data_ptr() + int(self.offsets[i]) * esz
            tl.npix = self.npix[i]
        info = nat.BatchInfo()
        nat.check(lib.hsr_batch_plan(tiles, T, self.nb, self.deg, None, _opt(self.opts), None, 0, C.byref(info)), "hsr_batch_plan")
        nunits = int(info.nunits)
        if getattr(self, "partials", None) is None:
            self.partials = torch.empty(max(1, lib.hsr_batch_partials_bytes(nunits, self.nb, self.deg) // 8), dtype=torch.float64, device=dev)
        units = (C.c_uint8 * (nat.BATCH_RECORD_BYTES * nunits))()
        nat.check(lib.hsr_batch_plan(tiles, T, self.nb, self.deg, _ptr(self.partials), _opt(self.opts), units, nunits, C.byref(info)),
                  "hsr_batch_plan")
        self.info = info
        self.tiles_dev = torch.frombuffer(bytearray(bytes(tiles)), dtype=torch.uint8).to(dev)
        self.units_dev = torch.frombuffer(bytearray(bytes(units)), dtype=torch.uint8).to(dev)
        self.slots = [int(tiles[i].slots) for i in range(T)]

    def place(self, probe, trials: int, pitch_gb: float = PLACEMENT_PITCH_GB, budget_gb: Optional[float] = None, stats=None):
        """Placement trials for the batch's output image (placement_search): ``probe(self)`` enqueues one batched K1
        launch; candidates are fresh allocations of the output image; the fastest is kept."""
        torch = nat.require_gpu()
        if trials <= 1 or self.pseudo.numel() * 4 < (1 << 24):
            return

        def run(cand):
            if cand is not self.pseudo:
                self._build(cand)
            probe(self)
        keep, times = placement_search(self.pseudo, lambda: torch.empty_like(self.pseudo), run, trials, pitch_gb, self.device,
                                       budget_gb, self.pseudo.numel() * 4, stats)
        if keep is not self.pseudo:
            self._build(keep)
        self.placement_log = times

    def tile_rows(self, i: int, image: str = "matched"):
        """(npix_i, row) view of tile i inside the batch's pseudo / matched image."""
        img = getattr(self, image)
        return img[int(self.offsets[i]): int(self.offsets[i + 1])]


def batch_srf_integrate_moments(tb: TileBatch, min_x=_NEG_INF, min_y=_NEG_INF, scale=None,
                                nodata: Optional[int] = TILE_NODATA, events=None):
    """K1 (+K2) of every tile of the batch in ONE launch -> tb.pseudo, partial sums in tb.partials."""
    torch = nat.require_gpu()
    lib = nat.load()
    wn = tb.table.device_weights(tb.device)
    k0a, k0p = _i32arr(tb.table.k0)
    kla, klp = _i32arr(tb.table.klen)
    with _launch(tb.pseudo) as st:
        if events is not None:
            events[0].record(torch.cuda.current_stream(tb.device))
        nat.check(lib.hsr_srf_integrate_moments_batched(_ptr(tb.units_dev), C.byref(tb.info), 2 if tb.u16 else 0,
                                                        _decode_scale(scale), -1 if nodata is None else int(nodata),
                                                        tb.table.B, _ptr(wn), k0p, klp, tb.nb, tb.row, tb.real_row,
                                                        min_x, min_y, tb.deg, _opt(tb.opts), st),
                  "hsr_srf_integrate_moments_batched")
        if events is not None:
            events[1].record(torch.cuda.current_stream(tb.device))
    return tb.pseudo


def batch_reduce_solve(tb: TileBatch, min_count: int):
    """Per-tile slot reduction + np.polyfit solve in one launch -> (tb.moments (T,nb,M), tb.coeffs (T,nb,deg+1))."""
    lib = nat.load()
    with _launch(tb.pseudo) as st:
        nat.check(lib.hsr_moments_reduce_solve_batched(_ptr(tb.tiles_dev), tb.T, _ptr(tb.partials), tb.nb, tb.deg,
                                                       int(min_count), _ptr(tb.moments), _ptr(tb.coeffs), st),
                  "hsr_moments_reduce_solve_batched")
    return tb.moments, tb.coeffs


def batch_poly_apply(tb: TileBatch, use_mask: bool = False, clip: bool = True, coeffs=None, shared: bool = False):
    """K3 of every tile in one launch -> tb.matched.  Every tile with its own coefficients (tb.coeffs, or ``coeffs`` of shape
    (T, nb, deg+1)); ``shared=True``: ``coeffs`` is ONE (nb, deg+1) set used by all tiles (a mosaic's global fit)."""
    lib = nat.load()
    co = tb.coeffs if coeffs is None else coeffs
    if shared and co.dim() != 2:
        raise ValueError("shared coefficients must be one (nb, deg+1) tensor")
    with _launch(tb.pseudo) as st:
        nat.check(lib.hsr_poly_apply_batched(_ptr(tb.tiles_dev), tb.T, int(tb.info.max_npix), _ptr(co), tb.nb,
                                             int(co.shape[-1]) - 1, tb.row, (1 if use_mask else 0) | (2 if shared else 0),
                                             1 if clip else 0, st),
                  "hsr_poly_apply_batched")
    return tb.matched


def moments_reduce(ws: MomentWorkspace):
    """Fixed-order reduction of the partial slots of the last moments launch -> ws.moments."""
    torch = nat.require_gpu()
    lib = nat.load()
    with _launch(ws.partials) as st:
        nat.check(lib.hsr_moments_reduce(_ptr(ws.partials), ws.slots, ws.nb, ws.deg, _ptr(ws.moments),
                                         st), "hsr_moments_reduce")
    return ws.moments


def moments_reduce_solve(ws: MomentWorkspace, min_count: int):
    """Reduce + np.polyfit solve in one launch (no exchange in between) -> (ws.moments, ws.coeffs)."""
    torch = nat.require_gpu()
    lib = nat.load()
    with _launch(ws.partials) as st:
        nat.check(lib.hsr_moments_reduce_solve(_ptr(ws.partials), ws.slots, ws.nb, ws.deg, int(min_count),
                                               _ptr(ws.moments), _ptr(ws.coeffs), st),
                  "hsr_moments_reduce_solve")
    return ws.moments, ws.coeffs


def reduce_slots(partials, slots: int, nb: int, deg: int, out=None):
    """hsr_moments_reduce over any [slot][band][moment] float64 array -> (nb, 3 deg + 2) moments: the fixed-order sum (lane
    l adds slots l, l + 64, ...; butterfly over the 64 lane sums) that every form of a mosaic's global fit uses."""
    torch = nat.require_gpu()
    lib = nat.load()
    mo = torch.empty((nb, 3 * deg + 2), dtype=torch.float64, device=partials.device) if out is None else out
    with _launch(partials) as st:
        nat.check(lib.hsr_moments_reduce(_ptr(partials), int(slots), int(nb), int(deg), _ptr(mo), st), "hsr_moments_reduce")
    return mo


def reduce_solve_slots(partials, slots: int, ws: MomentWorkspace, min_count: int, moments=None, coeffs=None):
    """hsr_moments_reduce_solve over any [slot][band][moment] float64 array (e.g. the per-tile moments of a mosaic, one
    "slot" per tile) -> (moments, coeffs): the given tensors, or the workspace's."""
    lib = nat.load()
    mo = ws.moments if moments is None else moments
    co = ws.coeffs if coeffs is None else coeffs
    with _launch(partials) as st:
        nat.check(lib.hsr_moments_reduce_solve(_ptr(partials), int(slots), ws.nb, ws.deg, int(min_count),
                                               _ptr(mo), _ptr(co), st), "hsr_moments_reduce_solve")
    return mo, co


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
    with _launch(x) as st:
        nat.check(lib.hsr_poly_moments(_ptr(x), xbs, xps, _ptr(y), ybs, yps, _ptr(mask), npix, n, deg,
                                       min_x, min_y, _ptr(lohi_x), _ptr(lohi_y), _ptr(ws.partials),
                                       C.byref(slots), st), "hsr_poly_moments")
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
    with _launch(x) as st:
        nat.check(lib.hsr_poly_moments_f64(_ptr(x), xs, _ptr(y), ys, npix, nb, deg,
                                           _ptr(ws.partials), C.byref(slots), st), "hsr_poly_moments_f64")
    ws.slots = slots.value
    return moments_reduce(ws)


def poly_solve(moments, deg: int, min_count: int, out=None):
    """np.polyfit from moments, on the device, stream ordered.  (nb, deg+1) float64, highest first."""
    torch = nat.require_gpu()
    lib = nat.load()
    nb = moments.shape[0]
    coeffs = out if out is not None else torch.empty((nb, deg + 1), dtype=torch.float64, device=moments.device)
    with _launch(moments) as st:
        nat.check(lib.hsr_poly_solve(_ptr(moments), nb, deg, int(min_count), _ptr(coeffs), st),
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
    with _launch(x) as st:
        nat.check(lib.hsr_poly_apply(_ptr(x), xbs, xps, _ptr(mask), _ptr(coeffs), n, deg, npix, _ptr(lohi),
                                     1 if clip else 0, _ptr(o), obs, ops, st), "hsr_poly_apply")
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
        with _launch(x) as st:
            nat.check(lib.hsr_percentile_limits(_ptr(x), xbs, xps, _ptr(mask), npix, n, float(pmin), float(pmax),
                                                _ptr(work), _ptr(lohi), st), "hsr_percentile_limits")
        return lohi
    w32 = work.view(torch.int32)
    with _launch(x) as st:
        nat.check(lib.hsr_percentile_begin(_ptr(work), n, st), "hsr_percentile_begin")
    for p in (1, 2, 3):
        with _launch(x) as st:
            nat.check(lib.hsr_percentile_hist(p, _ptr(x), xbs, xps, _ptr(mask), npix, n, _ptr(work), st),
                      "hsr_percentile_hist")
        off, cnt = C.c_int64(0), C.c_int64(0)
        nat.check(lib.hsr_percentile_hist_region(p, n, C.byref(off), C.byref(cnt)), "hsr_percentile_hist_region")
        region = w32[off.value // 4: off.value // 4 + cnt.value]
        if _reduce is not None:
            _reduce(p, region)
        else:
            import torch.distributed as dist
            dist.all_reduce(region, op=dist.ReduceOp.SUM, group=group)
        with _launch(x) as st:
            nat.check(lib.hsr_percentile_scan(p, n, float(pmin), float(pmax), _ptr(work), _ptr(lohi), st),
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
    with _launch(x) as st:
        nat.check(lib.hsr_valid_mask(_ptr(x), xbs, xps, nx, pos_band, _ptr(y), ybs, yps, ny, _ptr(mask_in), npix,
                                     _ptr(out), st), "hsr_valid_mask")
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
    with _launch(fine) as st:
        nat.check(lib.hsr_block_mean(_ptr(fine), dt, ibs, ips, n, Hc, Wc, factor, float(scale), _ptr(out), obs, ops,
                                     st), "hsr_block_mean")
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
    with _launch(coarse) as st:
        nat.check(lib.hsr_bilinear_upsample(_ptr(coarse), ibs, ips, n, Hc, Wc, factor, _ptr(out), obs, ops,
                                            st), "hsr_bilinear_upsample")
    return out


def bilinear_upsample_mask_limits(coarse, Hc: int, Wc: int, factor: int, pmin=2.0, pmax=98.0, nb: Optional[int] = None):
    """bilinear_upsample (band-last rows of 4 floats) + valid_mask(all bands finite) + percentile_limits of the masked fine
    image in one chain that reads the fine image twice instead of four times: the upsampling kernel writes the mask and
    counts the first radix pass of the percentile select while it still holds the values (hsr_bilinear_upsample_mask_hist).
    Returns (fine (npix, 4) float32, mask (npix,) uint8, lohi (nb, 2) float64) - the bits of the three separate operators."""
    torch = nat.require_gpu()
    lib = nat.load()
    ibs, ips, n, npc = _img(coarse, PIXMAJOR, nb)
    if npc != Hc * Wc or coarse.dtype != torch.float32 or n > 4:
        raise ValueError("bilinear_upsample_mask_limits: coarse must be a float32 band-last image of Hc*Wc pixels, <= 4 bands")
    npf = Hc * factor * Wc * factor
    out = alloc_image(torch, n, npf, PIXMAJOR, coarse.device)
    mask = torch.empty(npf, dtype=torch.uint8, device=coarse.device)
    work = torch.empty(lib.hsr_percentile_work_bytes(n) // 8 + 1, dtype=torch.int64, device=coarse.device)
    lohi = torch.empty((n, 2), dtype=torch.float64, device=coarse.device)
    obs, ops, _, _ = _img(out, PIXMAJOR, n)
    with _launch(coarse) as st:
        nat.check(lib.hsr_percentile_begin(_ptr(work), n, st), "hsr_percentile_begin")
        nat.check(lib.hsr_bilinear_upsample_mask_hist(_ptr(coarse), ibs, ips, n, Hc, Wc, factor, _ptr(out), _ptr(mask),
                                                      _ptr(work), st), "hsr_bilinear_upsample_mask_hist")
        nat.check(lib.hsr_percentile_scan(1, n, float(pmin), float(pmax), _ptr(work), _ptr(lohi), st), "hsr_percentile_scan")
        for p in (2, 3):
            nat.check(lib.hsr_percentile_hist(p, _ptr(out), obs, ops, _ptr(mask), npf, n, _ptr(work), st), "hsr_percentile_hist")
            nat.check(lib.hsr_percentile_scan(p, n, float(pmin), float(pmax), _ptr(work), _ptr(lohi), st), "hsr_percentile_scan")
    return out, mask, lohi


# ---------------------------------------------------------------------------------------------
# C1 from C: an RCCL communicator of the library's own (include/hsr.h "hsr_comm"), set up over a torch.distributed group
# ---------------------------------------------------------------------------------------------
class Comm:
    """hsr_comm: RCCL's C API behind the C ABI.  torch.distributed is used ONCE, to carry rank 0's 128-byte ncclUniqueId to the
    other ranks of ``group``; every collective afterwards is a stream-ordered call into the library (hsr_allreduce_f64 ...),
    issued by the step executor on its side stream or through the methods below.  One communicator per (group, device)."""

    def __init__(self, group=None, device=None, rank: Optional[int] = None, world: Optional[int] = None):
        import torch
        import torch.distributed as dist
        lib = nat.load()
        if not lib.hsr_comm_available():
            raise nat.HsrUnavailable("RCCL (librccl.so.1) could not be bound: no hsr_comm on this machine")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        idbuf = (C.c_ubyte * nat.HSR_COMM_ID_BYTES)()
        if rank is None:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        if rank == 0:
            nat.check(lib.hsr_comm_unique_id(idbuf), "hsr_comm_unique_id")
        if world > 1:
            t = torch.frombuffer(idbuf, dtype=torch.uint8)              # shares idbuf's memory
            on_gpu = dist.get_backend(group) == "nccl"
            tt = t.to(self.device) if on_gpu else t
            dist.broadcast(tt, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if on_gpu:
                t.copy_(tt.cpu())
        self.rank, self.world = int(rank), int(world)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(lib.hsr_comm_init(self.rank, self.world, idbuf, C.byref(self.handle)), "hsr_comm_init")
            # first collective now (connection set-up belongs here, not under the first pipelined tile)
            probe = torch.ones(1, dtype=torch.float64, device=self.device)
            self.allreduce_f64(probe)
            torch.cuda.synchronize(self.device)
            if int(probe.item()) != self.world:
                raise nat.HsrError(f"hsr_comm: an all-reduce of ones over {self.world} rank(s) gave {probe.item()}")

    def allreduce_f64(self, t):
        """In-place sum of a contiguous float64 GPU tensor over the ranks, on the current stream."""
        with _launch(t) as st:
            nat.check(nat.load().hsr_allreduce_f64(self.handle, _ptr(t), t.numel(), st), "hsr_allreduce_f64")
        return t

    def reduce_f64(self, t, root: int = 0):
        with _launch(t) as st:
            nat.check(nat.load().hsr_reduce_f64(self.handle, _ptr(t), t.numel(), int(root), st), "hsr_reduce_f64")
        return t

    def allreduce_u32(self, t):
        with _launch(t) as st:
            nat.check(nat.load().hsr_allreduce_u32(self.handle, _ptr(t), t.numel(), st), "hsr_allreduce_u32")
        return t

    def bcast(self, t, root: int = 0):
        with _launch(t) as st:
            nat.check(nat.load().hsr_bcast(self.handle, _ptr(t), t.numel() * t.element_size(), int(root), st), "hsr_bcast")
        return t

    def close(self):
        h, self.handle = self.handle, None
        if h and nat._lib is not None:
            nat._lib.hsr_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def probe_read_bandwidth(nbytes: int = 1 << 30, iters: int = 10, device="cuda:0", mode: int = 0) -> float:
    """Measured pure-read HBM rate of this box in bytes/s (diagnostic for the roofline report).
    mode 0: K1's own load shape (non-temporal LDS-DMA, 72 KiB slabs, 512 persistent workgroups) - a ceiling for K1;
    mode 1: a plain 16-byte global_load stream; modes 2 / 3: 36 KiB slabs with 2 / 4 workgroups per CU (hsr.h)."""
    torch = nat.require_gpu()
    lib = nat.load()
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    sink = torch.zeros(64, dtype=torch.float32, device=device)
    with _launch(buf) as st:
        for _ in range(2):
            nat.check(lib.hsr_probe_read(_ptr(buf), nbytes, mode, _ptr(sink), st))
        cur = torch.cuda.current_stream(buf.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(iters):
            nat.check(lib.hsr_probe_read(_ptr(buf), nbytes, mode, _ptr(sink), st))
        e1.record(cur)
    e1.synchronize()
    return nbytes * iters / (e0.elapsed_time(e1) * 1e-3)
