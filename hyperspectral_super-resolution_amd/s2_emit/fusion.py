"""The fused EMIT -> Sentinel-2 spectral matching pipeline, device resident.

This is the build's counterpart of the reference driver (s2_emit/poly_regression.py:96-139 ==
Pairs_EMIT_S2_demo-2.ipynb cell 81) for a grid-aligned pair, in its "per-band least squares"
flavour (calibrate_pseudo_to_real_linear, notebook cell 72, degree generalised):

    phase 1  K1+K2  one pass over the (H,W,B) cube: pseudo-S2 planes + Vandermonde moments
    phase 2  C1     (multi-GPU) RCCL exchange of the tiny moment / coefficient vectors
    phase 3  solve  np.polyfit from the moments, on the device (no host round trip)
    phase 4  K3     polynomial apply (+ optional mask, clip) -> matched planes

Everything is stream ordered; nothing synchronises with the host.  One process per GPU; with a
torch.distributed group (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests) every
rank owns whole spatial tiles and only (nb x (3deg+2)) doubles cross the fabric per step.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _engine as eng
from . import _native as nat

COEFF_SYNC_MODES = ("local", "allreduce", "broadcast")
_HOST_GROUPS: Dict[int, object] = {}     # gloo groups of the host transport (one per caller's group), see SpectralFusion._exchange_desc


def _tkey(t):
    """Identity of a tensor's storage view for the batch-plan cache: a reinterpreted view (view(dtype)), another stride or
    another device at the same address and shape is a different key."""
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), str(t.dtype), str(t.device))


@dataclass
class FusionOutput:
    names: List[str]           # supported bands, srf_dict order
    pseudo: object             # pseudo-S2 image (K1), float32 device tensor in ``layout``
    moments: object            # (nb, 3deg+2) float64 device tensor (after the exchange, if any)
    coeffs: object             # (nb, deg+1) float64 device tensor, highest power first
    matched: object            # matched image (K3), float32 device tensor in ``layout``
    layout: str = nat.PIXMAJOR  # PIXMAJOR: (npix, row>=nb) band-last;  PLANAR: (nb, npix)

    def band(self, which, image="matched"):
        """(npix,) view of one band of ``pseudo`` / ``matched`` (by name or index)."""
        i = self.names.index(which) if isinstance(which, str) else int(which)
        img = getattr(self, image)
        return img[i] if self.layout == nat.PLANAR else img[:, i]

    def planes(self, image="matched"):
        """(nb, npix) band-major tensor of ``pseudo`` / ``matched`` (a view for PLANAR, a copy otherwise)."""
        img = getattr(self, image)
        nb = len(self.names)
        return img if self.layout == nat.PLANAR else img[:, :nb].t().contiguous()


class BatchOutput:
    """Result of SpectralFusion.step_batch: per-tile coefficients and the images of all tiles, tile after tile."""

    def __init__(self, names, tb):
        self.names, self._tb = list(names), tb
        self.pseudo, self.matched = tb.pseudo, tb.matched          # (sum npix, row) float32, band-last rows
        self.moments, self.coeffs = tb.moments, tb.coeffs          # (T, nb, 3deg+2), (T, nb, deg+1) float64
        self.layout = nat.PIXMAJOR

    def __len__(self):
        return self._tb.T

    def tile(self, i: int) -> FusionOutput:
        """FusionOutput view of tile i (same fields as step())."""
        tb = self._tb
        return FusionOutput(self.names, tb.tile_rows(i, "pseudo"), tb.moments[i], tb.coeffs[i], tb.tile_rows(i, "matched"),
                            nat.PIXMAJOR)


def exchange_moments(moments, coeffs_solver, group=None, mode: str = "allreduce"):
    """C1: make every rank fit the same polynomial.  ``moments`` (nb, M) float64 tensor (CPU tensor
    with gloo, GPU tensor with RCCL); ``coeffs_solver(moments) -> coeffs`` tensor on the same device.

    mode "local"     : no exchange - independent coefficients per tile (what the reference's own
                       tiling does, tiles_helpers/utils.py:223-305)
    mode "allreduce" : one all-reduce(sum) of the moments, every rank solves redundantly; the
                       result is bit-identical on all ranks because they solve identical inputs
    mode "broadcast" : reduce(sum) to rank 0, rank 0 solves, broadcast of the fitted coefficients
    """
    import torch.distributed as dist
    if mode not in COEFF_SYNC_MODES:
        raise ValueError(f"mode must be one of {COEFF_SYNC_MODES}, got {mode!r}")
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 0
    if mode == "local" or world == 0:
        return moments, coeffs_solver(moments)
    if mode == "allreduce":
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
        return moments, coeffs_solver(moments)
    root = dist.get_global_rank(group, 0) if group is not None else 0
    dist.reduce(moments, dst=root, op=dist.ReduceOp.SUM, group=group)
    coeffs = coeffs_solver(moments)          # only rank 0's is meaningful; shape is what matters
    dist.broadcast(coeffs, src=root, group=group)
    return moments, coeffs


class SpectralFusion:
    """Plan object: SRF weight table + workspaces on one GPU, reused across tiles/steps."""

    def __init__(self, emit_w, srf_dict, good_mask=None, deg: int = 3, min_valid: Optional[float] = 0.0,
                 min_count: int = 50, clip: bool = True, apply_mask: bool = False, device=None,
                 group=None, coeff_sync: str = "allreduce", layout: str = nat.PIXMAJOR,
                 force_exchange: bool = False, tile_scale=None, tile_nodata: Optional[int] = eng.TILE_NODATA,
                 reserved_cus: Optional[int] = None, tile_pixels: int = 0, u16_single_buffer: bool = False,
                 u16_fast: bool = False, placement_trials: int = 0, fused_fit: bool = False,
                 placement_pitch_gb: float = eng.PLACEMENT_PITCH_GB, placement_budget_gb: Optional[float] = None,
                 side_stream=None, fuse_apply: bool = False, comm=None, rehearsal_collective=None, group_tiles: int = 1):
        """``placement_trials`` (default 0 = OFF: the plan allocates once and never synchronises with the host): opt in to
        the placement search of eng.placement_search for tiles of >= 65 536 pixels - up to min(4, trials) candidate output
        images on the first step()/submit()/step_batch() over a tile size, up to ``trials`` candidate (cube, target, image)
        sets in place_inputs().  Each extra candidate pins ``placement_pitch_gb`` GB of spacer (+ the candidate) while the
        search runs, and the memory stays in torch's caching allocator afterwards (release_search_memory()); the search
        never holds more than ``placement_budget_gb`` GB (default: half of the device memory that is free when it starts)
        and skips candidates that do not fit - on a GPU shared with other allocators pass a small budget or leave it off.
        ``placement_held_gb`` records what the searches of this plan held."""
        torch = nat.require_gpu()
        # decode of uint16 cubes (the reference's tile format, tiles_helpers/utils.py:362-374): x = u * tile_scale
        # (default float32(1e-4)), u == tile_nodata -> NaN (None: no nodata value).  Ignored for float32 cubes.
        self.tile_scale, self.tile_nodata = tile_scale, tile_nodata
        if not 1 <= deg <= nat.HSR_MAX_DEG:
            raise ValueError(f"deg must be in [1, {nat.HSR_MAX_DEG}], got {deg}")
        if coeff_sync not in COEFF_SYNC_MODES:
            raise ValueError(f"coeff_sync must be one of {COEFF_SYNC_MODES}")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.table = eng.build_srf_table(emit_w, srf_dict, good_mask)
        if self.table.nb == 0:
            raise ValueError("no band of srf_dict has support on the (masked) EMIT wavelength grid")
        if self.table.nb > nat.HSR_MAX_BANDS:
            raise ValueError(f"at most {nat.HSR_MAX_BANDS} supported bands per fusion plan")
        self.deg, self.min_count, self.clip, self.apply_mask = deg, int(min_count), bool(clip), bool(apply_mask)
        self.min_valid = float("-inf") if min_valid is None else float(min_valid)
        self.group, self.coeff_sync = group, coeff_sync
        if layout not in (nat.PIXMAJOR, nat.PLANAR):
            raise ValueError(f"layout must be {nat.PIXMAJOR!r} or {nat.PLANAR!r}")
        self.layout = layout
        self.force_exchange = bool(force_exchange)   # run the collective path even with one rank (tests)
        # K1 launch geometry of THIS plan (hsr_srf_options travel with every call; the library has no tuning state, so
        # two plans in one process cannot interfere).  reserved_cus: CUs left free of persistent K1 workgroups so that
        # the side-stream kernels of submit() (slot reduction, RCCL exchange, solve) run under K1 of the next tile; by
        # default 8 (one per XCD) for a plan that exchanges, 0 otherwise.  step(), submit() and step_batch() of one plan
        # use the same geometry, hence the same partial-slot layout and bit-identical coefficients.
        if reserved_cus is None:
            reserved_cus = 8 if self._exchanges() else 0
        self.opts = eng.srf_options(tile_pixels, reserved_cus, u16_single_buffer, u16_fast)
        self.u16_single_buffer = bool(u16_single_buffer)
        self._batches: Dict[tuple, object] = {}
        # Placement (profiles/r02_two_speeds.md, eng.placement_search): K1 runs at one of two speeds, ~9 % apart, depending
        # on which stretch of device memory its operands lie in - the cube above all (its read stream), the output image
        # and the target a little (3 %).  A property of the allocation, stable for its life.  The plan places what it owns:
        # on the first step over a tile size it times K1 on up to min(4, placement_trials) candidate output images,
        # `placement_pitch_gb` apart; place_inputs() searches jointly over copies of a resident tile's inputs (up to
        # `placement_trials` sets).  Results are bit-identical whichever is kept; 0 / 1 (the default) = first allocation,
        # no host synchronisation, no extra memory.  Opt-in and bounded by placement_budget_gb (see the docstring).
        self.placement_trials = max(0, int(placement_trials))
        self.placement_pitch_gb = float(placement_pitch_gb)
        self.placement_budget_gb = None if placement_budget_gb is None else float(placement_budget_gb)
        self.placement_log: Dict[int, list] = {}
        self._placement_stats: Dict[str, float] = {}
        # fused_fit: step() without an exchange lets the slot reduction and the solve ride in K1's launch
        # (hsr_srf_integrate_fit: the last workgroups to finish reduce and solve) - two launches per step instead of
        # three, same bits.  Off by default: measured on MI355X the tail costs K1 +13 us (six dependent round trips
        # through memory-side coherence at ~1.5 us each: ticket, slot loads, group store, ticket, group loads, solve)
        # against 8.5 us + one launch gap for the separate hsr_moments_reduce_solve - 3 us per step slower.
        self.fused_fit = bool(fused_fit)
        self._pipe = None                            # state of submit()/flush(), created on first use
        self.fused_fallback = None                   # why submit() fell back from a fused pipeline to the two-slot one, if it did
        # the stream the fits of submit() run on: None = chosen by measurement on the first submit (_pick_side_stream)
        self.side_stream = side_stream
        self.side_stream_log = None                  # us per pipelined step of each candidate stream, if measured
        # fuse_apply: submit() runs the FUSED pipeline - ONE kernel per tile on the caller's stream and nothing else: launch i
        # = K3 of tile i-2 as a pre-phase + K1+K2 of tile i + the fit of tile i-1 as tail work of the first workgroups to
        # finish (hsr_srf_integrate_moments_apply / _u16_apply).  No side stream, no events, no CUs kept free; results two
        # submits late instead of one.  Needs the pixel-major layout, no exchange (a collective cannot ride in a kernel's tail)
        # and, for uint16 tiles, a cube the ring kernel can load (16-byte aligned, 48 <= B <= 300); otherwise submit()
        # quietly uses the two-slot pipeline.
        self.fuse_apply = bool(fuse_apply)
        # With an exchange, fuse_apply=True runs the FOUR-slot pipeline (hsr_pipeline_create_exchange): still one kernel per tile on
        # the caller's stream - K3 of tile i-3 as a pre-phase, K1+K2 of tile i, the slot reduction of tile i-1 in the tail - while
        # the side stream carries gate -> collective -> solve, all issued from C: no torch.distributed call per step, no event or
        # stream wait on the caller's stream.  The collective is RCCL through the library's own communicator (``comm``: an
        # eng.Comm, or None = built over ``group`` on first use) when the group's backend is nccl; with any other backend (gloo)
        # the moments make a round trip through pinned host memory and torch.distributed sums them there (a host callback).
        # group_tiles = T > 1 (with fuse_apply, no exchange): submit() fits ONE polynomial per group of T consecutive tiles - a mosaic
        # held by this GPU (hsr_pipeline_create_group): launch n = K3 of tile n-T-1 | K1+K2 of tile n | slot reduction of tile n-1 (+ the
        # group's sum and solve) in the tail.  Outputs come T + 1 submits late, with their group's moments and coefficients; drain()
        # needs whole groups.  Same bits as fuse_mosaic() on the same tiles.
        self.group_tiles = int(group_tiles)
        if not 1 <= self.group_tiles <= 64:
            raise ValueError("group_tiles must be in [1, 64]")
        self._comm = comm
        # one-GPU rehearsals only: (microseconds, workgroups) of a stand-in kernel enqueued where the collective's kernel would run
        # (hsr_exchange.rehearsal_us; a one-rank RCCL all-reduce launches nothing)
        self.rehearsal_collective = rehearsal_collective
        self._host_cb = None                         # the ctypes callback object of the host transport (kept alive with the plan)
        self._backlog: List[FusionOutput] = []       # tiles finished by a pipeline rebuild, returned by the next submit() / drain()
        self._native: Dict[tuple, object] = {}       # prepared launches of step(), by _native_key
        self._native_handles: list = []              # ("plan" | "pipe", handle) to destroy with the plan
        self._pipe_images: Dict[int, list] = {}      # output images placed by place_inputs() for the pipeline's two slots
        self._pipe_matched: Dict[int, list] = {}     # ... and K3 output images placed by place_mosaic()
        self.ws = eng.MomentWorkspace(self.device, self.table.nb, deg)
        self.table.device_weights(self.device)
        self._buf: Dict[int, Tuple[object, object]] = {}

    @property
    def names(self) -> List[str]:
        return list(self.table.supported)

    def _buffers(self, npix: int, probe=None):
        """The plan's (pseudo, matched) images for tiles of npix pixels.  ``probe``: callable(pseudo) enqueueing one K1
        launch into a candidate image - given on the first step over this tile size, it drives the placement trials."""
        torch = nat.require_gpu()
        if npix not in self._buf:
            nb = self.table.nb
            pseudo = eng.alloc_image(torch, nb, npix, self.layout, self.device)
            if probe is not None and self.placement_trials > 1 and npix >= (1 << 16):
                pseudo = self._place(npix, pseudo, probe)
            self._buf[npix] = (pseudo, eng.alloc_image(torch, nb, npix, self.layout, self.device))
        return self._buf[npix]

    @property
    def placement_held_gb(self) -> float:
        """GB of spacers and losing candidates this plan's searches left in torch's caching allocator."""
        return float(self._placement_stats.get("held_gb", 0.0))

    def _trials(self, first, make, probe, cap=None, candidate_bytes: int = 0):
        n = self.placement_trials if cap is None else min(cap, self.placement_trials)
        return eng.placement_search(first, make, probe, n, self.placement_pitch_gb, self.device, self.placement_budget_gb,
                                    candidate_bytes, self._placement_stats)

    def _place(self, npix: int, first, probe, count: int = 1):
        """Time K1 on candidate output images and keep the fastest (the output's share of the effect is ~3 %: four
        candidates at most).  ``count`` > 1: a candidate is ``count`` images allocated back to back (the two slots of the
        submit() pipeline are searched ONCE, together); ``first`` is then a list and a list is returned."""
        torch = nat.require_gpu()
        nb = self.table.nb
        one = lambda: eng.alloc_image(torch, nb, npix, self.layout, self.device)
        if count == 1:
            keep, times = self._trials(first, one, probe, cap=4, candidate_bytes=first.numel() * 4)
        else:
            keep, times = self._trials(list(first), lambda: [one() for _ in range(count)], lambda c: probe(c[0]), cap=4,
                                       candidate_bytes=count * first[0].numel() * 4)
        self.placement_log[npix] = times
        return keep

    def place_inputs(self, cube, real, mask=None):
        """Placement trials for a tile that stays resident and is processed many times (a benchmark loop, a resident
        mosaic).  step() by itself only places the images the plan owns, but it is the CUBE's stretch of device memory
        that carries most of the effect (tools/dbg/placement_map.py: the slow set's cube with the fast set's target and
        output runs slow, the fast set's cube with the slow set's target and output runs fast).  This searches JOINTLY:
        candidate set i = (cube clone, real clone, output image) allocated back to back - one stretch - with
        placement_pitch_gb between sets; set 0 is the caller's tensors with a fresh output image.  One untimed and two
        timed K1 launches per set; the fastest set's cube and real are
        returned and its output image becomes the plan's image for this tile size (so step() runs no further trial).
        Same bytes, so results are bit-identical.
        Returns (cube, real, log)."""
        torch = nat.require_gpu()
        npix = cube.numel() // cube.shape[-1]
        if self.placement_trials <= 1 or npix in self._buf:
            return cube, real, {}
        nb = self.table.nb

        def images():          # the plan's output image for step() and the (up to three) of the submit() pipeline, in one stretch
            return [eng.alloc_image(torch, nb, npix, self.layout, self.device) for _ in range(4 if self.fuse_apply else 3)]

        def make():
            return (cube.clone(), real.clone(), images())

        def k1(cand):
            c, r, outs = cand
            rr, rl = self._real_image(r, npix)
            eng.srf_integrate_moments(c, self.table, rr, self.deg, self.ws, mask, self.min_valid, self.min_valid, out=outs[0],
                                      reduce=False, layout=self.layout, real_layout=rl, scale=self.tile_scale,
                                      nodata=self.tile_nodata, opts=self.opts)
        first = (cube, real, images())
        cand_bytes = cube.numel() * cube.element_size() + real.numel() * 4 + 4 * npix * eng.padded_row(nb) * 4
        (cube, real, outs), times = self._trials(first, make, k1, candidate_bytes=cand_bytes)
        self._buf[npix] = (outs[0], eng.alloc_image(torch, nb, npix, self.layout, self.device))
        self._pipe_images[npix] = outs[1:]
        self.placement_log[npix] = times
        return cube, real, {"joint_ms": times}

    def place_mosaic(self, tiles, oversample: float = 12.0, pitch_gb: float = 0.0):
        """place_inputs() for the resident tiles of a mosaic (same shapes).  The speed classes of device memory come in stretches of
        15-70 GB (profiles/r03_placement_mechanism.md) - long enough to hold MANY 1.2 GB tiles - so this samples DENSELY: up to
        ``oversample`` x T candidate (cube, target) buffer pairs allocated back to back (``pitch_gb`` of spacer between them, 0 by
        default) within placement_budget_gb, K1 timed on each, the T fastest kept and the tiles copied into them - which tile lands
        in which buffer does not matter, the buffers are interchangeable.  (A 9.6 GB mosaic allocated in one piece necessarily spans
        both classes; r03 left "individually probed chunks" open.)  Returns (tiles as (cube, real) in the kept buffers, log).  Needs
        placement_trials > 1; same bytes, same results."""
        torch = nat.require_gpu()
        tiles = [(c, r) for c, r in tiles]
        T = len(tiles)
        if self.placement_trials <= 1 or T == 0:
            return tiles, {}
        c0, r0 = tiles[0]
        npix = c0.numel() // c0.shape[-1]
        img = eng.alloc_image(torch, self.table.nb, npix, self.layout, self.device)

        def k1(cand):
            c, r = cand
            rr, rl = self._real_image(r, npix)
            eng.srf_integrate_moments(c, self.table, rr, self.deg, self.ws, None, self.min_valid, self.min_valid, out=img, reduce=False,
                                      layout=self.layout, real_layout=rl, scale=self.tile_scale, nodata=self.tile_nodata, opts=self.opts)
        cand_bytes = c0.numel() * c0.element_size() + r0.numel() * 4
        per = pitch_gb * (1 << 30) + cand_bytes
        free, _ = torch.cuda.mem_get_info(self.device)
        budget = 0.5 * free if self.placement_budget_gb is None else min(self.placement_budget_gb * (1 << 30), free)
        count = int(min(max(T, oversample * T), max(T, budget // per)))
        pitch = pitch_gb
        ranked = eng.placement_rank(lambda: (c0.clone(), r0.clone()), k1, count, pitch, self.device, self.placement_budget_gb, cand_bytes,
                                    self._placement_stats, retime=2 * T)
        ranked += [((c, r), float("inf")) for c, r in tiles[:max(0, T - len(ranked))]]      # too little memory for T candidates: the originals
        order = sorted(range(len(ranked)), key=lambda i: ranked[i][1])[:T]
        kept = []
        for (c, r), i in zip(tiles, order):
            bc, br = ranked[i][0]
            if bc is not c:
                bc.copy_(c)
                br.copy_(r)
            kept.append((bc, br))
        times = [round(t, 4) for _, t in ranked if t != float("inf")]
        log = {"candidate_ms": times, "kept_ms": [round(ranked[i][1], 4) for i in order], "pitch_gb": round(pitch, 2)}
        # the images of the group pipeline (T + 2 slots, a K1 output and a K3 output each): 50 MB apiece, ranked the same way - K1 on the
        # fastest cube writing into each of 3 x as many candidates as needed - and handed to the pipeline the next submit() builds
        if self.fuse_apply and self.group_tiles == T and npix not in self._pipe_images:
            need = 2 * (T + 2)
            fast = kept[0]

            def k1img(cand):
                rr, rl = self._real_image(fast[1], npix)
                eng.srf_integrate_moments(fast[0], self.table, rr, self.deg, self.ws, None, self.min_valid, self.min_valid, out=cand, reduce=False,
                                          layout=self.layout, real_layout=rl, scale=self.tile_scale, nodata=self.tile_nodata, opts=self.opts)
            imgs = eng.placement_rank(lambda: eng.alloc_image(torch, self.table.nb, npix, self.layout, self.device), k1img, 3 * need, 0.0, self.device,
                                      self.placement_budget_gb, img.numel() * 4, self._placement_stats)
            if len(imgs) >= need:
                best = sorted(range(len(imgs)), key=lambda i: imgs[i][1])[:need]
                self._pipe_images[npix] = [imgs[i][0] for i in best[:T + 2]]
                self._pipe_matched[npix] = [imgs[i][0] for i in best[T + 2:]]
                log["image_ms"] = [round(imgs[i][1], 4) for i in best]
        return kept, log

    def _exchanges(self) -> bool:
        import torch.distributed as dist
        if self.coeff_sync == "local" or not (dist.is_available() and dist.is_initialized()):
            return False
        return self.force_exchange or dist.get_world_size(self.group) > 1

    def _solve(self, moments):
        return eng.poly_solve(moments, self.deg, self.min_count, out=self.ws.coeffs)

    def _real_image(self, real, npix):
        """Accept the real-S2 target as (H,W,C>=nb)/(npix,C) band-last or (nb,H,W)/(nb,npix) band-major."""
        nb = self.table.nb
        if real.dim() == 3:
            if real.shape[0] * real.shape[1] == npix and real.shape[2] >= nb:
                return real.reshape(npix, real.shape[2]), nat.PIXMAJOR
            if real.shape[0] == nb and real.shape[1] * real.shape[2] == npix:
                return real.reshape(nb, npix), nat.PLANAR
        elif real.dim() == 2:
            if real.shape[0] == npix and real.shape[1] >= nb and not (real.shape == (nb, npix) and self.layout == nat.PLANAR):
                return real, nat.PIXMAJOR
            if tuple(real.shape) == (nb, npix):
                return real, nat.PLANAR
        raise ValueError(f"real S2 target of shape {tuple(real.shape)} matches neither (npix,{nb}+) nor ({nb},npix)")

    # ---- prepared launches (include/hsr.h "step executor", csrc/hsr_exec.hip) -----------------------------------
    # step() / submit() of a plan normally see the same shapes again and again.  The first call for a (cube shape and dtype,
    # target shape and strides, mask or not) combination goes through the individual operators (which validate everything
    # and run the opt-in placement trials); it also builds an hsr_step_plan holding every argument of the three launches.
    # Later calls are ONE ctypes call with the three pointers that change - the same launches, hence the same bits, without
    # ~35 us of Python per step (profiles/r03_strong_scaling.md).
    def _native_key(self, cube, real, mask):
        return (tuple(cube.shape), cube.dtype, tuple(real.shape), tuple(real.stride()), real.dtype, mask is not None)

    def _native_desc(self, cube, real2, real_layout, pseudo, matched, ws):
        """hsr_step_desc for tiles shaped like ``cube`` / ``real2`` writing into the given images and workspace."""
        torch = nat.require_gpu()
        import ctypes as C
        c2 = eng._as_cube2d(cube)
        npix, B = c2.shape
        nb = self.table.nb
        rbs, rps, _, _ = eng._img(real2, real_layout, nb)
        obs, ops_, _, _ = eng._img(pseudo, self.layout, nb)
        mbs, mps, _, _ = eng._img(matched, self.layout, nb)
        k0 = (C.c_int32 * nb)(*[int(v) for v in self.table.k0])
        kl = (C.c_int32 * nb)(*[int(v) for v in self.table.klen])
        wn = self.table.device_weights(self.device)
        d = nat.StepDesc()
        d.cube_dtype = 2 if c2.dtype == torch.uint16 else 0
        d.B, d.npix = int(B), int(npix)
        d.scale = eng._decode_scale(self.tile_scale)
        d.nodata = -1 if self.tile_nodata is None else int(self.tile_nodata)
        d.wn_dev = wn.data_ptr()
        d.k0, d.klen = C.cast(k0, C.POINTER(C.c_int32)), C.cast(kl, C.POINTER(C.c_int32))
        d.nb, d.deg = nb, self.deg
        d.pseudo_dev, d.out_bs, d.out_ps = pseudo.data_ptr(), obs, ops_
        d.real_bs, d.real_ps = rbs, rps
        d.min_x = d.min_y = self.min_valid
        d.partials_dev, d.moments_dev, d.coeffs_dev = ws.partials.data_ptr(), ws.moments.data_ptr(), ws.coeffs.data_ptr()
        d.min_count = self.min_count
        d.matched_dev, d.matched_bs, d.matched_ps = matched.data_ptr(), mbs, mps
        d.apply_mask, d.clip = int(self.apply_mask), int(self.clip)
        d.opts = self.opts
        return d, (k0, kl, wn)

    def _native_plan(self, cube, real2, real_layout, pseudo, matched, ws):
        import ctypes as C
        lib = nat.load()
        d, keep = self._native_desc(cube, real2, real_layout, pseudo, matched, ws)
        h = C.c_void_p()
        nat.check(lib.hsr_step_plan_create(C.byref(d), C.byref(h)), "hsr_step_plan_create")
        self._native_handles.append(("plan", h))
        return h, (keep, pseudo, matched, ws)         # the plan stores raw pointers: keep the tensors alive with it

    def close(self):
        """Destroy the prepared launches of this plan (also done when the plan is garbage collected)."""
        lib = nat._lib
        handles, self._native_handles = getattr(self, "_native_handles", []), []
        self._native = {}
        self._pipe = None
        if lib is None:
            return
        self._destroy_handles(handles)

    @staticmethod
    def _destroy_handles(handles):
        lib = nat._lib
        for kind, h in reversed(handles):                       # pipelines before the plans they point to
            (lib.hsr_pipeline_destroy if kind == "pipe" else lib.hsr_step_plan_destroy)(h)

    # ---- the exchange of an exchange pipeline ----------------------------------------------------------------------------
    def comm(self):
        """The library's own RCCL communicator over this plan's group (built on first use; collective over the group)."""
        if self._comm is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise nat.HsrError("no torch.distributed group to carry the communicator's id")
            self._comm = eng.Comm(self.group, self.device)
        return self._comm

    def _exchange_desc(self):
        """hsr_exchange of this plan: RCCL from C for an nccl group, the host transport (pinned round trip + torch.distributed on the
        CPU tensor, called from a runtime thread in stream order) for any other backend."""
        import ctypes as C
        import torch.distributed as dist
        torch = nat.require_gpu()
        x = nat.Exchange()
        x.mode = nat.HSR_SYNC_BROADCAST if self.coeff_sync == "broadcast" else nat.HSR_SYNC_ALLREDUCE
        x.root = 0
        if self.rehearsal_collective:
            x.rehearsal_us, x.rehearsal_blocks = int(self.rehearsal_collective[0]), int(self.rehearsal_collective[1])
        if self._comm is not None or dist.get_backend(self.group) == "nccl":
            x.comm = self.comm().handle
            x.host_sum = nat.HOST_SUM_FN(0)
            return x, "rccl"
        # A process group of ITS OWN for the callback thread: its collectives are ordered by the side stream - the same order on
        # every rank - while the caller's thread goes on using ``group`` (barriers, its own all-reduces).  Sharing one group let the
        # two threads' collectives pair up differently on different ranks (seen as a gloo abort in bench.py's rehearsal, whose
        # barrier raced the last tiles' sums).  dist.new_group is collective over the default group: plans are built in the same
        # order on every rank (SPMD), as the RCCL communicator requires too.
        world_pg = dist.distributed_c10d._get_default_group()
        key = id(self.group)
        ent = _HOST_GROUPS.get(key)
        if ent is None or ent[0] is not world_pg:          # (a cached group dies with the default group it was made from)
            ranks = None if self.group is None else dist.get_process_group_ranks(self.group)
            ent = _HOST_GROUPS[key] = (world_pg, dist.new_group(ranks=ranks, backend="gloo"))
        group = ent[1]

        def host_sum(_user, values, count):
            try:
                t = torch.from_numpy(np.ctypeslib.as_array(values, shape=(int(count),)))
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            except BaseException:                      # never let an exception cross into the runtime's thread
                return 1
        self._host_cb = nat.HOST_SUM_FN(host_sum)
        x.comm = None
        x.host_sum = self._host_cb
        x.mode = nat.HSR_SYNC_ALLREDUCE                # the host transport always sums everywhere; every rank solves
        return x, "host"

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, cube, real, mask=None, reuse_buffers: bool = True, k1_events=None) -> FusionOutput:
        """One pass of the hot path over one tile.
        cube (H,W,B) or (npix,B) float32 GPU tensor; real: real-S2 target, band-last (H,W,C>=nb) /
        (npix,C) [fast] or band-major (nb,H,W) / (nb,npix); mask optional uint8 (npix), 1 = use."""
        torch = nat.require_gpu()
        exchanges = self._exchanges()
        fast = reuse_buffers and k1_events is None and not exchanges and not self.fused_fit
        if fast:
            ent = self._native.get(self._native_key(cube, real, mask))
            if ent is not None and cube.is_contiguous() and cube.device == self.device and real.device == self.device and \
                    (mask is None or (mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == ent[2] and mask.device == self.device)):
                with eng._launch(cube) as st:
                    nat.check(nat._lib.hsr_step_run(ent[0], cube.data_ptr(), real.data_ptr(), None if mask is None else mask.data_ptr(), st),
                              "hsr_step_run")
                ent[3].slots = nat._lib.hsr_step_plan_slots(ent[0])
                return ent[1]
        npix = cube.numel() // cube.shape[-1]
        real_in = real
        real, real_layout = self._real_image(real, npix)
        if reuse_buffers:
            def probe(img):
                eng.srf_integrate_moments(cube, self.table, real, self.deg, self.ws, mask, self.min_valid, self.min_valid,
                                          out=img, reduce=False, layout=self.layout, real_layout=real_layout,
                                          scale=self.tile_scale, nodata=self.tile_nodata, opts=self.opts)
            pseudo, matched = self._buffers(npix, probe if npix not in self._buf else None)
        else:
            pseudo = matched = None
        fit = self.min_count if (self.fused_fit and not exchanges) else None
        pseudo, fitted = eng.srf_integrate_moments(cube, self.table, real, self.deg, self.ws, mask,
                                                   self.min_valid, self.min_valid, out=pseudo, events=k1_events,
                                                   reduce=False, layout=self.layout, real_layout=real_layout,
                                                   scale=self.tile_scale, nodata=self.tile_nodata, opts=self.opts,
                                                   fit_min_count=fit)
        if exchanges:
            moments = eng.moments_reduce(self.ws)
            moments, coeffs = exchange_moments(moments, self._solve, self.group, self.coeff_sync)
        elif fitted is not None:    # reduce + solve rode in K1's launch
            moments, coeffs = fitted
        else:                       # no exchange between reduce and solve: one launch for both
            moments, coeffs = eng.moments_reduce_solve(self.ws, self.min_count)
        matched = eng.poly_apply(pseudo, coeffs, mask if self.apply_mask else None, None, self.clip,
                                 self.layout, out=matched, nb=self.table.nb)
        out = FusionOutput(self.names, pseudo, moments, coeffs, matched, self.layout)
        if fast and real.is_contiguous() and self.table.nb <= nat.HSR_MAX_BANDS:
            # everything was validated by the operators above: prepare the launches for the next tile of this shape
            h, keep = self._native_plan(cube, real, real_layout, pseudo, matched, self.ws)
            self._native[self._native_key(cube, real_in, mask)] = (h, out, npix, self.ws, keep)
        return out


    # ---- a batch of independent tiles in three launches ---------------------------------------------------
    def step_batch(self, cubes, reals, masks=None, k1_events=None) -> "BatchOutput":
        """The hot path over T independent tiles - the reference's own workload: 100 x 100 EMIT tiles, one fit per
        tile (tiles_helpers/utils.py:223-305, Spectral_matching.ipynb raw lines 293-294) - in THREE launches for the
        whole batch instead of three per tile: batched K1+K2, batched slot reduction + solve, batched K3, every tile
        with its own coefficients (coeff_sync "local" semantics; no collective).  Bit-identical to T step() calls.

        cubes: sequence of (H,W,B)/(npix,B) GPU tensors (float32, or uint16 tiles), or one stacked (T,H,W,B) tensor;
        reals: matching band-last real-S2 targets (H,W,C>=nb)/(npix,C) or one stacked (T,H,W,C) tensor;
        masks: optional sequence of uint8 (npix) tensors / None.  The returned BatchOutput (and the batch's buffers)
        are reused by the next step_batch() call on the same input tensors."""
        torch = nat.require_gpu()
        # stacked inputs seen before: no per-tile slicing on the host (at T = 256 the slices and the per-tile key cost more
        # host time than the three launches take on the GPU)
        stack_key = None
        if masks is None and hasattr(cubes, "dim") and hasattr(reals, "dim") and cubes.dim() == 4 and reals.dim() == 4:
            stack_key = ("stack", _tkey(cubes), _tkey(reals))
            tb = self._batches.get(stack_key)
            if tb is not None:
                eng.batch_srf_integrate_moments(tb, self.min_valid, self.min_valid, self.tile_scale, self.tile_nodata,
                                                events=k1_events)
                eng.batch_reduce_solve(tb, self.min_count)
                eng.batch_poly_apply(tb, use_mask=self.apply_mask, clip=self.clip)
                return BatchOutput(self.names, tb)
        if hasattr(cubes, "dim"):
            cubes = [cubes[i] for i in range(cubes.shape[0])] if cubes.dim() == 4 else [cubes]
        if hasattr(reals, "dim"):
            reals = [reals[i] for i in range(reals.shape[0])] if reals.dim() == 4 else [reals]
        cubes, reals = list(cubes), list(reals)
        masks = list(masks) if masks is not None else [None] * len(cubes)
        key = tuple((_tkey(c), _tkey(r), None if m is None else _tkey(m)) for c, r, m in zip(cubes, reals, masks))
        tb = self._batches.get(key)
        if tb is None:
            if len(self._batches) >= 8:           # a few live batch plans at most (stacked inputs hold two keys)
                self._batches.pop(next(iter(self._batches)))
            tb = eng.TileBatch(cubes, reals, masks, self.table, self.deg, self.opts)
            tb.place(lambda b: eng.batch_srf_integrate_moments(b, self.min_valid, self.min_valid, self.tile_scale, self.tile_nodata),
                     self.placement_trials, self.placement_pitch_gb, self.placement_budget_gb, self._placement_stats)
            self._batches[key] = tb
        if stack_key is not None:
            self._batches[stack_key] = tb              # same batch, found without slicing next time
        eng.batch_srf_integrate_moments(tb, self.min_valid, self.min_valid, self.tile_scale, self.tile_nodata,
                                        events=k1_events)
        eng.batch_reduce_solve(tb, self.min_count)
        eng.batch_poly_apply(tb, use_mask=self.apply_mask, clip=self.clip)
        return BatchOutput(self.names, tb)

    def clear_batches(self):
        """Drop the cached batch plans of step_batch() / fuse_mosaic(resident=True).  A cached plan holds the device tables
        of its tiles AND references to the caller's tensors (the tables store raw pointers); the cache is keyed by each
        tensor's address, shape, strides, dtype and device and keeps at most 8 plans."""
        self._batches.clear()

    @staticmethod
    def release_search_memory():
        """The placement searches leave their spacers and losing candidates (up to ~200 GB) in torch's caching allocator, where
        later torch allocations reuse them.  This hands them back to the driver (torch.cuda.empty_cache()) for allocators
        that do not go through torch - at a price: measured right after a search it cost K1 0.2-3 % of the speed the search
        had just found (eng.placement_search), so call it only if that memory is needed elsewhere."""
        nat.require_gpu().cuda.empty_cache()

    def place_batch_inputs(self, cubes, reals):
        """place_inputs() for a resident batch given as stacked tensors (T, H, W, B) / (T, H, W, C): candidate sets = (copy
        of the stacked cube, copy of the stacked targets, the batch's own output images) one stretch of device memory each,
        one batched K1 launch timed on each, the fastest set kept and its batch cached, so that step_batch() on the returned
        tensors goes straight to the launches.  Returns (cubes, reals, log)."""
        torch = nat.require_gpu()
        if self.placement_trials <= 1 or not (hasattr(cubes, "dim") and cubes.dim() == 4 and hasattr(reals, "dim") and reals.dim() == 4):
            return cubes, reals, {}

        def build(c, r):
            T = c.shape[0]
            return (c, r, eng.TileBatch([c[i] for i in range(T)], [r[i] for i in range(T)], None, self.table, self.deg, self.opts))

        def k1(cand):
            eng.batch_srf_integrate_moments(cand[2], self.min_valid, self.min_valid, self.tile_scale, self.tile_nodata)
        cand_bytes = cubes.numel() * cubes.element_size() + reals.numel() * 4
        (cubes, reals, tb), times = self._trials(build(cubes, reals), lambda: build(cubes.clone(), reals.clone()), k1,
                                                 candidate_bytes=cand_bytes)
        tb.placement_log = times
        if len(self._batches) >= 8:
            self._batches.pop(next(iter(self._batches)))
        self._batches[("stack", _tkey(cubes), _tkey(reals))] = tb
        return cubes, reals, {"joint_ms": times}

    # ---- one fit over several tiles on one GPU --------------------------------------------------------
    def fuse_mosaic(self, tiles, masks=None, k1_events=None, resident: bool = False):
        """Global fit over a mosaic held by ONE GPU (BASELINE configs[4] on a single device, and the single-process
        twin of the multi-GPU step): K1+K2 per tile, the per-tile moments added in tile order, one solve (plus the
        exchange when a process group is active, so that ranks holding several tiles each still fit one polynomial),
        K3 per tile.  ``tiles``: sequence of (cube, real) GPU tensors as in step(); ``masks`` optional sequence.
        Returns (coeffs (nb, deg+1), moments, [FusionOutput per tile]) - the outputs own their buffers.

        ``resident=True``: the same tiles (same tensors) will be fused again and again - the mosaic then runs through
        the batch machinery of step_batch() in five launches instead of four per tile: batched K1+K2, batched slot
        reduction, the sum over the tiles + solve, a broadcast of the coefficients, batched K3.  Per-tile slots and
        trees are those of the per-tile launches, so the per-tile moments carry the same bits; the outputs - images,
        moments and coefficients alike - are tensors of the batch, reused by the next call on the same tiles (clone what
        you keep)."""
        torch = nat.require_gpu()
        tiles = list(tiles)
        if not tiles:
            raise ValueError("fuse_mosaic needs at least one tile")
        masks = list(masks) if masks is not None else [None] * len(tiles)
        if len(masks) != len(tiles):
            raise ValueError("masks must match tiles")
        band_last = all(self._real_image(r, c.numel() // c.shape[-1])[1] == nat.PIXMAJOR for c, r in tiles)
        if resident and band_last and self.layout == nat.PIXMAJOR and len(tiles) <= 64:
            return self._fuse_mosaic_batched(tiles, masks, k1_events)
        moms = []
        pseudos = []
        for ti, ((cube, real), mask) in enumerate(zip(tiles, masks)):
            npix = cube.numel() // cube.shape[-1]
            real2, real_layout = self._real_image(real, npix)
            pseudo, mom = eng.srf_integrate_moments(cube, self.table, real2, self.deg, self.ws, mask, self.min_valid,
                                                    self.min_valid, out=None, events=k1_events if ti == 0 else None,
                                                    reduce=True, layout=self.layout,
                                                    real_layout=real_layout, scale=self.tile_scale, nodata=self.tile_nodata, opts=self.opts)
            moms.append(mom.clone())
            pseudos.append(pseudo)
        # the sum over the tiles by the same fixed-order reduction as the resident form (one "slot" per tile), so that both
        # forms of a mosaic give the same bits (a sequential sum here differed from it in the last bit: tools/dbg/stress_mosaic.py)
        total = eng.reduce_slots(torch.stack(moms), len(moms), self.table.nb, self.deg)
        if self._exchanges():
            total, coeffs = exchange_moments(total, self._solve, self.group, self.coeff_sync)
            coeffs = coeffs.clone()                    # _solve writes into the plan's workspace
        else:
            coeffs = eng.poly_solve(total, self.deg, self.min_count)
        outs = []
        for pseudo, mask in zip(pseudos, masks):
            matched = eng.poly_apply(pseudo, coeffs, mask if self.apply_mask else None, None, self.clip, self.layout,
                                     nb=self.table.nb)
            outs.append(FusionOutput(self.names, pseudo, total, coeffs, matched, self.layout))
        return coeffs, total, outs

    def _fuse_mosaic_batched(self, tiles, masks, k1_events):
        torch = nat.require_gpu()
        cubes, reals = [c for c, _ in tiles], [r for _, r in tiles]
        key = ("mosaic",) + tuple((_tkey(c), _tkey(r), None if m is None else _tkey(m)) for c, r, m in zip(cubes, reals, masks))
        tb = self._batches.get(key)
        if tb is None:
            if len(self._batches) >= 4:
                self._batches.pop(next(iter(self._batches)))
            tb = eng.TileBatch(cubes, reals, masks, self.table, self.deg, self.opts)
            self._batches[key] = tb
        eng.batch_srf_integrate_moments(tb, self.min_valid, self.min_valid, self.tile_scale, self.tile_nodata, events=k1_events)
        eng.batch_reduce_solve(tb, self.min_count)          # per-tile moments (the per-tile coefficients are not used)
        # the tiles' moments are T "slots" of the same [slot][band][moment] layout: one more fixed-order reduction
        # the sum over the tiles and its polynomial are written into the batch's own tensors (no copies out of the plan's
        # workspace), and K3 reads that ONE set for every tile (no broadcast copy): r03 trace of the 8-tile mosaic, three
        # 5 us copy kernels between the solve and K3
        total, coeffs = eng.reduce_solve_slots(tb.moments, tb.T, self.ws, self.min_count, tb.total_moments, tb.global_coeffs)
        if self._exchanges():
            total, coeffs = exchange_moments(total, lambda m: eng.poly_solve(m, self.deg, self.min_count, out=tb.global_coeffs),
                                             self.group, self.coeff_sync)
        eng.batch_poly_apply(tb, use_mask=self.apply_mask, clip=self.clip, coeffs=coeffs, shared=True)
        outs = [FusionOutput(self.names, tb.tile_rows(i, "pseudo"), total, coeffs, tb.tile_rows(i, "matched"), self.layout)
                for i in range(tb.T)]
        return coeffs, total, outs

    # ---- host -> device tile feed -------------------------------------------------------------------
    # once K1 runs at TB/s the host -> device feed of the 1.2 GB cube is the bottleneck of an
    # end-to-end run over tiles that live in host memory (the reference reads them from GeoTIFF / ENVI files).
    # stream() double-buffers the device inputs and issues the H2D copies on their own stream, so the copy of
    # tile i+1 (PCIe) runs under K1..K3 of tile i; results come back through pinned buffers.  Pass tiles as
    # pinned tensors (pinned_like / torch.Tensor.pin_memory) - pageable sources are first copied by the CPU
    # (~10 GB/s, five times slower than the link).
    @staticmethod
    def pinned_like(array):
        """Pinned host tensor with the shape and dtype of a NumPy array (fill it in place: .numpy())."""
        torch = nat.require_gpu()
        dt = {np.dtype(np.float32): torch.float32, np.dtype(np.uint16): torch.uint16, np.dtype(np.uint8): torch.uint8}[np.dtype(array.dtype)]
        t = torch.empty(tuple(array.shape), dtype=dt, pin_memory=True)
        t.numpy()[...] = array
        return t

    def stream(self, tiles, depth: int = 2, to_host: bool = True):
        """Run the hot path over an iterable of host tiles ``(cube, real)`` or ``(cube, real, mask)``
        (NumPy arrays or CPU tensors; cube float32 or uint16 (H,W,B)/(npix,B), real as in step()).
        ``cube`` may also be an ``emit_io.EnviCubeFile``: the BIL / BSQ / BIP file then goes file -> pinned staging
        -> GPU in file order and is transposed to pixel-major on the GPU (reference loader s2_emit/emit_io.py:7-16
        does the transpose on the host).
        Yields ``(index, coeffs, matched, out)``: with ``to_host`` coeffs is a (nb, deg+1) float64 and matched a
        float32 NumPy array in the plan's layout (pinned memory reused every ``depth`` tiles - copy what you
        keep); otherwise the device tensors of ``out`` (valid until the next iteration)."""
        torch = nat.require_gpu()
        if depth < 1:
            raise ValueError("depth must be >= 1")
        dev = self.device
        copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        from .emit_io import EnviCubeFile
        slots = [dict(cube=None, real=None, mask=None, ready=torch.cuda.Event(), free=torch.cuda.Event(), h2d=torch.cuda.Event(), used=False,
                      coeffs_h=None, matched_h=None, done=torch.cuda.Event()) for _ in range(depth)]

        def as_tensor(x):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
            return t if t.is_contiguous() else t.contiguous()

        def upload(i, tile):
            sl = slots[i % depth]
            src_cube = tile[0]
            real = as_tensor(tile[1])
            mask = as_tensor(tile[2]) if len(tile) > 2 and tile[2] is not None else None
            if mask is not None and mask.dtype == torch.bool:
                mask = mask.view(torch.uint8)
            envi = isinstance(src_cube, EnviCubeFile)
            if envi:
                # BIL / BSQ / BIP file: file -> pinned staging in FILE order (the host's only pass over the samples),
                # H2D of the raw bytes, transpose on the GPU (hsr_interleave_to_bip), all on the copy stream
                st = sl.get("stage")
                if st is None or st.shape[0] != src_cube.raw.shape[0] or st.dtype != src_cube._torch_dtype(torch):
                    if sl["used"]:
                        sl["free"].synchronize()        # the H2D copy out of the old staging buffer has finished
                    st = sl["stage"] = src_cube.new_staging(torch)
                elif sl["used"]:
                    sl["h2d"].synchronize()             # previous copy out of this staging buffer has finished
                src_cube.stage(st)
            else:
                cube = as_tensor(src_cube)
            with torch.cuda.stream(copy_stream):
                if sl["used"]:
                    copy_stream.wait_event(sl["free"])       # the step that last read this buffer set
                if envi:
                    raw = sl.get("raw")
                    if raw is None or raw.shape != st.shape or raw.dtype != st.dtype:
                        raw = sl["raw"] = torch.empty(st.shape, dtype=st.dtype, device=dev)
                    raw.copy_(st, non_blocking=True)
                    sl["h2d"].record(copy_stream)
                    odt = src_cube.cube_dtype(torch)
                    cube_dev = sl["cube"]
                    if src_cube.interleave == "bip" or cube_dev is None or tuple(cube_dev.shape) != src_cube.shape or cube_dev.dtype != odt:
                        cube_dev = None
                    sl["cube"] = src_cube.to_bip(raw, out=cube_dev)
                pairs = (("real", real), ("mask", mask)) if envi else (("cube", cube), ("real", real), ("mask", mask))
                for key, src in pairs:
                    if src is None:
                        sl[key] = None
                        continue
                    if sl[key] is None or sl[key].shape != src.shape or sl[key].dtype != src.dtype:
                        sl[key] = torch.empty(src.shape, dtype=src.dtype, device=dev)
                    sl[key].copy_(src, non_blocking=True)
                sl["ready"].record(copy_stream)
            sl["used"] = True

        it = iter(tiles)
        pending = []
        idx = 0
        for _ in range(depth):                       # prime: the first copies start before any compute
            tile = next(it, None)
            if tile is None:
                break
            upload(idx, tile)
            pending.append(idx)
            idx += 1
        while pending:
            i = pending.pop(0)
            sl = slots[i % depth]
            main.wait_event(sl["ready"])
            out = self.step(sl["cube"], sl["real"], None if sl["mask"] is None else sl["mask"].reshape(-1))
            if to_host:
                if sl["matched_h"] is None or sl["matched_h"].shape != out.matched.shape:
                    sl["matched_h"] = torch.empty(out.matched.shape, dtype=torch.float32, pin_memory=True)
                    sl["coeffs_h"] = torch.empty(out.coeffs.shape, dtype=torch.float64, pin_memory=True)
                sl["matched_h"].copy_(out.matched, non_blocking=True)
                sl["coeffs_h"].copy_(out.coeffs, non_blocking=True)
            sl["free"].record(main)                  # inputs of this slot may be overwritten from here on
            sl["done"].record(main)
            tile = next(it, None)                    # refill the slot just consumed: its copy overlaps the next step
            if tile is not None:
                upload(idx, tile)
                pending.append(idx)
                idx += 1
            if to_host:
                sl["done"].synchronize()
                yield i, sl["coeffs_h"].numpy(), sl["matched_h"].numpy(), out
            else:
                yield i, out.coeffs, out.matched, out

    # ---- one-tile-deep software pipeline -----------------------------------------------------------
    # The fit of tile i (slot reduction -> RCCL exchange -> solve: a few small kernels and a latency-bound
    # collective) runs on a side stream underneath K1 of tile i+1 ("overlap the collective of tile
    # i with K1 of tile i+1"); K3 of tile i stays on the caller's stream, enqueued right behind K1 of tile i+1:
    #
    #     caller's stream :  K1(0)  K1(1)  K3(0)  K1(2)  K3(1)  K1(3)  K3(2) ...
    #     side stream     :  fit(0)        fit(1)        fit(2)        fit(3) ...      (fit(i) under K1(i+1))
    #
    # so every bandwidth-bound kernel has the whole chip to itself and the exchange latency (and, on one GPU, the
    # reduce+solve launch) disappears from the critical path.  K1's persistent workgroups own every CU they run on,
    # so CUs are left free for the side stream (hsr_set_srf_reserved_cus).  Measured (tools/corun_probe.py): a kernel
    # from another queue is only dispatched next to the persistent K1 if every XCD has a completely free CU - with 8
    # free CUs (496 workgroups) the reduce + solve pair runs in 25 us under K1, with 6 or fewer it waits for K1 to end.  The side stream is created with high priority: streams of the default priority were mapped to the
    # SAME hardware queue as the caller's stream on this stack (rocprofv3 Queue_Id), which serialised everything.
    # The first version also sent K3 to the side stream: K1 and K3 then fought for HBM and a tile took
    # 0.265 ms against 0.256 ms sequential.  Two alternating buffer sets; ordering between tiles i and i+2 needs no
    # extra events: K3(i) waits for fit(i) and precedes K1(i+2) on the caller's stream.  The FusionOutput returned
    # for tile i is valid until the second submit() after it.
    def _pick_side_stream(self, slots, cube, real2, mask, steps: int = 10):
        """The stream the fits of submit() run on, chosen by MEASUREMENT.  Which hardware queue a HIP stream is served by is
        the runtime's business (a few queues shared round-robin by all streams of the process), and it decides everything
        here (profiles/r03_strong_scaling.md): a default-priority stream may share the caller's queue (the fit then simply
        serialises with K1), and some high-priority streams of torch's pool make every pipelined step of a small tile take
        ~200 us instead of ~55 us (the 4th and 5th stream of the pool, reproducibly, in every process tried).  So a few
        candidates - high-priority streams first, one of default priority last - each run a short local pipeline over the
        caller's tile (results are overwritten by the first real submit) and the fastest is kept; ties go to the earlier
        candidate.  ~4 x 10 steps, once per tile shape; pass ``side_stream=`` to the constructor to skip it."""
        torch = nat.require_gpu()
        import ctypes as C
        import time
        lib = nat.load()
        cands = [torch.cuda.Stream(device=self.device, priority=-1) for _ in range(3)] + [torch.cuda.Stream(device=self.device, priority=0)]
        fin = C.c_int32(-1)
        mptr = None if mask is None else mask.data_ptr()
        times = []
        with eng._launch(cube) as stream:
            for cand in cands:
                ph = C.c_void_p()
                nat.check(lib.hsr_pipeline_create(slots[0]["plan"], slots[1]["plan"], C.c_void_p(cand.cuda_stream), 0, C.byref(ph)),
                          "hsr_pipeline_create")
                try:
                    best = float("inf")
                    for rep in range(2):                     # first repetition warms the queue up
                        torch.cuda.synchronize(self.device)
                        t0 = time.perf_counter()
                        for _ in range(steps):
                            nat.check(lib.hsr_pipeline_submit(ph, cube.data_ptr(), real2.data_ptr(), mptr, mptr, stream, C.byref(fin), None, None),
                                      "hsr_pipeline_submit")
                        nat.check(lib.hsr_pipeline_flush(ph, mptr, stream, C.byref(fin)), "hsr_pipeline_flush")
                        torch.cuda.synchronize(self.device)
                        best = min(best, time.perf_counter() - t0)
                    times.append(best / steps)
                finally:
                    lib.hsr_pipeline_destroy(ph)
        pick = min(range(len(cands)), key=lambda i: (times[i] > 1.03 * min(times), i))     # first one within 3 % of the best
        self.side_stream_log = [round(t * 1e6, 2) for t in times]
        return cands[pick]

    def _pipe_build(self, cube, real, mask, key):
        """Buffers (two slots), prepared launches and the native pipeline for tiles shaped like ``cube`` / ``real``."""
        torch = nat.require_gpu()
        import ctypes as C
        lib = nat.load()
        if self._pipe is not None:
            # tiles of the previous shape still in flight: finish them and keep their outputs - the next submit() / drain() calls
            # return them first, in order (round 3 dropped them here: a mosaic with a ragged last tile lost one or two tiles)
            self._backlog = self.drain()               # (drain() hands over what the backlog already held, in front)
            old, self._pipe = self._pipe, None
            self._native_handles = [e for e in self._native_handles if not any(e[1] is h for _, h in old["handles"])]
            self._destroy_handles(old["handles"])
        npix = cube.numel() // cube.shape[-1]
        real2, real_layout = self._real_image(real, npix)
        if not (real2.is_cuda and real2.dtype == torch.float32):
            raise ValueError("real must be a float32 GPU image with one value per pixel and band")
        eng._as_cube2d(cube)                               # dtype / contiguity / device checks of the operators
        nb = self.table.nb

        def probe(img):
            eng.srf_integrate_moments(cube, self.table, real2, self.deg, self.ws, mask, self.min_valid, self.min_valid,
                                      out=img, reduce=False, layout=self.layout, real_layout=real_layout,
                                      scale=self.tile_scale, nodata=self.tile_nodata, opts=self.opts)
        placed = list(self._pipe_images.pop(npix, []))     # placed together with the resident inputs (place_inputs)
        if len(placed) < 2:
            placed = [eng.alloc_image(torch, nb, npix, self.layout, self.device) for _ in range(2)]
            if self.placement_trials > 1 and npix >= (1 << 16):
                placed = self._place(npix, placed, probe, count=2)       # ONE search for both slots' images
        # Three forms: two slots (fit on the side stream, K3 as its own launch); fused, three slots (one kernel per tile, fit in the
        # tail: no exchange possible); fused with the exchange issued from C, four slots (hsr_pipeline_create_exchange).
        # The C side decides whether a K1 launch of this geometry can carry the older tile's K3 (hsr_srf_fused_launch_supported, called
        # by the create functions); what it cannot see is the cube pointer - uint16 tiles need the 16-byte aligned loader.
        exchange = self._exchanges()
        if self.group_tiles > 1 and (exchange or not self.fuse_apply or self.layout != nat.PIXMAJOR):
            raise ValueError("group_tiles > 1 needs fuse_apply=True, the pixel-major layout and no exchange (multi-rank mosaics: fuse_mosaic)")
        fused = (self.fuse_apply and self.layout == nat.PIXMAJOR and
                 (cube.dtype == torch.float32 or (cube.data_ptr() % 16 == 0 and not self.u16_single_buffer)))
        grouped = fused and not exchange and self.group_tiles > 1
        nslots = (self.group_tiles + 2 if grouped else (4 if exchange else 3)) if fused else 2
        while len(placed) < nslots:
            placed.append(eng.alloc_image(torch, nb, npix, self.layout, self.device))
        slots, outs, handles = [], [], []
        placed_m = list(self._pipe_matched.pop(npix, []))
        for k in range(nslots):
            ws = eng.MomentWorkspace(self.device, nb, self.deg)
            matched = placed_m[k] if k < len(placed_m) else eng.alloc_image(torch, nb, npix, self.layout, self.device)
            h, keep = self._native_plan(cube, real2, real_layout, placed[k], matched, ws)
            handles.append(("plan", h))
            slots.append(dict(plan=h, ws=ws, keep=keep, mask=None))
            outs.append(FusionOutput(self.names, placed[k], ws.moments, ws.coeffs, matched, self.layout))
        if self.side_stream is not None:
            side = self.side_stream
        elif fused and not exchange:                   # the three-slot pipeline has no side-stream work: nothing to choose
            side = torch.cuda.Stream(device=self.device)
        elif fused:
            # gate -> collective -> solve run under K1 on this stream: chosen by MEASUREMENT like the two-slot pipeline's (some of a
            # process's high-priority streams are served by a hardware queue on which side work next to the persistent K1 costs
            # ~100 us per step: r04 shard curve, 338 us per 1024-row step on the 5th such stream of the process against 229)
            side = self._pick_side_stream(slots, cube, real2, mask)
        else:
            side = self._pick_side_stream(slots, cube, real2, mask)
        ph = C.c_void_p()
        transport = None
        gbuf = None
        if grouped:
            T, M = self.group_tiles, 3 * self.deg + 2
            gbuf = (torch.zeros((2, T, nb, M), dtype=torch.float64, device=self.device), torch.zeros((2, nb, M), dtype=torch.float64, device=self.device),
                    torch.zeros((2, nb, self.deg + 1), dtype=torch.float64, device=self.device))
            arr = (C.c_void_p * nslots)(*[sl["plan"] for sl in slots])
            rc = lib.hsr_pipeline_create_group(arr, nslots, T, gbuf[0].data_ptr(), gbuf[1].data_ptr(), gbuf[2].data_ptr(),
                                               C.c_void_p(side.cuda_stream), C.byref(ph))
        elif fused and exchange:
            x, transport = self._exchange_desc()
            arr = (C.c_void_p * 4)(*[sl["plan"] for sl in slots])
            rc = lib.hsr_pipeline_create_exchange(arr, C.c_void_p(side.cuda_stream), C.byref(x), C.byref(ph))
        elif fused:
            rc = lib.hsr_pipeline_create_fused(slots[0]["plan"], slots[1]["plan"], slots[2]["plan"], C.c_void_p(side.cuda_stream),
                                               0, C.byref(ph))
        if grouped and rc != nat.HSR_OK:
            nat.check(rc, "hsr_pipeline_create_group")      # a group fit has no two-slot form to fall back to
        if fused and rc != nat.HSR_OK:                 # geometry the fused launch does not cover: the two-slot pipeline
            fused, transport = False, None
            self.fused_fallback = nat.load().hsr_last_error().decode("utf-8", "replace")
            for kind, h in reversed(handles[2:]):
                lib.hsr_step_plan_destroy(h)
            self._native_handles = [e for e in self._native_handles if not any(e[1] is h for _, h in handles[2:])]
            slots, outs, handles = slots[:2], outs[:2], handles[:2]
            if self.side_stream is None:
                side = self._pick_side_stream(slots, cube, real2, mask)
        if not fused:
            nat.check(lib.hsr_pipeline_create(slots[0]["plan"], slots[1]["plan"], C.c_void_p(side.cuda_stream), 1 if exchange else 0,
                                              C.byref(ph)), "hsr_pipeline_create")
        self._native_handles.append(("pipe", ph))
        handles.append(("pipe", ph))
        self._pipe = dict(key=key, npix=npix, h=ph, slots=slots, outs=outs, side=side, side_handle=C.c_void_p(side.cuda_stream),
                          exchange=exchange, n=0, fin=C.c_int32(-1), fused=fused, S=len(slots), inflight=[], handles=handles,
                          c_exchange=bool(fused and exchange), transport=transport, group=gbuf, done=0)
        return self._pipe

    @staticmethod
    def _event_handle(ev, stream):
        """Raw hipEvent_t of a torch event (torch creates it lazily, on the first record)."""
        if not ev.cuda_event:
            ev.record(stream)
        import ctypes as C
        return C.c_void_p(ev.cuda_event)

    def submit(self, cube, real, mask=None, k1_events=None) -> Optional[FusionOutput]:
        """Pipelined step: start tile i, finish and return tile i-1 (None on the first call) - tile i-2 with
        ``fuse_apply=True`` (None on the first two calls), tile i-3 with ``fuse_apply=True`` and an exchange.  One call into the native
        pipeline (csrc/hsr_exec.hip) enqueues K1(i) and K3(i-1) on the caller's stream and - without an exchange - the fit
        of tile i on the side stream; with an exchange the fit (slot reduction -> collective -> solve) is enqueued from
        here on the side stream."""
        torch = nat.require_gpu()
        import ctypes as C
        lib = nat._lib or nat.load()
        key = (tuple(cube.shape), cube.dtype, tuple(real.shape), tuple(real.stride()), real.dtype,
               cube.dtype == torch.float32 or cube.data_ptr() % 16 == 0)   # (the fused launch of uint16 tiles needs the aligned loader)
        st = self._pipe
        if st is None or st["key"] != key:
            st = self._pipe_build(cube, real, mask, key)
        if not (cube.is_contiguous() and cube.device == self.device and real.device == self.device):
            raise ValueError("cube must be a contiguous tensor on the plan's GPU, real on the same GPU")
        if mask is not None and not (mask.dtype == torch.uint8 and mask.numel() == st["npix"] and mask.is_contiguous()
                                     and mask.device == self.device):
            raise ValueError("mask must be a contiguous uint8 tensor with one byte per pixel")
        S = st["S"]
        cur = st["n"] % S
        # the tile this call finishes: i-1 (two slots) or i-2 (fused) - the oldest one in flight, once S - 1 are
        prev_mask = st["slots"][st["inflight"][0]]["mask"] if len(st["inflight"]) == S - 1 else None
        fin = st["fin"]
        with eng._launch(cube) as stream:
            e0 = e1 = None
            if k1_events is not None:
                ts = torch.cuda.current_stream(self.device)
                e0, e1 = self._event_handle(k1_events[0], ts), self._event_handle(k1_events[1], ts)
            nat.check(lib.hsr_pipeline_submit(st["h"], cube.data_ptr(), real.data_ptr(), None if mask is None else mask.data_ptr(),
                                              None if prev_mask is None else prev_mask.data_ptr(), stream, C.byref(fin), e0, e1),
                      "hsr_pipeline_submit")
            slot = st["slots"][cur]
            slot["mask"] = mask                        # K3 of this tile reads it one submit() later
            slot["ws"].slots = lib.hsr_step_plan_slots(slot["plan"])
            if st["exchange"] and not st["c_exchange"]:      # two slots: the collective goes through torch.distributed from here
                sh = st["side_handle"]
                with torch.cuda.stream(st["side"]):
                    nat.check(lib.hsr_step_run_reduce(slot["plan"], sh), "hsr_step_run_reduce")

                    def solve(_m, _p=slot["plan"], _ws=slot["ws"]):
                        nat.check(lib.hsr_step_run_solve(_p, sh), "hsr_step_run_solve")
                        return _ws.coeffs
                    exchange_moments(slot["ws"].moments, solve, self.group, self.coeff_sync)
                nat.check(lib.hsr_pipeline_fit_done(st["h"]), "hsr_pipeline_fit_done")
        st["n"] += 1
        st["inflight"].append(cur)
        out = None
        if fin.value >= 0:
            st["inflight"].remove(fin.value)
            st["slots"][fin.value]["mask"] = None
            out = self._finished(st, fin.value)
        if self._backlog:                        # tiles a pipeline rebuild finished come first
            if out is not None:
                self._backlog.append(out)
            out = self._backlog.pop(0)
        return out

    def _finished(self, st, slot) -> FusionOutput:
        """Output of the tile the pipeline has just finished (tiles finish in submission order)."""
        out = st["outs"][slot]
        if st["group"] is not None:                 # a group fit: the tile carries its GROUP's moments and polynomial
            par = (st["done"] // self.group_tiles) & 1
            out = FusionOutput(out.names, out.pseudo, st["group"][1][par], st["group"][2][par], out.matched, out.layout)
        st["done"] += 1
        return out

    def pipeline_status(self) -> int:
        """Exchange pipelines (fuse_apply with an exchange): synchronise both streams and return what the device-side polls and the
        host transport recorded - 0, or 1 / 2 (a poll hit its 20 s limit: moments gate / coefficients of a K3), + 16 (the host
        callback failed).  0 for the other pipelines."""
        st = self._pipe
        if st is None or not st.get("c_exchange"):
            return 0
        import ctypes as C
        code = C.c_uint32(0)
        with eng._launch(st["outs"][0].pseudo) as stream:
            nat.check(nat.load().hsr_pipeline_status(st["h"], stream, C.byref(code)), "hsr_pipeline_status")
        return int(code.value)

    def drain(self) -> List[FusionOutput]:
        """Finish (K3, on the caller's stream) every tile still in the pipeline, oldest first, and return their outputs."""
        st = self._pipe
        outs, self._backlog = self._backlog, []
        if st is None or st["n"] == 0:
            return outs
        import ctypes as C
        lib = nat._lib or nat.load()
        fin = st["fin"]
        with eng._launch(st["outs"][0].pseudo) as stream:
            while st["inflight"]:
                slot = st["inflight"][0]                 # oldest first; its mask was kept with the slot
                mask = st["slots"][slot]["mask"]
                try:
                    nat.check(lib.hsr_pipeline_flush(st["h"], None if mask is None else mask.data_ptr(), stream, C.byref(fin)),
                              "hsr_pipeline_flush")
                except nat.HsrError:
                    self._backlog = outs                 # what has been finished so far is not lost
                    raise
                if fin.value != slot:
                    raise nat.HsrError(f"pipeline out of step: expected slot {slot}, library finished {fin.value}")
                st["inflight"].pop(0)
                st["slots"][slot]["mask"] = None
                outs.append(self._finished(st, slot))
        return outs

    def flush(self) -> Optional[FusionOutput]:
        """Finish the tile(s) left in the pipeline by the last submit() and return the LAST one (drain() returns all)."""
        outs = self.drain()
        return outs[-1] if outs else None


def fuse_pair(R, emit_w, srf_dict, good_mask, real_s2: Dict[str, np.ndarray], deg: int = 3,
              min_valid: Optional[float] = 0.0, min_count: int = 50, clip: bool = True):
    """NumPy convenience wrapper: one EMIT cube + real S2 planes (dict band -> (H,W)) on the same
    grid -> (pseudo dict, coeffs dict, matched dict), all host arrays.  Bands without SRF support map
    to None like pseudo_s2_srf_integral."""
    torch = nat.require_gpu()
    plan = SpectralFusion(emit_w, srf_dict, good_mask, deg, min_valid, min_count, clip)
    H, W = R.shape[:2]
    cube = torch.from_numpy(np.ascontiguousarray(R, dtype=np.float32)).to(plan.device)
    real = torch.from_numpy(np.stack([np.asarray(real_s2[b], dtype=np.float32) for b in plan.names])).to(plan.device)
    out = plan.step(cube, real.reshape(len(plan.names), -1), reuse_buffers=False)
    ps, co, ma = out.planes("pseudo").cpu().numpy(), out.coeffs.cpu().numpy(), out.planes("matched").cpu().numpy()
    pseudo = {b: None for b in plan.table.names}
    coeffs = {b: None for b in plan.table.names}
    matched = {b: None for b in plan.table.names}
    for i, b in enumerate(plan.names):
        pseudo[b], coeffs[b], matched[b] = ps[i].reshape(H, W), co[i], ma[i].reshape(H, W)
    return pseudo, coeffs, matched


def calibrate_pseudo_to_real_linear(pseudo_stack, real_stack, valid_mask, min_valid=0.0):
    """Per-band linear least squares over all valid pixels (Pairs_EMIT_S2_demo-2.ipynb cell 72):
    returns (corrected (nb,H,W) float32, [(a, b)] * nb).  NumPy in / NumPy out, computed on the GPU."""
    torch = nat.require_gpu()
    nb, H, W = pseudo_stack.shape
    x = torch.from_numpy(np.ascontiguousarray(pseudo_stack, dtype=np.float32)).cuda().reshape(nb, -1)
    y = torch.from_numpy(np.ascontiguousarray(real_stack, dtype=np.float32)).cuda().reshape(nb, -1)
    m = torch.from_numpy(np.ascontiguousarray(valid_mask, dtype=np.bool_).view(np.uint8)).cuda().reshape(-1)
    ws = eng.MomentWorkspace(x.device, nb, 1)
    mom = eng.poly_moments(x, y, 1, ws, m, float(min_valid), float(min_valid), layout=nat.PLANAR)
    coeffs = eng.poly_solve(mom, 1, 50)
    corrected = eng.poly_apply(x, coeffs, None, None, clip=False, layout=nat.PLANAR)
    params = [(float(a), float(b)) for a, b in coeffs.cpu().numpy()]
    return corrected.reshape(nb, H, W).cpu().numpy(), params


_MP_SIDE = {}


def _match_pair_side_stream(torch, dev, which: int = 0):
    """Side streams of match_pair (created once per device): 0 = the 10 m producer chain, 1 = one of the two 60 m selects."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), which)
    st = _MP_SIDE.get(key)
    if st is None:
        st = _MP_SIDE[key] = torch.cuda.Stream(device=dev)
    return st


def match_pair(R, emit_w, srf_dict, good_mask, s2_rgb_hi, factor: int = 6, deg: int = 4, use_ot: bool = True,
               n_samples: int = 5000, reg: float = 0.05, numItermax: int = 300, stopThr: float = 1e-6, seed: int = 0,
               src_scale: float = 1.0 / 255.0, rgb_bands=("B4", "B3", "B2"), positive_band: str = "B2",
               as_numpy: bool = True):
    """The reference driver (s2_emit/poly_regression.py:96-172 == notebook cell 81) on in-memory,
    grid-aligned arrays, device resident:

      phase 1  SRF integration of the three RGB bands (K1), valid60 = finite & (B2 > 0)        :101-106
      phase 2  real S2 RGB (Hh,Wh,3) uint8/float -> 60 m by an f x f block mean, * src_scale   :110-118
      phase 3  shared 2/98 percentile stretch of both, fit, apply at 60 m                      :122-139
               use_ot=True : fit_ot_poly_rgb (Sinkhorn targets, host PCG64 sampling -> one sync)
               use_ot=False: per-channel least squares over all valid pixels (no sync)
      phase 4  bilinear x f to the S2 grid, stretch inside the finite mask, apply at 10 m      :150-162

    R (H,W,B) float32 (NumPy or GPU tensor); s2_rgb_hi (H*f, W*f, 3).  Returns a dict with
    ``coeffs`` (3, deg+1), ``emit_rgb_matched_60m`` (H,W,3), ``s2_rgb_60m_n`` (H,W,3), ``valid60`` (H,W),
    ``emit_rgb_10m_matched`` (H*f, W*f, 3), ``mask10`` - NumPy arrays, or GPU tensors if as_numpy=False.
    The two resampling steps replace GDAL's reproject for aligned grids (parity unpinned, see hsr.h).
    """
    torch = nat.require_gpu()
    from .poly_regression import fit_ot_poly_rgb
    dev = torch.device("cuda", torch.cuda.current_device())
    sub = {}
    for b in rgb_bands:
        if b not in srf_dict:
            raise ValueError(f"Band {b} is None/missing in pseudo_s2.")
        sub[b] = srf_dict[b]
    table = eng.build_srf_table(emit_w, sub, good_mask)
    if table.supported != list(rgb_bands):
        missing = [b for b in rgb_bands if b not in table.supported][0]
        raise ValueError(f"Band {missing} is None/missing in pseudo_s2.")
    cube = R if type(R).__module__.startswith("torch") else torch.from_numpy(np.ascontiguousarray(R, dtype=np.float32))
    cube = cube.to(dev, torch.float32).contiguous()
    H, W = int(cube.shape[0]), int(cube.shape[1])
    s2 = s2_rgb_hi if type(s2_rgb_hi).__module__.startswith("torch") else torch.from_numpy(np.ascontiguousarray(s2_rgb_hi))
    s2 = s2.to(dev).contiguous()
    if s2.dtype not in (torch.uint8, torch.uint16, torch.float32):
        s2 = s2.to(torch.float32)
    if tuple(s2.shape) != (H * factor, W * factor, 3):
        raise ValueError(f"s2_rgb_hi must be ({H * factor},{W * factor},3) for factor {factor}; got {tuple(s2.shape)}")
    PM = nat.PIXMAJOR
    emit_rgb = eng.srf_integrate(cube, table, layout=PM)                                   # (npix, 4): R,G,B,pad
    # r04: the 10 m producer chain (upsample -> finite mask -> percentile limits: ~430 us of streaming kernels) needs emit_rgb only, the
    # 60 m phase below (two launch-bound selects, moments, solve: ~250 us of mostly idle GPU) needs it too and nothing else of it - they
    # run on two streams and meet at the last apply.  No host synchronisation; the same kernels on the same inputs, the same bits.
    cur = torch.cuda.current_stream(dev)
    side = _match_pair_side_stream(torch, dev)
    ready = torch.cuda.Event()
    ready.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        # the upsampling kernel writes the mask and counts the first radix pass of the select while it holds the values (same bits as
        # bilinear_upsample + valid_mask + percentile_limits, two reads of the 10 m image fewer)
        emit_rgb_10, mask10, lohi10 = eng.bilinear_upsample_mask_limits(emit_rgb, H, W, factor, 2, 98, nb=3)
    s2_60 = eng.block_mean(s2.reshape(-1, 3), H, W, factor, src_scale, layout=PM, nb=3)     # (npix, 4)
    pos = list(rgb_bands).index(positive_band)
    valid60 = eng.valid_mask(emit_rgb, pos, s2_60, None, PM, nbx=3, nby=3)
    # the two 60 m selects are launch-bound (nine launches each for 16 MB): side by side on two streams
    side2 = _match_pair_side_stream(torch, dev, 1)
    ready2 = torch.cuda.Event()
    ready2.record(cur)
    with torch.cuda.stream(side2):
        side2.wait_event(ready2)
        lohi_e = eng.percentile_limits(emit_rgb, valid60, 2, 98, PM, nb=3)
    lohi_s = eng.percentile_limits(s2_60, valid60, 2, 98, PM, nb=3)
    cur.wait_stream(side2)
    lohi_e.record_stream(cur)
    s2_n = eng.poly_apply_stretch_only(s2_60, lohi_s, PM, nb=3)
    if use_ot:
        emit_n = eng.poly_apply_stretch_only(emit_rgb, lohi_e, PM, nb=3)
        co = fit_ot_poly_rgb(emit_n[:, :3].reshape(H, W, 3), s2_n[:, :3].reshape(H, W, 3), valid60.reshape(H, W),
                             deg=deg, n_samples=n_samples, reg=reg, numItermax=numItermax, stopThr=stopThr, seed=seed)
        coeffs = torch.from_numpy(np.ascontiguousarray(co)).to(dev)
    else:
        ws = eng.MomentWorkspace(dev, 3, deg)
        mom = eng.poly_moments(emit_rgb, s2_60, deg, ws, valid60, lohi_x=lohi_e, lohi_y=lohi_s, layout=PM, nb=3)
        coeffs = eng.poly_solve(mom, deg, 200).clone()
    matched60 = eng.poly_apply(emit_rgb, coeffs, valid60, lohi_e, True, PM, nb=3)
    cur.wait_stream(side)
    for t_ in (emit_rgb_10, mask10, lohi10):
        t_.record_stream(cur)
    matched10 = eng.poly_apply(emit_rgb_10, coeffs, mask10, lohi10, True, PM, nb=3)
    Hh, Wh = H * factor, W * factor
    res = dict(coeffs=coeffs, emit_rgb_matched_60m=matched60[:, :3].reshape(H, W, 3),
               s2_rgb_60m_n=s2_n[:, :3].reshape(H, W, 3), valid60=valid60.reshape(H, W).bool(),
               emit_rgb_10m_matched=matched10[:, :3].reshape(Hh, Wh, 3), mask10=mask10.reshape(Hh, Wh).bool(),
               lohi_emit_60m=lohi_e, lohi_s2_60m=lohi_s, lohi_emit_10m=lohi10)
    if as_numpy:
        res = {k: v.cpu().numpy() for k, v in res.items()}
    return res
