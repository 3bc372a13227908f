"""ctypes binding of libhsr_mi355x.so (C ABI: include/hsr.h).

The HIP library is the product path.  There is deliberately NO CPU fallback: if the shared
library is missing, or no MI355X-class device is visible, every compute entry point raises
``HsrUnavailable`` loudly instead of silently computing something else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_SO = os.path.join(os.path.dirname(_HERE), "lib", "libhsr_mi355x.so")

HSR_OK = 0
HSR_MAX_BANDS = 16
HSR_MAX_DEG = 4
HSR_MAX_APPLY_DEG = 8
HSR_MAX_SPECTRAL = 560
HSR_TILE_PIXELS = 64
HSR_SRF_U16_FAST = 1
HSR_MAX_PARTIALS = 4096
HSR_FIT_GROUPS = 64
HSR_FIT_TICKETS = 65
PLANAR = "planar"        # band-major planes: tensor (nb, npix), unit pixel stride
PIXMAJOR = "pixmajor"    # pixel-major / band-last: tensor (npix, row) with row >= nb, unit band stride


class HsrUnavailable(RuntimeError):
    """The HIP extension (or a GPU to run it on) is not available."""


class HsrError(RuntimeError):
    """A library call returned an error code."""


_i32, _i64, _f32, _f64, _vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p
_pi32 = C.POINTER(C.c_int32)


class SrfOptions(C.Structure):
    """hsr_srf_options (include/hsr.h): per-call tuning of the K1 launches; the library keeps no tuning state."""
    _fields_ = [("tile_pixels", _i32), ("reserved_cus", _i32), ("u16_single_buffer", _i32), ("flags", _i32)]


class BatchTile(C.Structure):
    """hsr_batch_tile: one tile of a batch (64 bytes)."""
    _fields_ = [("cube_dev", _vp), ("real_dev", _vp), ("mask_dev", _vp), ("pseudo_dev", _vp), ("matched_dev", _vp),
                ("npix", _i64), ("slot0", _i64), ("slots", _i32), ("ngroups", _i32)]


class BatchInfo(C.Structure):
    """hsr_batch_info: what hsr_batch_plan found."""
    _fields_ = [("nunits", _i64), ("total_pixels", _i64), ("max_npix", _i64), ("ntiles", _i32), ("aligned16", _i32)]


class FusedFit(C.Structure):
    """hsr_fused_fit: workspaces and outputs of the reduce + solve folded into K1 (hsr_srf_integrate_fit)."""
    _fields_ = [("group_partials_dev", _vp), ("tickets_dev", _vp), ("moments_dev", _vp), ("coeffs_dev", _vp),
                ("min_count", _i64)]


class StepDesc(C.Structure):
    """hsr_step_desc: every argument of one tile's step for fixed output / workspace buffers (include/hsr.h, ABI 4)."""
    _fields_ = [("cube_dtype", _i32), ("B", _i32), ("npix", _i64), ("scale", _f32), ("nodata", _i32),
                ("wn_dev", _vp), ("k0", _pi32), ("klen", _pi32), ("nb", _i32), ("deg", _i32),
                ("pseudo_dev", _vp), ("out_bs", _i64), ("out_ps", _i64), ("real_bs", _i64), ("real_ps", _i64),
                ("min_x", _f32), ("min_y", _f32), ("partials_dev", _vp), ("moments_dev", _vp), ("coeffs_dev", _vp),
                ("min_count", _i64), ("matched_dev", _vp), ("matched_bs", _i64), ("matched_ps", _i64),
                ("apply_mask", _i32), ("clip", _i32), ("opts", SrfOptions)]


HSR_ABI_VERSION = 5
HSR_COMM_ID_BYTES = 128
HSR_SYNC_ALLREDUCE, HSR_SYNC_BROADCAST = 1, 2
# hsr_host_sum_fn: int (*)(void* user, double* values, int32_t count) - the host transport of an exchange pipeline
HOST_SUM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)


class Exchange(C.Structure):
    """hsr_exchange: how the fit crosses the ranks in hsr_pipeline_create_exchange (an RCCL communicator or a host callback)."""
    _fields_ = [("comm", _vp), ("mode", _i32), ("root", _i32), ("host_sum", HOST_SUM_FN), ("host_user", _vp),
                ("rehearsal_us", _i32), ("rehearsal_blocks", _i32)]


BATCH_RECORD_BYTES = 64          # sizeof(hsr_batch_tile) == sizeof(hsr_batch_unit)
assert C.sizeof(BatchTile) == BATCH_RECORD_BYTES
_popt = C.POINTER(SrfOptions)

# name -> (restype, argtypes); must list every symbol include/hsr.h declares
SIGNATURES = {
    "hsr_abi_version": (C.c_int, []),
    "hsr_last_error": (C.c_char_p, []),
    "hsr_moment_count": (C.c_int, [_i32]),
    "hsr_partial_slots": (C.c_int, [_i64, _popt]),
    "hsr_partials_bytes": (C.c_size_t, [_i32, _i32]),
    "hsr_srf_integrate": (C.c_int, [_vp, _i64, _i32, _vp, _pi32, _pi32, _i32, _vp, _i64, _i64, _popt, _vp]),
    "hsr_srf_integrate_moments": (C.c_int, [_vp, _i64, _i32, _vp, _pi32, _pi32, _i32, _vp, _i64, _i64,
                                            _vp, _i64, _i64, _vp, _f32, _f32, _i32, _vp, _pi32, _popt, _vp]),
    "hsr_poly_moments": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i32, _i32, _f32, _f32, _vp, _vp,
                                   _vp, _pi32, _vp]),
    "hsr_poly_moments_f64": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _pi32, _vp]),
    "hsr_moments_reduce": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "hsr_poly_solve": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp]),
    "hsr_moments_reduce_solve": (C.c_int, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp]),
    "hsr_poly_solve_host": (C.c_int, [C.POINTER(_f64), _i32, _i32, _i64, C.POINTER(_f64)]),
    "hsr_poly_apply": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp, _i64, _i64, _vp]),
    "hsr_percentile_work_bytes": (C.c_size_t, [_i32]),
    "hsr_percentile_limits": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i32, _f64, _f64, _vp, _vp, _vp]),
    "hsr_percentile_begin": (C.c_int, [_vp, _i32, _vp]),
    "hsr_percentile_hist_region": (C.c_int, [_i32, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hsr_percentile_hist": (C.c_int, [_i32, _vp, _i64, _i64, _vp, _i64, _i32, _vp, _vp]),
    "hsr_percentile_scan": (C.c_int, [_i32, _i32, _f64, _f64, _vp, _vp, _vp]),
    "hsr_tile_encode_u16": (C.c_int, [_vp, _i64, _f32, _i32, _f32, _i32, _vp, _vp]),
    "hsr_tile_decode_u16": (C.c_int, [_vp, _i64, _f32, _i32, _vp, _vp]),
    "hsr_srf_integrate_u16": (C.c_int, [_vp, _i64, _i32, _f32, _i32, _vp, C.POINTER(_i32), C.POINTER(_i32), _i32,
                                        _vp, _i64, _i64, _popt, _vp]),
    "hsr_srf_integrate_moments_u16": (C.c_int, [_vp, _i64, _i32, _f32, _i32, _vp, C.POINTER(_i32), C.POINTER(_i32),
                                                _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _f32, _f32, _i32, _vp,
                                                C.POINTER(_i32), _popt, _vp]),
    "hsr_srf_integrate_moments_apply": (C.c_int, [_vp, _i64, _i32, _vp, _pi32, _pi32, _i32, _vp, _i64, _i64,
                                                  _vp, _i64, _i64, _vp, _f32, _f32, _i32, _vp, _pi32, _popt, _vp, _vp]),
    "hsr_srf_integrate_moments_u16_apply": (C.c_int, [_vp, _i64, _i32, _f32, _i32, _vp, _pi32, _pi32, _i32, _vp, _i64, _i64,
                                                      _vp, _i64, _i64, _vp, _f32, _f32, _i32, _vp, _pi32, _popt, _vp, _vp]),
    "hsr_srf_integrate_fit": (C.c_int, [_vp, _i64, _i32, _vp, _pi32, _pi32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp,
                                        _f32, _f32, _i32, _vp, _pi32, C.POINTER(FusedFit), _popt, _vp]),
    "hsr_srf_integrate_fit_u16": (C.c_int, [_vp, _i64, _i32, _f32, _i32, _vp, _pi32, _pi32, _i32, _vp, _i64, _i64, _vp,
                                            _i64, _i64, _vp, _f32, _f32, _i32, _vp, _pi32, C.POINTER(FusedFit), _popt, _vp]),
    "hsr_batch_partials_bytes": (C.c_size_t, [_i64, _i32, _i32]),
    "hsr_batch_plan": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _popt, _vp, _i64, C.POINTER(BatchInfo)]),
    "hsr_srf_integrate_moments_batched": (C.c_int, [_vp, C.POINTER(BatchInfo), _i32, _f32, _i32, _i32, _vp, _pi32, _pi32,
                                                    _i32, _i32, _i32, _f32, _f32, _i32, _popt, _vp]),
    "hsr_moments_reduce_solve_batched": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    "hsr_poly_apply_batched": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "hsr_interleave_to_bip": (C.c_int, [_vp, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _vp]),
    "hsr_ot_work_bytes": (_i64, [_i64, _i64]),
    "hsr_ot_begin": (C.c_int, [_vp, _i64, _vp, _i64, _f64, _vp, _vp]),
    "hsr_ot_iterate": (C.c_int, [_i64, _i64, _i32, _i32, _f64, _vp, _vp, _vp]),
    "hsr_ot_finish": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "hsr_ot_sinkhorn_barycentric": (C.c_int, [_vp, _i64, _vp, _i64, _f64, _i32, _f64, _vp, _vp, _vp, _vp]),
    "hsr_chol_work_bytes": (C.c_size_t, [_i32]),
    "hsr_chol_solve_f64": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    "hsr_valid_mask": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp]),
    "hsr_polyfeat_count": (C.c_int, [_i32, _i32]),
    "hsr_polyfeat_table": (C.c_int, [_i32, _i32, _vp]),
    "hsr_polyfeat_prepare": (C.c_int, [_i32, _i32]),
    "hsr_polyfeat_expand_f64": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "hsr_gram_work_bytes": (C.c_size_t, [_i32, _i32, _i64]),
    "hsr_gram_f64": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp]),
    "hsr_polyfeat_predict": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i32, _i32,
                                       _vp, _i64, _vp]),
    "hsr_ridge_stats_work_bytes": (C.c_size_t, [_i32]),
    "hsr_ridge_stats": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "hsr_ridge_assemble": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _f64, _vp, _i32, _vp, _i64, _vp, _vp]),
    "hsr_ridge_finish": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hsr_polyfeat_predict_cube": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i32, _i32,
                                            _i32, _f32, _i32, _vp, _i64, _vp]),
    "hsr_block_mean": (C.c_int, [_vp, _i32, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _i64, _vp]),
    "hsr_bilinear_upsample": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp]),
    "hsr_bilinear_upsample_mask_hist": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "hsr_probe_read": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "hsr_step_plan_create": (C.c_int, [C.POINTER(StepDesc), C.POINTER(_vp)]),
    "hsr_step_plan_destroy": (None, [_vp]),
    "hsr_step_plan_slots": (C.c_int, [_vp]),
    "hsr_step_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "hsr_step_run_k1": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "hsr_step_run_reduce": (C.c_int, [_vp, _vp]),
    "hsr_step_run_solve": (C.c_int, [_vp, _vp]),
    "hsr_step_run_apply": (C.c_int, [_vp, _vp, _vp]),
    "hsr_pipeline_create": (C.c_int, [_vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "hsr_pipeline_create_fused": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "hsr_pipeline_destroy": (None, [_vp]),
    "hsr_pipeline_submit": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _pi32, _vp, _vp]),
    "hsr_pipeline_fit_done": (C.c_int, [_vp]),
    "hsr_pipeline_flush": (C.c_int, [_vp, _vp, _vp, _pi32]),
    "hsr_pipeline_count": (_i64, [_vp]),
    "hsr_pipeline_create_exchange": (C.c_int, [C.POINTER(_vp), _vp, C.POINTER(Exchange), C.POINTER(_vp)]),
    "hsr_pipeline_status": (C.c_int, [_vp, _vp, C.POINTER(C.c_uint32)]),
    "hsr_pipeline_create_group": (C.c_int, [C.POINTER(_vp), _i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "hsr_srf_fused_launch_supported": (C.c_int, [_i32, _i32, _i32, _pi32, _pi32, _i64, _i32, _popt]),
    "hsr_comm_available": (C.c_int, []),
    "hsr_comm_version": (C.c_int, []),
    "hsr_comm_unique_id": (C.c_int, [_vp]),
    "hsr_comm_init": (C.c_int, [_i32, _i32, _vp, C.POINTER(_vp)]),
    "hsr_comm_destroy": (C.c_int, [_vp]),
    "hsr_comm_rank": (C.c_int, [_vp]),
    "hsr_comm_ranks": (C.c_int, [_vp]),
    "hsr_allreduce_f64": (C.c_int, [_vp, _vp, _i64, _vp]),
    "hsr_reduce_f64": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "hsr_allreduce_u32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "hsr_bcast": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
}

_lib: Optional[C.CDLL] = None
_lib_path: Optional[str] = None


def library_path() -> str:
    return os.environ.get("HSR_LIBRARY", _DEFAULT_SO)


def load() -> C.CDLL:
    """Load the shared library (once) and type every entry point."""
    global _lib, _lib_path
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise HsrUnavailable(
            f"HIP extension not built: {path} is missing. Build it with "
            f"`make -C hyperspectral_super-resolution_amd/csrc` (or __graft_entry__.build()). "
            f"There is no CPU fallback for the s2_emit hot path.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 not resolvable
        raise HsrUnavailable(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib, _lib_path = lib, path
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != HSR_OK:
        msg = load().hsr_last_error().decode("utf-8", "replace")
        raise HsrError(f"{what or 'libhsr'} failed (code {rc}): {msg}")


def require_gpu():
    """torch is the device-memory / stream plumbing; a real GPU is mandatory for compute."""
    import torch
    if not torch.cuda.is_available():
        raise HsrUnavailable("no ROCm device visible (torch.cuda.is_available() is False); "
                             "the s2_emit hot path runs only on the GPU - there is no CPU fallback.")
    load()
    return torch
