"""CPU-side tests of the product: host logic, the C-ABI surface, loud failure without a GPU."""
import ctypes
import os
import re
import zipfile

import numpy as np
import pytest

from conftest import ROOT, load_golden, unpack_srf
from oracle import oracle_np as onp

import s2_emit
from s2_emit import _engine as eng
from s2_emit import _native as nat


# ---------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hsr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hsr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = nat.load()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hsr.h but not exported"
        assert n in nat.SIGNATURES, f"{n} has no ctypes signature in _native.py"
    assert set(nat.SIGNATURES) == set(names)
    assert lib.hsr_abi_version() == 5


def test_sizing_helpers_and_error_strings():
    lib = nat.load()
    assert [lib.hsr_moment_count(d) for d in (1, 2, 3, 4)] == [5, 8, 11, 14]
    assert lib.hsr_moment_count(0) == -1 and lib.hsr_moment_count(5) == -1
    assert lib.hsr_partial_slots(1, None) == 1 and lib.hsr_partial_slots(64, None) == 1 and lib.hsr_partial_slots(65, None) == 2
    assert lib.hsr_partial_slots(1 << 20, None) == 512 and lib.hsr_partial_slots(1 << 30, None) == 512
    # the slot layout is a function of (npix, options) only: no process-wide tuning state exists any more
    o = nat.SrfOptions(64, 0, 0, 0)
    assert lib.hsr_partial_slots(65, ctypes.byref(o)) == 2 and lib.hsr_partial_slots(1 << 20, ctypes.byref(o)) == 512
    assert lib.hsr_partial_slots(65, ctypes.byref(nat.SrfOptions(32, 0, 0, 0))) == -1        # 32-pixel geometry removed
    o = nat.SrfOptions(0, 8, 0, 0)
    assert lib.hsr_partial_slots(1 << 20, ctypes.byref(o)) == 496 and lib.hsr_partial_slots(1 << 20, None) == 512
    assert lib.hsr_partial_slots(100, ctypes.byref(nat.SrfOptions(48, 0, 0, 0))) == -1 and b"tile_pixels" in lib.hsr_last_error()
    assert lib.hsr_partial_slots(100, ctypes.byref(nat.SrfOptions(0, 200, 0, 0))) == -1
    assert not hasattr(lib, "hsr_set_srf_tile") and not hasattr(lib, "hsr_set_srf_reserved_cus")
    assert lib.hsr_partials_bytes(12, 3) == 12 * 11 * 4096 * 8
    assert lib.hsr_percentile_work_bytes(3) > 3 * (2048 + 4 * 2048 + 4 * 1024) * 4
    # argument validation happens before any device work -> testable without a GPU
    k = (ctypes.c_int32 * 1)(0)
    rc = lib.hsr_srf_integrate(None, 10, 285, None, k, k, 1, None, 10, 1, None, None)
    assert rc == 1 and b"NULL" in lib.hsr_last_error()
    rc = lib.hsr_srf_integrate(ctypes.c_void_p(16), 10, 9999, ctypes.c_void_p(16), k, k, 1, ctypes.c_void_p(16), 10, 1, None, None)
    assert rc == 2 and b"B=9999" in lib.hsr_last_error()
    rc = lib.hsr_poly_solve(None, 3, 2, 50, None, None)
    assert rc == 1
    # strides must describe band-major planes or pixel-major rows
    rc = lib.hsr_srf_integrate(ctypes.c_void_p(16), 10, 285, ctypes.c_void_p(16), k, k, 1, ctypes.c_void_p(16), 3, 7, None, None)
    assert rc == 1 and b"neither band-major nor pixel-major" in lib.hsr_last_error()
    bad = nat.SrfOptions(0, 0, 0, 7)
    rc = lib.hsr_srf_integrate(ctypes.c_void_p(16), 10, 285, ctypes.c_void_p(16), k, k, 1, ctypes.c_void_p(16), 10, 1, ctypes.byref(bad), None)
    assert rc == 1 and b"unknown bits" in lib.hsr_last_error()
    # the rows added later in the round validate the same way
    P = ctypes.c_void_p(256)
    assert lib.hsr_tile_decode_u16(P, 10, 1e-4, 70000, P, None) == 1 and b"not a uint16 value" in lib.hsr_last_error()
    assert lib.hsr_tile_decode_u16(None, 0, 1e-4, 65535, None, None) == 0            # empty input: nothing to do
    assert lib.hsr_tile_encode_u16(P, 10, 1e4, 0, 0.0, 0, P, None) == 1 and b"nodata_u16" in lib.hsr_last_error()
    assert lib.hsr_srf_integrate_u16(P, 10, 285, 1e-4, 1 << 20, P, k, k, 1, P, 10, 1, None, None) == 1
    assert lib.hsr_percentile_hist(4, P, 10, 1, None, 10, 1, P, None) == 1
    off, cnt = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.hsr_percentile_hist_region(1, 3, ctypes.byref(off), ctypes.byref(cnt)) == 0 and cnt.value == 3 * (2048 + 4)
    assert lib.hsr_percentile_hist_region(3, 3, ctypes.byref(off), ctypes.byref(cnt)) == 0 and cnt.value == 3 * 4 * 1024
    assert off.value + cnt.value * 4 <= lib.hsr_percentile_work_bytes(3)
    assert lib.hsr_ot_work_bytes(5000, 5000) > 5000 * 5000 * 8 and lib.hsr_ot_work_bytes(0, 5) == 0
    assert lib.hsr_ot_sinkhorn_barycentric(P, 10, P, 10, -1.0, 5, 1e-6, P, P, None, None) == 1 and b"reg must be > 0" in lib.hsr_last_error()
    assert lib.hsr_ot_iterate(10, 10, 0, 5, 1e-6, ctypes.c_void_p(8), None, None) == 1 and b"256-byte aligned" in lib.hsr_last_error()
    assert lib.hsr_gram_f64(P, 30, 32, P, 48, 48, 10, P, P, 48, None) == 1
    # one 2 KB tile of partial sums per (chunk, output tile): 29 chunks of <= 1024 rows for the register-operand kernel
    # (the LDS-panel kernel uses fewer, longer chunks: one workgroup per CU)
    assert lib.hsr_gram_work_bytes(288, 576, 29127) >= 29 * 18 * 36 * 256 * 8
    assert lib.hsr_gram_work_bytes(288, 320, 29127) >= 29 * 18 * 20 * 256 * 8 and lib.hsr_gram_work_bytes(8, 16, 100) == 0
    assert lib.hsr_probe_read(P, 1 << 20, 4, P, None) == 1 and b"mode" in lib.hsr_last_error()
    # entry points added in round 2 validate the same way (no launch without a GPU: every call below fails its checks first)
    assert lib.hsr_srf_integrate_fit(P, 10, 285, P, k, k, 1, P, 1, 16, P, 1, 16, None, 0.0, 0.0, 3, P, None, None, None, None) == 1
    assert b"NULL hsr_fused_fit" in lib.hsr_last_error()
    bad_fit = nat.FusedFit(0, 0, 0, 0, 50)
    assert lib.hsr_srf_integrate_fit(P, 10, 285, P, k, k, 1, P, 1, 16, P, 1, 16, None, 0.0, 0.0, 3, P, None, ctypes.byref(bad_fit), None, None) == 1
    assert b"NULL pointer in hsr_fused_fit" in lib.hsr_last_error()
    assert lib.hsr_ridge_stats_work_bytes(10) == 64 * 10 * 2 * 8 and lib.hsr_ridge_stats_work_bytes(17) == 0
    assert lib.hsr_ridge_stats(P, 10, 1, 0, 10, P, P, P, P, None) == 1 and b"hsr_ridge_stats" in lib.hsr_last_error()
    assert lib.hsr_ridge_assemble(P, 300, 288, 285, 32, 1.0, P, 256, P, 32, P, None) == 1 and b"bad shape" in lib.hsr_last_error()   # npad < nf
    assert lib.hsr_ridge_finish(P, 288, 285, 32, P, 16, P, P, 10, 286, P, P, P, P, P, None) == 1                                  # ldw < T
    assert lib.hsr_polyfeat_predict_cube(None, 1, 1, P, P, 10, 10, 3, P, 32, P, 32, 1, 1, -9999.0, 1, P, 10, None) == 1 and b"NULL" in lib.hsr_last_error()
    assert lib.hsr_interleave_to_bip(P, 0, 3, 4, 4, 4, P, 0, None) == 1


def test_batch_plan_host_side():
    """hsr_batch_plan is pure host code: per-tile slot ranges == hsr_partial_slots of each tile, one unit per
    (tile, slot), longest units first, part_dev pointers inside the workspace."""
    lib = nat.load()
    npix = [10000, 64, 65, 1 << 20, 10000, 1]
    T = len(npix)
    tiles = (nat.BatchTile * T)()
    for i, n in enumerate(npix):
        tiles[i].cube_dev = 0x10000000 + i * 0x1000000
        tiles[i].real_dev = 0x20000000 + i * 0x1000000
        tiles[i].pseudo_dev = 0x30000000 + i * 0x1000000
        tiles[i].matched_dev = 0x40000000 + i * 0x1000000
        tiles[i].npix = n
    info = nat.BatchInfo()
    nb, deg, M = 12, 3, 11
    assert lib.hsr_batch_plan(tiles, T, nb, deg, None, None, None, 0, ctypes.byref(info)) == 0
    slots = [lib.hsr_partial_slots(n, None) for n in npix]
    assert [tiles[i].slots for i in range(T)] == slots == [64, 1, 2, 512, 64, 1]     # 157 groups -> 64 slots (hsr.h)
    assert [tiles[i].ngroups for i in range(T)] == [157, 1, 2, 16384, 157, 1]
    assert [tiles[i].slot0 for i in range(T)] == list(np.cumsum([0] + slots[:-1]))
    assert info.nunits == sum(slots) and info.total_pixels == sum(npix) and info.max_npix == 1 << 20
    assert info.ntiles == T and info.aligned16 == 1
    assert lib.hsr_batch_partials_bytes(info.nunits, nb, deg) == info.nunits * nb * M * 8
    n = int(info.nunits)
    buf = (ctypes.c_uint8 * (64 * n))()
    base = 0x50000000
    assert lib.hsr_batch_plan(tiles, T, nb, deg, ctypes.c_void_p(base), None, buf, n - 1, ctypes.byref(info)) == 1   # too small
    assert lib.hsr_batch_plan(tiles, T, nb, deg, ctypes.c_void_p(base), None, buf, n, ctypes.byref(info)) == 0
    rec = np.frombuffer(buf, dtype=np.dtype([("cube", "<u8"), ("real", "<u8"), ("mask", "<u8"), ("pseudo", "<u8"), ("part", "<u8"),
                                              ("npix", "<i8"), ("slots", "<i4"), ("slot", "<i4"), ("ngroups", "<i4"), ("pad", "<i4")]))
    assert len(rec) == n
    seen = set()
    lens = []
    for r in rec:
        t = [i for i in range(T) if tiles[i].cube_dev == int(r["cube"])][0]
        assert (int(r["npix"]), int(r["slots"]), int(r["ngroups"])) == (npix[t], slots[t], tiles[t].ngroups)
        assert 0 <= r["slot"] < slots[t] and (t, int(r["slot"])) not in seen
        seen.add((t, int(r["slot"])))
        assert int(r["part"]) == base + 8 * (int(tiles[t].slot0) + int(r["slot"])) * nb * M      # slot-major partials
        lens.append((tiles[t].ngroups - int(r["slot"]) + slots[t] - 1) // slots[t])
    assert len(seen) == n and lens == sorted(lens, reverse=True) and lens[0] == 32 and lens[-1] == 1
    # options: reserved CUs shrink the slot cap of the big tile only
    o = nat.SrfOptions(0, 8, 0, 0)
    assert lib.hsr_batch_plan(tiles, T, nb, deg, None, ctypes.byref(o), None, 0, ctypes.byref(info)) == 0
    assert tiles[3].slots == 496 and tiles[0].slots == 64
    # validation
    tiles[1].npix = 0
    assert lib.hsr_batch_plan(tiles, T, nb, deg, None, None, None, 0, ctypes.byref(info)) == 1 and b"npix" in lib.hsr_last_error()
    tiles[1].npix = 64
    tiles[2].pseudo_dev = 0x30000004
    assert lib.hsr_batch_plan(tiles, T, nb, deg, None, None, None, 0, ctypes.byref(info)) == 1 and b"16-byte" in lib.hsr_last_error()


def test_fused_launch_predicate_and_exchange_validation_host_side():
    """hsr_srf_fused_launch_supported is pure host code: the one predicate behind hsr_pipeline_create_fused / _exchange (ADVICE r3:
    Python's looser test let a launch fail with tiles in flight).  And the argument checks of the ABI 5 entry points that need no GPU."""
    lib = nat.load()
    from oracle import oracle_np as onp
    from s2_emit import _engine as eng
    w, good = onp.synthetic_wavelengths()
    t12 = eng.build_srf_table(w, onp.synthetic_srf(), good)
    k0 = (ctypes.c_int32 * t12.nb)(*[int(v) for v in t12.k0])
    kl = (ctypes.c_int32 * t12.nb)(*[int(v) for v in t12.klen])
    ok = lambda *a: lib.hsr_srf_fused_launch_supported(*a)
    assert ok(0, 285, t12.nb, k0, kl, 12, 3, None) == 0 and ok(2, 285, t12.nb, k0, kl, 12, 3, None) == 0
    assert ok(0, 285, t12.nb, k0, kl, 12, 0, None) == 2 and ok(0, 285, t12.nb, k0, kl, 10, 3, None) == 2      # degree 0; rows of 10 floats
    assert ok(2, 285, t12.nb, k0, kl, 12, 3, ctypes.byref(nat.SrfOptions(0, 0, 1, 0))) == 2                    # single-buffer uint16 kernel
    assert ok(1, 285, t12.nb, k0, kl, 12, 3, None) == 1
    # uint16 tiles with a wide spectrum, 13 bands in rows of 16 and long supports: the two ring buffers no longer fit next to them
    B = 300
    k0w = (ctypes.c_int32 * 13)(*[int(i * 10) for i in range(13)])
    klw = (ctypes.c_int32 * 13)(*([70] * 13))                   # 13 x 80 padded taps = 1040 > the 1024 floats reserved -> no LDS weights
    assert ok(2, B, 13, k0w, klw, 16, 3, None) == 2 and b"weight taps" in lib.hsr_last_error()
    klm = (ctypes.c_int32 * 13)(*([40] * 13))                   # 13 x 48 = 624 taps: fit, but 2 x 38 400 B + rows of 16 + taps > 80 KB
    assert ok(2, B, 13, k0w, klm, 16, 3, None) == 2 and b"80 KB" in lib.hsr_last_error()
    assert ok(0, B, 13, k0w, klm, 16, 3, None) == 0             # the float32 kernel has no such limit at B = 300
    assert ok(2, 40, 3, k0w, (ctypes.c_int32 * 3)(8, 8, 8), 4, 2, None) == 2                                   # B < 48: group buffers too small for the ring
    # exchange pipeline / communicator: argument errors come back as codes, nothing is dereferenced
    import torch  # noqa: F401  (first: hsr_comm_* binds the librccl.so.1 that is already mapped - PyTorch's own copy; loading ROCm's
    #                            copy and then PyTorch's puts two rocm_smi instances into the process, a double free at exit)
    assert lib.hsr_comm_available() in (0, 1)
    h = ctypes.c_void_p()
    assert lib.hsr_comm_init(3, 2, (ctypes.c_ubyte * 128)(), ctypes.byref(h)) == 1 and lib.hsr_comm_init(0, 1, None, ctypes.byref(h)) == 1
    assert lib.hsr_comm_unique_id(None) == 1 and lib.hsr_comm_destroy(None) == 0
    assert lib.hsr_comm_ranks(None) == -1 and lib.hsr_comm_rank(None) == -1
    assert lib.hsr_allreduce_f64(None, None, 0, None) == 1 and lib.hsr_bcast(None, None, 0, 0, None) == 1
    x = nat.Exchange()
    assert lib.hsr_pipeline_create_exchange(None, None, ctypes.byref(x), ctypes.byref(h)) == 1
    four = (ctypes.c_void_p * 4)()
    assert lib.hsr_pipeline_create_exchange(four, ctypes.c_void_p(1), ctypes.byref(x), ctypes.byref(h)) == 1 and b"exactly one" in lib.hsr_last_error()
    code = ctypes.c_uint32(7)
    assert lib.hsr_pipeline_status(None, None, ctypes.byref(code)) == 1


def test_polyfeat_table_matches_sklearn_order():
    lib = nat.load()
    assert lib.hsr_polyfeat_count(10, 3) == 285 and lib.hsr_polyfeat_count(10, 2) == 65
    assert lib.hsr_polyfeat_count(17, 3) == -1 and lib.hsr_polyfeat_count(10, 4) == -1
    buf = (ctypes.c_uint8 * (285 * 3))()
    assert lib.hsr_polyfeat_table(10, 3, buf) == 0
    idx = np.frombuffer(buf, dtype=np.uint8).reshape(285, 3)
    expo = np.zeros((285, 10), int)
    for f in range(285):
        for k in idx[f]:
            if k < 10:
                expo[f, k] += 1
    np.testing.assert_array_equal(expo, load_golden("g7_ridge")["powers"])     # sklearn's powers_


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    srf = onp.synthetic_srf()
    w, good = onp.synthetic_wavelengths()
    with pytest.raises(nat.HsrUnavailable):
        s2_emit.pseudo_s2_srf_integral(np.zeros((2, 2, 285), np.float32), w, srf, good)
    with pytest.raises(nat.HsrUnavailable):
        s2_emit.apply_poly_rgb(np.zeros((2, 2, 3), np.float32), np.zeros((3, 3)))
    with pytest.raises(nat.HsrUnavailable):
        s2_emit.apply_shared_percentile_stretch(np.zeros((2, 2, 3), np.float32), np.ones((2, 2), bool))


def test_missing_library_message(monkeypatch):
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setenv("HSR_LIBRARY", "/nonexistent/libhsr.so")
    with pytest.raises(nat.HsrUnavailable) as ei:
        nat.load()
    assert "no CPU fallback" in str(ei.value)
    monkeypatch.delenv("HSR_LIBRARY")
    nat.load()


# ---------------------------------------------------------------------------------------------
# host solve: np.polyfit from moments
# ---------------------------------------------------------------------------------------------
def _moments(x, y, deg):
    x = x.astype(np.float64)
    S = [np.sum(x ** k) for k in range(2 * deg + 1)]
    T = [np.sum(x ** j * y) for j in range(deg + 1)]
    return np.array(S + T)


@pytest.mark.parametrize("deg", [1, 2, 3, 4])
def test_solve_host_matches_polyfit_golden(deg):
    g = load_golden("g3_polyfit")
    for N in (200, 5000):
        x, y = g[f"x_{N}"], g[f"y_{N}"]
        c = eng.poly_solve_host(_moments(x, y, deg)[None], deg, 0)[0]
        np.testing.assert_allclose(c, g[f"coef_{N}_{deg}"], rtol=2e-8, atol=1e-10)
    xf, yf = g["xf32"].astype(np.float64), g["yf32"].astype(np.float64)
    c = eng.poly_solve_host(_moments(xf, yf, deg)[None], deg, 50)[0]
    np.testing.assert_allclose(c, g[f"coef_f32_{deg}"], rtol=2e-8, atol=1e-10)


def test_solve_host_fallbacks_and_rank_loss():
    m = _moments(np.linspace(0, 1, 40), np.linspace(0, 1, 40), 2)
    np.testing.assert_array_equal(eng.poly_solve_host(m[None], 2, 50)[0], [0.0, 1.0, 0.0])
    np.testing.assert_array_equal(eng.poly_solve_host(np.zeros((1, 14)), 4, 0)[0], [0, 0, 0, 1, 0])
    # constant x: rank-1 Vandermonde -> minimum-norm solution like np.polyfit (no NaN, same values)
    x = np.full(300, 0.5)
    y = np.full(300, 0.25)
    c = eng.poly_solve_host(_moments(x, y, 2)[None], 2, 50)[0]
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = np.polyfit(x, y, 2)
    np.testing.assert_allclose(c, ref, rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# SRF weight table
# ---------------------------------------------------------------------------------------------
def test_srf_table_matches_oracle_weights():
    g = load_golden("g1_srf")
    srf = unpack_srf(g)
    t = eng.build_srf_table(g["emit_w"], srf, g["good_mask"])
    Wn, names = onp.srf_weight_matrix(g["emit_w"], srf, g["good_mask"])
    assert t.names == list(srf) and t.supported == names and "B10" not in t.supported
    np.testing.assert_array_equal(t.weights, Wn)
    for b in range(t.nb):
        nz = np.nonzero(t.weights[b])[0]
        assert t.k0[b] == nz[0] and t.k0[b] + t.klen[b] == nz[-1] + 1
    assert t.klen.sum() < 400            # the sparse structure the kernel exploits
    # weight-matrix product reproduces the reference planes (float64)
    got = g["R"].astype(np.float64) @ t.weights.T
    for i, k in enumerate(t.supported):
        np.testing.assert_allclose(got[..., i], g[f"masked_{k}"], rtol=1e-12, atol=1e-15)
    t2 = eng.build_srf_table(g["emit_w"], srf, None)
    assert "B10" in t2.supported and t2.nb == 13


def test_shape_errors_match_reference_text():
    g = load_golden("g2_srf_edge")
    srf = unpack_srf(g)
    R, w = g["R"], g["emit_w"]
    for (bR, bw), msg in zip(((R[0], w), (R, w[:-1]), (R, w.reshape(1, -1))), g["error_messages"]):
        with pytest.raises(ValueError) as ei:
            s2_emit.pseudo_s2_srf_integral(bR, bw, srf)
        assert str(ei.value) == str(msg)
    with pytest.raises(ValueError) as ei:
        s2_emit.pseudo_s2_rgb({"B4": np.zeros((2, 2)), "B3": np.zeros((2, 2)), "B2": None})
    assert str(ei.value) == "Band B2 is None/missing in pseudo_s2."
    rgb = s2_emit.pseudo_s2_rgb({"B4": np.ones((2, 2)), "B3": np.zeros((2, 2)), "B2": np.zeros((2, 2))})
    assert rgb.shape == (2, 2, 3) and rgb[0, 0, 0] == 1
    # every band unsupported -> dict of None, no GPU needed
    out = s2_emit.pseudo_s2_srf_integral(R, w, srf, np.zeros(285, bool))
    assert list(out) == list(srf) and all(v is None for v in out.values())


# ---------------------------------------------------------------------------------------------
# SRF table loader (xlsx / csv / npz), API surface
# ---------------------------------------------------------------------------------------------
def _write_xlsx(path, sheets):
    """Minimal xlsx writer for the test: sheets = {name: (header list, rows list)}."""
    def col(i):
        s = ""
        i += 1
        while i:
            i, r = divmod(i - 1, 26)
            s = chr(65 + r) + s
        return s
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("[Content_Types].xml", "<Types xmlns='http://schemas.openxmlformats.org/package/2006/content-types'/>")
        wb = ["<workbook xmlns='http://schemas.openxmlformats.org/spreadsheetml/2006/main' "
              "xmlns:r='http://schemas.openxmlformats.org/officeDocument/2006/relationships'><sheets>"]
        rels = ["<Relationships xmlns='http://schemas.openxmlformats.org/package/2006/relationships'>"]
        for i, (name, (header, rows)) in enumerate(sheets.items(), 1):
            wb.append(f"<sheet name='{name}' sheetId='{i}' r:id='rId{i}'/>")
            rels.append(f"<Relationship Id='rId{i}' Type='x' Target='worksheets/sheet{i}.xml'/>")
            xml = ["<worksheet xmlns='http://schemas.openxmlformats.org/spreadsheetml/2006/main'><sheetData>"]
            xml.append("<row r='1'>" + "".join(
                f"<c r='{col(j)}1' t='inlineStr'><is><t>{h}</t></is></c>" for j, h in enumerate(header)) + "</row>")
            for r, row in enumerate(rows, 2):
                cells = []
                for j, v in enumerate(row):
                    if v is None:
                        continue
                    if isinstance(v, str):
                        cells.append(f"<c r='{col(j)}{r}' t='inlineStr'><is><t>{v}</t></is></c>")
                    else:
                        cells.append(f"<c r='{col(j)}{r}'><v>{v!r}</v></c>")
                xml.append(f"<row r='{r}'>" + "".join(cells) + "</row>")
            xml.append("</sheetData></worksheet>")
            z.writestr(f"xl/worksheets/sheet{i}.xml", "".join(xml))
        wb.append("</sheets></workbook>")
        rels.append("</Relationships>")
        z.writestr("xl/workbook.xml", "".join(wb))
        z.writestr("xl/_rels/workbook.xml.rels", "".join(rels))


def test_load_s2_srf_from_xlsx_local_file(tmp_path):
    lam = np.arange(400.0, 460.0)
    header = ["SR_WL"] + [f"S2A_SR_AV_{b}" for b in s2_emit.S2_BANDS_13]
    rows = []
    for i, l in enumerate(lam):
        row = [float(l)]
        for j in range(13):
            v = float(np.exp(-0.5 * ((l - 410 - 3 * j) / 6.0) ** 2))
            row.append(v if v > 1e-3 else 0.0)
        rows.append(row)
    rows[5][1] = None            # blank cell -> NaN -> dropped
    rows[6][2] = "n/a"           # text -> coerced to NaN -> dropped
    p = tmp_path / "srf.xlsx"
    _write_xlsx(p, {"Readme": (["x"], [[1.0]]), "Spectral Responses (S2A)": (header, rows),
                    "Spectral Responses (S2B)": (header, rows)})
    srf = s2_emit.load_s2_srf_from_xlsx(str(p))
    assert list(srf) == s2_emit.S2_BANDS_13
    lam1, r1 = srf["B1"]
    assert lam1.dtype == np.float64 and (r1 > 0).all() and 405.0 not in lam1
    assert 406.0 not in srf["B2"][0]
    assert len(srf["B12"][0]) < len(lam)           # zero responses filtered (resp > 0)
    sub = s2_emit.load_s2_srf_from_xlsx(str(p), bands=["B2", "B3"])
    assert list(sub) == ["B2", "B3"]
    with pytest.raises(ValueError) as ei:
        s2_emit.load_s2_srf_from_xlsx(str(p), platform="S2C")
    assert "No sheet containing 'Spectral Responses' and 'S2C' found." in str(ei.value)
    with pytest.raises(KeyError) as ei:
        s2_emit.load_s2_srf_from_xlsx(str(p), bands=["B99"])
    assert "Column 'S2A_SR_AV_B99' not found in sheet 'Spectral Responses (S2A)'." in str(ei.value)
    # csv / npz / dict sources give the same table
    import csv
    with open(tmp_path / "srf.csv", "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(header)
        for row in rows:
            wr.writerow(["" if v is None else v for v in row])
    c = s2_emit.load_s2_srf_from_xlsx(str(tmp_path / "srf.csv"))
    for b in srf:
        np.testing.assert_array_equal(c[b][0], srf[b][0])
        np.testing.assert_allclose(c[b][1], srf[b][1], rtol=1e-15)


def test_api_surface_matches_reference_all():
    ref_all = ["load_s2_srf_from_xlsx", "load_emit_envi_rfl", "load_emit_wavelengths_from_nc",
               "pseudo_s2_srf_integral", "pseudo_s2_rgb", "show_side_by_side", "resize_s2_rgb_to",
               "robust_norm", "robust_norm_rgb", "apply_shared_percentile_stretch",
               "histogram_match_rgb", "ot_match_rgb_sinkhorn_pot", "load_s2_rgb_u8"]
    assert s2_emit.__all__[:13] == ref_all
    for n in s2_emit.__all__:
        assert callable(getattr(s2_emit, n))
    import inspect
    sig = inspect.signature(s2_emit.fit_ot_poly_rgb)
    assert [(k, v.default) for k, v in sig.parameters.items()][3:] == [
        ("deg", 2), ("n_samples", 5000), ("reg", 0.05), ("numItermax", 300), ("stopThr", 1e-6), ("seed", 0)]
    assert list(inspect.signature(s2_emit.apply_poly_rgb).parameters) == ["rgb", "coeffs", "mask"]
    assert list(inspect.signature(s2_emit.pseudo_s2_srf_integral).parameters)[:4] == ["R", "emit_w", "srf_dict", "good_mask"]
    assert list(inspect.signature(s2_emit.apply_shared_percentile_stretch).parameters) == ["img", "mask", "pmin", "pmax"]


def test_host_numpy_api_functions_against_golden():
    g5 = load_golden("g5_stretch")
    np.testing.assert_array_equal(s2_emit.robust_norm_rgb(g5["img"], g5["mask"]), g5["robust_norm_rgb"])
    np.testing.assert_array_equal(s2_emit.robust_norm(g5["img"][..., 0]), g5["robust_norm"])
    g8 = load_golden("g8_histmatch")
    np.testing.assert_array_equal(s2_emit.histogram_match_rgb(g8["src"], g8["ref"], g8["mask"]), g8["out"])
    g9 = load_golden("g9_fit_fallback")     # identity fallback needs no GPU
    np.testing.assert_array_equal(s2_emit.fit_ot_poly_rgb(g9["src"], g9["ref"], g9["mask"], deg=2), g9["coeffs_deg2"])
    np.testing.assert_array_equal(s2_emit.fit_ot_poly_rgb(g9["src_nan"], g9["ref"], g9["mask_b"], deg=3), g9["coeffs_nan_deg3"])


def test_envi_loader_layouts(tmp_path):
    rng = np.random.default_rng(0)
    cube = rng.random((5, 7, 9)).astype(np.float32)          # (H, W, B)
    for inter, arr in (("bip", cube), ("bil", cube.transpose(0, 2, 1)), ("bsq", cube.transpose(2, 0, 1))):
        np.ascontiguousarray(arr).tofile(tmp_path / f"c_{inter}.bin")
        (tmp_path / f"c_{inter}.hdr").write_text(
            f"ENVI\ndescription = {{\n test }}\nsamples = 7\nlines = 5\nbands = 9\nheader offset = 0\n"
            f"data type = 4\ninterleave = {inter}\nbyte order = 0\n")
        R = s2_emit.load_emit_envi_rfl(str(tmp_path / f"c_{inter}.hdr"), str(tmp_path / f"c_{inter}.bin"))
        assert R.dtype == np.float32 and R.flags.c_contiguous
        np.testing.assert_array_equal(R, cube)
