"""The oracle (oracle/oracle_np.py) against the golden fixtures frozen from the reference's own
functions (tests/golden/gen_golden.py).  CPU only.  Bit-exact where the oracle mirrors the
reference's operation order; 1e-12 where it is an algebraic restatement."""
import warnings

import numpy as np
import pytest

from conftest import load_golden, unpack_srf
from oracle import oracle_np as onp

warnings.simplefilter("ignore")


def test_g1_srf_integral_bitexact():
    g = load_golden("g1_srf")
    srf = unpack_srf(g)
    for tag, gm in (("masked", g["good_mask"]), ("nomask", None)):
        out = onp.pseudo_s2_srf_integral(g["R"], g["emit_w"], srf, gm)
        assert list(out.keys()) == list(srf.keys())
        none = {str(n) for n in g[f"{tag}_none"]}
        for k, v in out.items():
            if k in none:
                assert v is None
            else:
                assert v.dtype == np.float64
                np.testing.assert_array_equal(v, g[f"{tag}_{k}"])
    assert {str(n) for n in g["masked_none"]} == {"B10"}
    assert len(g["nomask_none"]) == 0
    np.testing.assert_array_equal(onp.pseudo_s2_rgb(onp.pseudo_s2_srf_integral(
        g["R"], g["emit_w"], srf, g["good_mask"])), g["rgb_masked"])


def test_g1_weight_matrix_identity():
    """out_b == R . Wn[b]: the identity the device kernel is built on."""
    g = load_golden("g1_srf")
    srf = unpack_srf(g)
    Wn, names = onp.srf_weight_matrix(g["emit_w"], srf, g["good_mask"])
    assert names == [k for k in srf if k != "B10"]
    got = g["R"].astype(np.float64) @ Wn.T
    for i, k in enumerate(names):
        np.testing.assert_allclose(got[..., i], g[f"masked_{k}"], rtol=1e-12, atol=1e-15)
    nnz = np.count_nonzero(Wn)
    assert nnz < 0.1 * Wn.size


def test_g2_edge_semantics():
    g = load_golden("g2_srf_edge")
    srf = unpack_srf(g)
    out = onp.pseudo_s2_srf_integral(g["R"], g["emit_w"], srf, g["good_mask"])
    for k, v in out.items():
        if k in {str(n) for n in g["none"]}:
            assert v is None
        else:
            np.testing.assert_array_equal(v, g[f"out_{k}"])   # NaN == NaN, inf signs included
    # Inf inside a supported band stays Inf (not NaN); outside the support it poisons to NaN
    assert np.isposinf(g["out_B1"][0, 2]) and np.isneginf(g["out_B1"][0, 3])
    assert np.isnan(g["out_B1"][0, 4]) and np.isnan(g["out_B12"][0, 2]) and np.isnan(g["out_B1"][0, 5])
    oz = onp.pseudo_s2_srf_integral(g["R"], g["emit_w"], srf, np.zeros(285, bool))
    assert all(v is None for v in oz.values()) and bool(g["allmasked_all_none"])
    R, w = g["R"], g["emit_w"]
    for (bR, bw), msg in zip(((R[0], w), (R, w[:-1]), (R, w.reshape(1, -1))), g["error_messages"]):
        with pytest.raises(ValueError) as ei:
            onp.pseudo_s2_srf_integral(bR, bw, srf)
        assert str(ei.value) == str(msg)
    with pytest.raises(ValueError) as ei:
        onp.pseudo_s2_rgb(out, order=("B4", "B10", "B2"))
    assert str(ei.value) == str(g["rgb_error"])


def test_g3_polyfit():
    g = load_golden("g3_polyfit")
    for N in (200, 5000, 65536):
        if N <= 5000:
            x, y = g[f"x_{N}"], g[f"y_{N}"]
        else:
            rng = np.random.default_rng(100 + N)
            x = rng.random(N)
            y = np.clip(0.9 * x ** 0.8 + 0.03 + 0.02 * rng.standard_normal(N), 0, 1)
            assert x.sum() == g[f"xsum_{N}"] and y.sum() == g[f"ysum_{N}"]
        for deg in (1, 2, 3, 4):
            c = onp.polyfit_channels(x[:, None], y[:, None], deg)[0]
            np.testing.assert_array_equal(c, g[f"coef_{N}_{deg}"])


def test_g4_apply_poly():
    g = load_golden("g4_apply")
    for deg in (1, 2, 3, 4):
        co = g[f"coeffs_{deg}"]
        a = onp.apply_poly_rgb(g["rgb"], co, None)
        b = onp.apply_poly_rgb(g["rgb"], co, g["mask"])
        assert a.dtype == np.float32 and b.dtype == np.float32
        np.testing.assert_array_equal(a, g[f"out_nomask_{deg}"])
        np.testing.assert_array_equal(b, g[f"out_mask_{deg}"])
    np.testing.assert_array_equal(onp.apply_poly_rgb(g["rgb64"], g["coeffs_3"], g["mask"]), g["out_f64in_mask_3"])
    planes = np.ascontiguousarray(np.moveaxis(g["rgb"], -1, 0))
    np.testing.assert_array_equal(np.moveaxis(onp.apply_poly_planes(planes, g["coeffs_3"], g["mask"]), 0, -1),
                                  g["out_mask_3"])


def test_g5_stretch():
    g = load_golden("g5_stretch")
    np.testing.assert_array_equal(onp.apply_shared_percentile_stretch(g["img"], g["mask"]), g["out_f32"])
    np.testing.assert_array_equal(onp.apply_shared_percentile_stretch(g["img"].astype(np.float64), g["mask"]), g["out_f64"])
    np.testing.assert_array_equal(onp.apply_shared_percentile_stretch(g["img"], g["mask"], 5, 95), g["out_5_95"])
    np.testing.assert_array_equal(onp.robust_norm_rgb(g["img"], g["mask"]), g["robust_norm_rgb"])
    np.testing.assert_array_equal(onp.robust_norm(g["img"][..., 0]), g["robust_norm"])
    for c in range(3):
        assert onp.percentile_limits(g["img"][..., c], g["mask"]) == tuple(g["lohi"][c])


def test_g6_per_band_linear():
    g = load_golden("g6_lsq")
    for tag, mask, mv in (("", g["mask"], 0.0), ("_few", g["mask_few"], 0.0), ("_mv", g["mask"], 0.1)):
        corr, params = onp.calibrate_pseudo_to_real_linear(g["pseudo"], g["real"], mask, mv)
        np.testing.assert_array_equal(np.array(params), g[f"params{tag}"])
        np.testing.assert_array_equal(corr, g[f"corrected{tag}"])
    assert (g["params_few"] == np.array([[1.0, 0.0]] * 4)).all()


def test_g7_ridge_pipeline():
    g = load_golden("g7_ridge")
    np.testing.assert_array_equal(onp.poly_feature_exponents(10, 3), g["powers"])
    assert g["powers"].shape == (285, 10)
    np.testing.assert_array_equal(onp.logit(g["Y"].astype(np.float64)), g["Ylogit"])
    m = onp.ridge_poly_fit(g["X"].astype(np.float64), g["Ylogit"], 3, 1.0)
    np.testing.assert_allclose(m["mean"], g["mean"], rtol=1e-14)
    np.testing.assert_allclose(m["scale"], g["scale"], rtol=1e-14)
    pl = onp.ridge_poly_predict(m, g["Xtest"].reshape(-1, 10).astype(np.float64))
    np.testing.assert_allclose(pl, g["pred_logit"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(onp.sigmoid(pl), g["pred"], rtol=1e-7)
    np.testing.assert_allclose(m["intercept"], g["intercept"], rtol=1e-7, atol=1e-8)
    np.testing.assert_array_equal(onp.sigmoid(np.array([-80.0, -50.0, -1.0, 0.0, 2.5, 50.0, 80.0])), g["sigmoid_probe"])
    np.testing.assert_array_equal(onp.logit(np.array([-0.01, 0.0, 1e-4, 0.3, 0.9999, 1.0, 1.2])), g["logit_probe"])


def test_g11_ridge_at_notebook_shapes():
    """The restated pipeline against scikit-learn at the notebook's own shapes (10000 x 10 -> 32 targets) and against the
    NOTEBOOK'S predict_cube_logit on the 10 x 600 x 600 cube (strided sample, one full row, NaN count, per-band checksums);
    plus the 97-target model (the many-target predict kernels' reference)."""
    from conftest import g11_cube
    g = load_golden("g11_ridge_notebook_shapes")
    X = g["X"].astype(np.float32)
    Y = g["Yu16"].astype(np.float32) * np.float32(1e-4)
    m = onp.ridge_poly_fit(X.astype(np.float64), onp.logit(Y.astype(np.float64)), 3, 1.0)
    np.testing.assert_allclose(m["mean"], g["mean"], rtol=1e-13)
    np.testing.assert_allclose(m["scale"], g["scale"], rtol=1e-13)
    np.testing.assert_allclose(m["intercept"], g["intercept"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(onp.ridge_poly_predict(m, X[:512].astype(np.float64)), g["train_pred_logit"], rtol=0, atol=2e-6)
    cube = g11_cube(g)
    assert cube.shape == (10, 600, 600)
    # full-size prediction on a slab of rows (the whole cube would take ~1 min of pure-NumPy feature expansion): the rows
    # the strided sample and the stored full row fall into
    rows = sorted(set(range(0, 600, 7)) | {299})[::6] + [299, 595]
    sub = cube[:, rows, :]
    pred = onp.predict_cube_logit(m, sub, nodata=float(g["nodata"]))
    for k, r in enumerate(rows):
        if r % 7 == 0:
            np.testing.assert_allclose(pred[:, k, ::11], g["pred_sample"][:, r // 7, :], rtol=0, atol=2e-6, equal_nan=True)
        if r == 299:
            np.testing.assert_allclose(pred[:, k, :], g["pred_rows_299"], rtol=0, atol=2e-6, equal_nan=True)
    assert np.isnan(pred[:, rows.index(0), 0]).all()                  # cube_nan[0] = (0, 0, 0)
    assert int(g["pred_nan_count"]) == 32 * 7
    m2 = onp.ridge_poly_fit(X[:2000].astype(np.float64), onp.logit((g["Y2u16"].astype(np.float32) * np.float32(1e-4)).astype(np.float64)), 3, 1.0)
    np.testing.assert_allclose(onp.ridge_poly_predict(m2, g["Xtest2"].astype(np.float64)), g["pred2_logit"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(m2["intercept"], g["intercept2"], rtol=1e-6, atol=1e-8)


def test_g8_histogram_match():
    g = load_golden("g8_histmatch")
    np.testing.assert_array_equal(onp.histogram_match_rgb(g["src"], g["ref"], g["mask"]), g["out"])


def test_g9_fit_fallback_identity():
    g = load_golden("g9_fit_fallback")
    np.testing.assert_array_equal(onp.fit_ot_poly_rgb(g["src"], g["ref"], g["mask"], deg=2), g["coeffs_deg2"])
    np.testing.assert_array_equal(onp.fit_ot_poly_rgb(g["src"], g["ref"], g["mask"], deg=4), g["coeffs_deg4"])
    np.testing.assert_array_equal(onp.fit_ot_poly_rgb(g["src_nan"], g["ref"], g["mask_b"], deg=3), g["coeffs_nan_deg3"])
    assert (g["coeffs_deg2"] == np.array([[0.0, 1.0, 0.0]] * 3)).all()


def test_sinkhorn_invariants_parity_unpinned():
    """POT is absent: the Sinkhorn restatement is checked by OT invariants only (parity unpinned)."""
    rng = np.random.default_rng(0)
    X, Y = rng.random((300, 3)), rng.random((280, 3)) * 0.8 + 0.1
    a, b = np.full(300, 1 / 300), np.full(280, 1 / 280)
    P = onp.sinkhorn_knopp(a, b, onp.sqeuclidean_cost(X, Y), 0.05, 300, 1e-6)
    assert np.linalg.norm(P.sum(0) - b) < 1e-5 and np.linalg.norm(P.sum(1) - a) < 1e-5
    Yb = onp.ot_barycentric_targets(X, Y)
    assert (Yb >= Y.min(0) - 1e-12).all() and (Yb <= Y.max(0) + 1e-12).all()
    co = onp.fit_ot_poly_rgb(np.tile(X.reshape(300, 1, 3), (1, 4, 1)), np.tile(rng.random((300, 1, 3)), (1, 4, 1)),
                             np.ones((300, 4), bool), deg=2, n_samples=400)
    assert co.shape == (3, 3) and np.isfinite(co).all()


def test_g10_tile_u16_quantisation():
    """uint16 tile writer (tiles_helpers/utils.py:362-374): the restatement reproduces the reference's bits,
    including round-half-even ties, the int32 overflow wrap-to-zero and source-nodata handling."""
    g = load_golden("g10_tile_u16")
    t = g["tile"]
    np.testing.assert_array_equal(onp.tile_encode_u16(t), g["u16_plain"])
    np.testing.assert_array_equal(onp.tile_encode_u16(t, src_nodata=-9999.0), g["u16_srcnodata"])
    np.testing.assert_array_equal(onp.tile_encode_u16(t, None, 2000.0, 4095), g["u16_scale2000_nd4095"])
    # decode(encode(x)) is within half a quantisation step wherever x is representable
    x = t[np.isfinite(t) & (t >= 0) & (t < 6.5)]
    d = onp.tile_decode_u16(onp.tile_encode_u16(x))
    assert np.max(np.abs(d - x)) <= 0.5e-4 * (1 + 1e-3)
    assert np.isnan(onp.tile_decode_u16(np.array([65535, 0, 1234], np.uint16))[0])
    assert onp.tile_decode_u16(np.array([1234], np.uint16))[0] == np.float32(1234) * np.float32(1e-4)
