"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN FUNCTIONS (build container only).

    python tests/golden/gen_golden.py

Every expected output stored here comes from a reference function loaded from /root/reference by
``oracle/ref_loader.py`` (or, for the notebook-only ridge pipeline, from scikit-learn exactly as
the notebook composes it).  Inputs are stored next to the outputs so the fixtures are
self-contained data; no reference source text is stored.  NumPy 2.2.6 / scikit-learn 1.7.2.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle_np as onp      # noqa: E402  (inputs only: synthetic generators)
from oracle import ref_loader            # noqa: E402

warnings.simplefilter("ignore")


def pack_srf(srf):
    names = list(srf.keys())
    lens = np.array([len(srf[n][0]) for n in names], dtype=np.int64)
    lam = np.concatenate([srf[n][0] for n in names])
    rsp = np.concatenate([srf[n][1] for n in names])
    return dict(srf_names=np.array(names), srf_lens=lens, srf_lam=lam, srf_rsp=rsp)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def main():
    assert ref_loader.available(), "reference tree not present"
    synth = ref_loader.load_synth()
    color = ref_loader.load_color()
    poly = ref_loader.load_poly_functions()
    nbf = ref_loader.load_pairs_notebook_functions()
    smf = ref_loader.load_spectral_matching_functions()

    srf = onp.synthetic_srf()
    w, good = onp.synthetic_wavelengths()

    # ---- G1: SRF integral on a 10x10x285 cube, with and without good_mask -------------------
    R = onp.synthetic_cube(10, 10, seed=0)
    out_m = synth.pseudo_s2_srf_integral(R, w, srf, good_mask=good)
    out_n = synth.pseudo_s2_srf_integral(R, w, srf, good_mask=None)
    g1 = dict(R=R, emit_w=w, good_mask=good, **pack_srf(srf))
    for tag, out in (("masked", out_m), ("nomask", out_n)):
        g1[f"{tag}_none"] = np.array([k for k, v in out.items() if v is None])
        for k, v in out.items():
            if v is not None:
                g1[f"{tag}_{k}"] = v
    g1["rgb_masked"] = synth.pseudo_s2_rgb(out_m)
    save("g1_srf", **g1)

    # ---- G2: edge semantics (NaN / +-Inf / nodata rows, zero-weight bands) ---------------------
    Re = onp.synthetic_cube(4, 6, seed=3)
    Re[0, 0, 40] = np.nan          # inside several SRF supports
    Re[0, 1, 284] = np.nan         # last band: zero weight everywhere with good_mask
    Re[0, 2, 10] = np.inf          # inside B1/B2 support
    Re[0, 3, 10] = -np.inf
    Re[0, 4, 10] = np.inf
    Re[0, 4, 11] = -np.inf
    Re[0, 5, 130] = np.inf         # water band: masked => zero weight in every band
    Re[1, 0, :] = -9999.0          # nodata spectrum
    Re[1, 1, :] = -0.01            # masked-band fill value everywhere
    Re[1, 2, :] = 0.0
    Re[1, 3, 0] = np.inf           # first band
    Re[1, 4, 150:160] = np.nan
    Re[1, 5, 200] = 3.0e38         # huge but finite
    oe = synth.pseudo_s2_srf_integral(Re, w, srf, good_mask=good)
    g2 = dict(R=Re, emit_w=w, good_mask=good, **pack_srf(srf))
    g2["none"] = np.array([k for k, v in oe.items() if v is None])
    for k, v in oe.items():
        if v is not None:
            g2[f"out_{k}"] = v
    # all-masked (every band unsupported) and exception texts
    allbad = np.zeros_like(good)
    oz = synth.pseudo_s2_srf_integral(Re, w, srf, good_mask=allbad)
    g2["allmasked_all_none"] = np.array(all(v is None for v in oz.values()))
    msgs = []
    for bad_R, bad_w in ((Re[0], w), (Re, w[:-1]), (Re, w.reshape(1, -1))):
        try:
            synth.pseudo_s2_srf_integral(bad_R, bad_w, srf)
            msgs.append("")
        except ValueError as e:
            msgs.append(str(e))
    g2["error_messages"] = np.array(msgs)
    try:
        synth.pseudo_s2_rgb(oe, order=("B4", "B10", "B2"))
        g2["rgb_error"] = np.array("")
    except ValueError as e:
        g2["rgb_error"] = np.array(str(e))
    save("g2_srf_edge", **g2)

    # ---- G3: np.polyfit on (x, ybar) columns ---------------------------------------------------
    g3 = {}
    for N in (200, 5000, 65536):
        rng = np.random.default_rng(100 + N)
        x = rng.random(N)
        y = np.clip(0.9 * x ** 0.8 + 0.03 + 0.02 * rng.standard_normal(N), 0, 1)
        if N <= 5000:
            g3[f"x_{N}"], g3[f"y_{N}"] = x, y
        else:   # regenerated from the seed by the test; checksums guard RNG drift
            g3[f"xsum_{N}"], g3[f"ysum_{N}"] = np.array(x.sum()), np.array(y.sum())
        for deg in (1, 2, 3, 4):
            g3[f"coef_{N}_{deg}"] = np.polyfit(x, y, deg)
    # float32-valued samples (what the device sees) for the all-pixel flavour
    rng = np.random.default_rng(7)
    xf = rng.random(4096).astype(np.float32)
    yf = np.clip(1.1 * xf ** 1.2 + 0.01 + 0.01 * rng.standard_normal(4096), 0, 1).astype(np.float32)
    g3["xf32"], g3["yf32"] = xf, yf
    for deg in (1, 2, 3, 4):
        g3[f"coef_f32_{deg}"] = np.polyfit(xf.astype(np.float64), yf.astype(np.float64), deg)
    save("g3_polyfit", **g3)

    # ---- G4: apply_poly_rgb ---------------------------------------------------------------------
    rng = np.random.default_rng(4)
    rgb = (rng.random((12, 9, 3)) * 1.4 - 0.2).astype(np.float32)
    rgb[0, 0, 0] = np.nan
    rgb[0, 1, 1] = np.inf
    rgb[0, 2, 2] = -np.inf
    mask = rng.random((12, 9)) > 0.3
    mask[0, :3] = True
    g4 = dict(rgb=rgb, mask=mask)
    for deg in (1, 2, 3, 4):
        co = rng.standard_normal((3, deg + 1)) * 0.5
        co[:, -2] += 1.0
        g4[f"coeffs_{deg}"] = co
        g4[f"out_nomask_{deg}"] = poly["apply_poly_rgb"](rgb, co, None)
        g4[f"out_mask_{deg}"] = poly["apply_poly_rgb"](rgb, co, mask)
    g4["rgb64"] = rgb.astype(np.float64)
    g4["out_f64in_mask_3"] = poly["apply_poly_rgb"](rgb.astype(np.float64), g4["coeffs_3"], mask)
    save("g4_apply", **g4)

    # ---- G5: percentile stretch -------------------------------------------------------------------
    rng = np.random.default_rng(5)
    img = (rng.random((20, 17, 3)) * 0.5).astype(np.float32)
    img[..., 2] = 0.25                                   # hi == lo plane
    m5 = rng.random((20, 17)) > 0.2
    img[~m5, 0] = 5.0                                    # outliers outside the mask
    g5 = dict(img=img, mask=m5)
    g5["out_f32"] = color.apply_shared_percentile_stretch(img, m5)
    g5["out_f64"] = color.apply_shared_percentile_stretch(img.astype(np.float64), m5)
    g5["out_5_95"] = color.apply_shared_percentile_stretch(img, m5, 5, 95)
    g5["lohi"] = np.array([np.percentile(img[..., c][m5], [2, 98]) for c in range(3)])
    g5["robust_norm_rgb"] = color.robust_norm_rgb(img, m5)
    g5["robust_norm"] = color.robust_norm(img[..., 0])
    big = (rng.random((300, 257, 3)) ** 2).astype(np.float32)
    mb = rng.random((300, 257)) > 0.5
    g5["big_seed"] = np.array(5)
    g5["big_lohi"] = np.array([np.percentile(big[..., c][mb], [2, 98]) for c in range(3)])
    g5["big_out_checksum"] = np.array(color.apply_shared_percentile_stretch(big, mb).astype(np.float64).sum())
    save("g5_stretch", **g5)

    # ---- G6: per-band all-pixel linear fit (notebook cell 72) ------------------------------------
    cal = nbf["calibrate_pseudo_to_real_linear"]
    rng = np.random.default_rng(6)
    ps = (rng.random((4, 24, 20)) * 0.5).astype(np.float32)
    rl = (ps * np.array([1.1, 0.9, 1.0, 1.3], dtype=np.float32)[:, None, None] + 0.02
          + 0.01 * rng.standard_normal(ps.shape)).astype(np.float32)
    ps[0, 0, 0] = np.nan
    rl[1, 0, 1] = np.nan
    ps[2, 3, :] = -0.01
    vm = rng.random((24, 20)) > 0.1
    vm_few = np.zeros_like(vm)
    vm_few[:2, :20] = True                                # 40 px < 50 -> (1, 0)
    corr, params = cal(ps, rl, vm)
    corr_few, params_few = cal(ps, rl, vm_few)
    corr_mv, params_mv = cal(ps, rl, vm, min_valid=0.1)
    save("g6_lsq", pseudo=ps, real=rl, mask=vm, mask_few=vm_few, corrected=corr,
         params=np.array(params), corrected_few=corr_few, params_few=np.array(params_few),
         corrected_mv=corr_mv, params_mv=np.array(params_mv))

    # ---- G7: ridge pipeline of Spectral_matching.ipynb (sklearn, float64) --------------------------
    from sklearn.linear_model import Ridge
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import PolynomialFeatures, StandardScaler
    rng = np.random.default_rng(7)
    N, C, T = 1500, 10, 6
    base = rng.random((N, 3))
    mix = rng.random((3, C))
    Xdn = np.round(600 + 4600 * np.clip(base @ mix / 1.5 + 0.02 * rng.standard_normal((N, C)), 0, 1)).astype(np.uint16)
    Wt = rng.random((3, T))
    Y = np.clip(base @ Wt / 2.0 + 0.01 * rng.standard_normal((N, T)), -0.01, 0.6).astype(np.float32)
    X64 = Xdn.astype(np.float64)
    Yl = smf["logit"](Y.astype(np.float64))
    model = Pipeline([("scaler", StandardScaler()),
                      ("poly", PolynomialFeatures(degree=3, include_bias=False)),
                      ("ridge", Ridge(alpha=1.0))])
    model.fit(X64, Yl)
    Xte = np.round(600 + 4600 * rng.random((16, 16, C))).astype(np.uint16)
    pred_logit = model.predict(Xte.reshape(-1, C).astype(np.float64))
    powers = model.named_steps["poly"].powers_
    save("g7_ridge", X=Xdn, Y=Y, Ylogit=Yl, mean=model.named_steps["scaler"].mean_,
         scale=model.named_steps["scaler"].scale_, coef=model.named_steps["ridge"].coef_,
         intercept=model.named_steps["ridge"].intercept_, powers=powers.astype(np.int8),
         Xtest=Xte, pred_logit=pred_logit, pred=smf["sigmoid"](pred_logit),
         sigmoid_probe=smf["sigmoid"](np.array([-80.0, -50.0, -1.0, 0.0, 2.5, 50.0, 80.0])),
         logit_probe=smf["logit"](np.array([-0.01, 0.0, 1e-4, 0.3, 0.9999, 1.0, 1.2])),
         subsample_285_32=smf["subsample_bands_evenly"](285, 32))

    # ---- G11: the ridge pipeline at the notebook's own shapes (Spectral_matching.ipynb raw :426 "Train pixels: (10000, 10)
    #      Targets: (10000, 32)", :634 "Pred EMIT @10m: (32, 600, 600)"), predict_cube_logit run FROM THE NOTEBOOK on the
    #      600 x 600 cube - stored as a strided sample plus per-band checksums (the full output is 46 MB) - and a second
    #      sklearn model with 97 targets (the many-target predict kernels, T 97-512, against sklearn itself) --------------
    rng = np.random.default_rng(11)
    N, C, T = 10000, 10, 32
    base = rng.random((N, 3))
    mix = rng.random((3, C))
    Xdn = np.round(600 + 4600 * np.clip(base @ mix / 1.5 + 0.02 * rng.standard_normal((N, C)), 0, 1)).astype(np.uint16)
    Wt = rng.random((3, T))
    Yu = np.round(1e4 * np.clip(base @ Wt / 2.0 + 0.01 * rng.standard_normal((N, T)), 0.0, 0.6)).astype(np.uint16)
    Y = Yu.astype(np.float32) * np.float32(1e-4)              # the targets in the tile format's precision (u16 x 1e-4)
    # The notebook feeds float32 arrays, so scikit-learn runs its whole pipeline in float32 there; on this (cond ~ 1e9)
    # system that run deviates from the float64 evaluation of the SAME pipeline by ~7e-3 in logit / 1e-3 in reflectance
    # (stored below as f32_pipeline_dev_*).  The fixture pins the float64 evaluation, as g7 does.
    Yl = smf["logit"](Y.astype(np.float64), eps=1e-4)
    model = Pipeline([("scaler", StandardScaler()),
                      ("poly", PolynomialFeatures(degree=3, include_bias=False)),
                      ("ridge", Ridge(alpha=1.0))])
    model.fit(Xdn.astype(np.float64), Yl)
    model32 = Pipeline([("scaler", StandardScaler()),
                        ("poly", PolynomialFeatures(degree=3, include_bias=False)),
                        ("ridge", Ridge(alpha=1.0))])
    model32.fit(Xdn.astype(np.float32), smf["logit"](Y, eps=1e-4))
    p64_, p32_ = model.predict(Xdn[:2000].astype(np.float64)), model32.predict(Xdn[:2000].astype(np.float32)).astype(np.float64)
    dev_logit, dev_refl = np.abs(p64_ - p32_).max(), np.abs(smf["sigmoid"](p64_) - smf["sigmoid"](p32_)).max()
    # the 10 m cube: a 100 x 100 coarse cube repeated 6 x 6 plus a fixed dither (the test rebuilds it from `cube_coarse`
    # with the two statements below), NaN at `cube_nan` (c, i, j) and the nodata value at `cube_nd`
    coarse = np.round(600 + 4600 * np.clip(rng.random((100, 100, 3)) @ mix / 1.5, 0, 1)).astype(np.uint16)
    coarse = np.ascontiguousarray(np.moveaxis(coarse, -1, 0))                       # (10, 100, 100)
    ci, ii, jj = np.meshgrid(np.arange(C), np.arange(600), np.arange(600), indexing="ij")
    cube = np.repeat(np.repeat(coarse, 6, axis=1), 6, axis=2).astype(np.float32) + ((7 * ii + 13 * jj + 5 * ci) % 17 - 8).astype(np.float32)
    cube_nan = np.array([[0, 0, 0], [3, 17, 500], [9, 599, 599], [5, 300, 301]])
    cube_nd = np.array([[1, 2, 3], [8, 400, 77], [2, 599, 0]])
    for c_, i_, j_ in cube_nan:
        cube[c_, i_, j_] = np.nan
    for c_, i_, j_ in cube_nd:
        cube[c_, i_, j_] = -9999.0
    pred = smf["predict_cube_logit"](model, cube, nodata=-9999.0)                   # the notebook's function, (32, 600, 600) float32
    assert pred.shape == (32, 600, 600) and pred.dtype == np.float32
    fin = np.isfinite(pred)
    Xtr = Xdn[:512].astype(np.float32)
    # second model: 97 targets on 2000 of the pixels (the sliced predict kernel's range), checked on an odd pixel count
    T2 = 97
    Wt2 = rng.random((3, T2))
    Y2u = np.round(1e4 * np.clip(base[:2000] @ Wt2 / 2.0 + 0.01 * rng.standard_normal((2000, T2)), 1e-3, 0.6)).astype(np.uint16)
    Yl2 = smf["logit"]((Y2u.astype(np.float32) * np.float32(1e-4)).astype(np.float64))
    model2 = Pipeline([("scaler", StandardScaler()),
                       ("poly", PolynomialFeatures(degree=3, include_bias=False)),
                       ("ridge", Ridge(alpha=1.0))])
    model2.fit(Xdn[:2000].astype(np.float64), Yl2)
    Xte2 = np.round(600 + 4600 * np.clip(rng.random((257, 3)) @ mix / 1.5, 0, 1)).astype(np.uint16)
    save("g11_ridge_notebook_shapes", X=Xdn, Yu16=Yu,
         mean=model.named_steps["scaler"].mean_, scale=model.named_steps["scaler"].scale_,
         coef=model.named_steps["ridge"].coef_, intercept=model.named_steps["ridge"].intercept_,
         train_pred_logit=model.predict(Xtr.astype(np.float64)[:512]),
         cube_coarse=coarse, cube_nan=cube_nan, cube_nd=cube_nd, nodata=np.float32(-9999.0),
         pred_sample=pred[:, ::7, ::11], pred_nan_count=np.int64((~fin).sum()),
         pred_band_sum=np.where(fin, pred, 0).sum(axis=(1, 2), dtype=np.float64),
         pred_band_sumsq=(np.where(fin, pred, 0).astype(np.float64) ** 2).sum(axis=(1, 2)),
         pred_rows_299=pred[:, 299, :], f32_pipeline_dev_logit=dev_logit, f32_pipeline_dev_refl=dev_refl,
         Y2u16=Y2u, coef2=model2.named_steps["ridge"].coef_.astype(np.float32),
         intercept2=model2.named_steps["ridge"].intercept_, Xtest2=Xte2,
         pred2_logit=model2.predict(Xte2.astype(np.float64)).astype(np.float32))

    # ---- G8: histogram matching ------------------------------------------------------------------------
    rng = np.random.default_rng(8)
    src = np.round(rng.random((16, 14, 3)) * 40) / 40
    ref = np.round(rng.random((16, 14, 3)) ** 2 * 50) / 50
    m8 = rng.random((16, 14)) > 0.25
    save("g8_histmatch", src=src, ref=ref, mask=m8, out=color.histogram_match_rgb(src, ref, m8))

    # ---- G9: fit_ot_poly_rgb identity fallback (<200 rows; the only branch runnable without POT) --------
    rng = np.random.default_rng(9)
    a = rng.random((20, 20, 3))
    b = rng.random((20, 20, 3))
    m9 = np.zeros((20, 20), bool)
    m9[:9, :] = True                                      # 180 rows
    a_nan = a.copy()
    a_nan[9:12, :, 1] = np.nan
    m9b = np.zeros((20, 20), bool)
    m9b[:12, :] = True                                    # 240 rows, 60 non-finite in X -> 180
    save("g9_fit_fallback", src=a, ref=b, mask=m9, src_nan=a_nan, mask_b=m9b,
         coeffs_deg2=poly["fit_ot_poly_rgb"](a, b, m9, deg=2),
         coeffs_deg4=poly["fit_ot_poly_rgb"](a, b, m9, deg=4),
         coeffs_nan_deg3=poly["fit_ot_poly_rgb"](a_nan, b, m9b, deg=3))

    # ---- G10: uint16 tile quantisation (tiles_helpers/utils.py:362-374, executed from the reference file) ------
    quant = ref_loader.load_tile_quantiser()
    rng = np.random.default_rng(10)
    t = np.clip(rng.normal(0.15, 0.12, (5, 12, 11)), -0.01, 0.7).astype(np.float32)
    # exact .5 ties in float32 after the product, band-fill / nodata values, range ends, non-finite samples
    specials = np.array([0.00005, 0.00015, 0.00025, 0.00035, 1.22075, 6.5534, 6.55345, 6.5535, 7.0, 100.0, -0.01,
                         -0.00004, -9999.0, 3.0e5, 1.0e30, -1.0e30, 2.14748e5, 2.147484e5, np.inf, -np.inf, np.nan, 0.0],
                        dtype=np.float32)
    t.reshape(-1)[: specials.size] = specials
    t[3, 5, :] = np.nan
    t[1, 2, 3] = -9999.0
    with np.errstate(invalid="ignore"):
        save("g10_tile_u16", tile=t, u16_plain=quant(t), u16_srcnodata=quant(t, src_nodata=-9999.0),
             u16_scale2000_nd4095=quant(t, src_nodata=None, emit_scale=2000.0, emit_nodata_u16=4095))


if __name__ == "__main__":
    main()
