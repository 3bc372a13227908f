"""Stage-by-stage error of match_pair against the oracle driver (poly_regression.py:96-172): where does the budget go?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import s2_emit
from oracle import oracle_np as onp
srf = onp.synthetic_srf(); w, good = onp.synthetic_wavelengths()
H, W, f = 40, 36, 6
R = onp.synthetic_cube(H, W, seed=21)
R[3, 4, :] = -0.01; R[10, 10, 50] = np.nan
rng = np.random.default_rng(5)
ps = onp.pseudo_s2_srf_integral(R, w, srf, good)
rgb60 = np.stack([ps["B4"], ps["B3"], ps["B2"]], -1)
hi = np.repeat(np.repeat(np.nan_to_num(rgb60, nan=0.1), f, 0), f, 1)
s2_hi = np.clip(np.clip(hi, 0, None) / 0.45, 0, None) ** 0.8 * 255 + rng.normal(0, 6, hi.shape)
s2_hi = np.clip(s2_hi, 0, 255).astype(np.uint8)
for use_ot in (False, True):
    deg = 4 if use_ot else 3
    ref = onp.match_pair_reference(R, w, srf, good, s2_hi, f, deg=deg, use_ot=use_ot, n_samples=600)
    got = s2_emit.match_pair(R, w, srf, good, s2_hi, f, deg=deg, use_ot=use_ot, n_samples=600)
    # oracle internals re-derived
    emit_sim = np.stack([ps[b] for b in ("B2", "B3", "B4")], 0).astype(np.float32)
    valid = ref["valid60"]
    e_rgb = np.transpose(emit_sim[[2, 1, 0]], (1, 2, 0))
    lohi_ref = np.array([np.percentile(e_rgb[..., c][valid], [2, 98]) for c in range(3)])
    print("use_ot", use_ot, "deg", deg)
    print("  lohi emit  max abs diff", np.abs(got["lohi_emit_60m"] - lohi_ref).max(), "range", (lohi_ref[:, 1] - lohi_ref[:, 0]))
    print("  s2_rgb_60m_n   max abs", np.abs(got["s2_rgb_60m_n"] - ref["s2_rgb_60m_n"]).max())
    xs = np.linspace(0, 1, 33)
    print("  poly on [0,1]  max abs", max(np.abs(np.polyval(got["coeffs"][c], xs) - np.polyval(ref["coeffs"][c], xs)).max() for c in range(3)))
    print("  coeffs ref", ref["coeffs"][0], " got", got["coeffs"][0])
    d60 = np.abs(got["emit_rgb_matched_60m"][valid] - ref["emit_rgb_matched_60m"][valid])
    print("  matched 60m    max abs", d60.max(), " p99", np.percentile(d60, 99))
    m10 = ref["mask10"]
    d10 = np.abs(got["emit_rgb_10m_matched"][m10] - ref["emit_rgb_10m_matched"][m10])
    print("  matched 10m    max abs", d10.max(), " p99", np.percentile(d10, 99))
    # the same polynomial (oracle's) applied to OUR stretched input: isolates the apply stage
    # slope of the fitted polynomials
    print("  max |p'| on [0,1]", max(np.abs(np.polyval(np.polyder(ref["coeffs"][c]), xs)).max() for c in range(3)))
