#!/usr/bin/env python3
"""Secondary benchmark: variant a9 (polynomial-ridge S2 -> EMIT fusion) on the matrix cores.
Prints fit / predict times and the achieved MFMA rate (fp64 for the Gram, fp32 for the predict GEMM)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))

import torch
import s2_emit


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for (n_lo, H, W, T) in ((10_000, 600, 600, 32), (29_127, 1024, 1024, 32), (29_127, 1024, 1024, 285)):
        X = (600 + 4600 * torch.rand((n_lo, 10), generator=g, device=dev)).float()
        Y = torch.logit((0.02 + 0.5 * torch.rand((n_lo, T), generator=g, device=dev)).double())
        cube = (600 + 4600 * torch.rand((10, H, W), generator=g, device=dev)).float()
        model = s2_emit.PolyRidge(3, 1.0)
        t_fit = timed(lambda: model.fit(X, Y), 5)
        t_pred = timed(lambda: model.predict_cube(cube), 10)
        na = 288
        fit_flops = 2.0 * n_lo * na * (na + (T + 15) // 16 * 16)
        pred_flops = 2.0 * H * W * 286 * ((T + 31) // 32 * 32)
        print(json.dumps({"n_fit": n_lo, "predict_pixels": H * W, "targets": T, "fit_ms": round(t_fit, 3),
                          "predict_ms": round(t_pred, 3),
                          "gram_fp64_TFLOPs_if_all_fit_time": round(fit_flops / t_fit / 1e9, 2),
                          "predict_fp32_TFLOPs": round(pred_flops / t_pred / 1e9, 2),
                          "predict_Mpix_s": round(H * W / t_pred / 1e3, 1)}))


if __name__ == "__main__":
    main()
