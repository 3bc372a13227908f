"""bench.py on the CPU: importable without a GPU, argument contract, and the cpu_baseline leg (the oracle timed
on a bounded slab) produces the fields the JSON line promises."""
import importlib.util
import os
import sys

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("hsr_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hsr_bench"] = mod          # the CPU pool pickles its worker functions by module name
    spec.loader.exec_module(mod)
    return mod


def test_bench_arguments_and_cpu_baseline(monkeypatch):
    bench = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.height, a.width, a.bands, a.deg) == (1, 1024, 1024, 285, 3)      # BASELINE.json configs[2]
    assert a.steps > 0 and a.warmup >= 0 and a.coeff_sync == "allreduce" and a.pipeline == "auto"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)
    assert a.tiles_per_gpu == 1 and a.k1_launches >= 20
    monkeypatch.setattr(sys, "argv", ["bench.py", "--height", "8", "--width", "16", "--cpu-rows", "4", "--deg", "2",
                                      "--cpu-workers", "2", "--tiles-per-gpu", "8"])
    a = bench.parse_args()
    assert a.tiles_per_gpu == 8
    # the CPU leg on a small cube: single thread on a slab, row-sharded pool on the whole cube, and the parity numbers
    # the JSON line carries (here the "GPU" images are the oracle's own, so the error must be exactly 0)
    import numpy as np
    from oracle import oracle_np as onp
    srf = onp.synthetic_srf()
    w, good = onp.synthetic_wavelengths(285)
    R = onp.synthetic_cube(8, 16, 285, seed=0)
    ps = onp.pseudo_s2_srf_integral(R, w, srf, good)
    names = [k for k, v in ps.items() if v is not None]
    real = onp.synthetic_real_planes(np.stack([ps[k] for k in names]).astype(np.float32))
    pseudo, coeffs, matched, _ = onp.fuse_lsq_reference(R, w, srf, good, real, 2)
    pool, nworkers, npids = bench.start_cpu_pool(a)
    assert nworkers == 2 and 1 <= npids <= 2
    cb, err = bench.cpu_baseline(a, pool, nworkers, R, real, pseudo, matched)
    pool.shutdown()
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Mpixel*bands/s" and cb["value"] > 0
    assert "8x16x285" in cb["sample"] and "first 4 rows" in cb["sample"]
    assert cb["all_cores"]["cores"] == 2 and cb["all_cores"]["value"] > 0
    assert err["pseudo"] == 0.0 and err["matched"] == 0.0 and err["pixels_checked"] == 128
    matched[3, 2, 5] += 0.25
    pool, nworkers, _ = bench.start_cpu_pool(a)
    _, err = bench.cpu_baseline(a, pool, nworkers, R, real, pseudo, matched)
    pool.shutdown()
    assert err["matched"] > 0.2 and err["pseudo"] == 0.0
    assert bench.HBM_PEAK_GBS == 8000.0
