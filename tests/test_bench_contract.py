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
    assert nworkers == 2 and npids == 2          # every worker is forked up front, before anything could touch a GPU
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


def test_bench_self_launch_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` starts its own ranks; with fewer than N visible GPUs (none in the build container) it
    must end within seconds, non-zero, with one clear line - before any rank is started."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""


def test_bench_self_launch_relays_one_line_and_the_exit_status(tmp_path, monkeypatch):
    """The launcher half of bench.py against a stand-in rank program: rank 0's JSON line is relayed once, noise on the
    children's stdout is dropped, a failing rank makes the parent fail, a silent job is an error."""
    bench = _load_bench()
    fake = tmp_path / "fake_bench.py"
    fake.write_text(
        "import os, sys, json\n"
        "mode = sys.argv[-1]\n"
        "rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])\n"
        "print('RCCL version banner on stdout')\n"
        "if mode == 'fail' and rank == 1: sys.exit(7)\n"
        "if mode != 'silent' and rank == 0: print(json.dumps({'metric': 'm', 'n_gpus': world}))\n")
    monkeypatch.setattr(bench, "__file__", str(fake))
    a = bench.parse_args(["--gpus", "2", "--same-device", "--backend", "gloo", "--deadline", "120"])
    import contextlib, io
    for mode, want in (("ok", 0), ("fail", None), ("silent", 4)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rc = bench.self_launch(a, [mode])
        if mode == "ok":
            assert rc == 0 and buf.getvalue().strip().splitlines() == ['{"metric": "m", "n_gpus": 2}']
        elif mode == "fail":
            assert rc != 0 and buf.getvalue() == ""
        else:
            assert rc == want and buf.getvalue() == ""
