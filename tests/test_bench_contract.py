"""bench.py on the CPU: importable without a GPU, argument contract, and the cpu_baseline leg (the oracle timed
on a bounded slab) produces the fields the JSON line promises."""
import importlib.util
import os
import sys

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("hsr_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
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
    monkeypatch.setattr(sys, "argv", ["bench.py", "--height", "8", "--width", "16", "--cpu-rows", "8", "--deg", "2"])
    a = bench.parse_args()
    cb = bench.cpu_baseline(a)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Mpixel*bands/s" and cb["value"] > 0
    assert "8x16x285" in cb["sample"]
    assert bench.HBM_PEAK_GBS == 8000.0
