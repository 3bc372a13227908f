#!/usr/bin/env python3
"""Run bench.py once per argument set (one fresh process each) and print one compact row per run.

    python tools/bench_sweep.py [--common "<args for every run>"] [--out FILE.md] "<args of run 1>" "<args of run 2>" ...

Used for the round-4 exchange-pipeline sweeps (reserved CUs x stand-in collective), profiles/r04_exchange_pipeline.md."""
import argparse
import json
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--common", default="--steps 50 --warmup 5 --no-cpu-baseline --no-probe")
    ap.add_argument("--out", default=None)
    ap.add_argument("--timeout", type=float, default=300.0)
    ap.add_argument("runs", nargs="+")
    a = ap.parse_args()
    rows = ["| bench.py arguments | ms/step | kernel ms | frac (cube bytes) | step frac | host issue us | cold ms/step | pipeline |",
            "|---|---|---|---|---|---|---|---|"]
    for run in a.runs:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + shlex.split(a.common) + shlex.split(run)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=a.timeout)
        except subprocess.TimeoutExpired:
            rows.append(f"| `{run}` | TIMEOUT | | | | | | |")
            print(rows[-1], flush=True)
            continue
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            rows.append(f"| `{run}` | FAILED rc={r.returncode} | | | | | | {r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else ''} |")
            print(rows[-1], flush=True)
            continue
        ln = json.loads(lines[-1])
        roof = ln["roofline"]
        cold = ln.get("cold") or {}
        rows.append(f"| `{run or '(default)'}` | {ln['ms_per_step']:.4f} | {roof['kernel_ms']:.4f} | {roof['frac']:.4f} | "
                    f"{roof['step_frac_of_peak']:.4f} | {ln.get('host_issue_us_per_step')} | {cold.get('ms_per_step')} | "
                    f"{str(ln['config']['pipeline'])[:70]} |")
        print(rows[-1], flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write("\n".join(rows) + "\n\n")


if __name__ == "__main__":
    main()
