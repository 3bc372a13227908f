#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite file (rocprofv3 --kernel-trace writes *_results.db by default on
ROCm 7.2): calls, total / average / min / max duration in us, share of GPU time - the `--stats` table, as CSV text.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [substring-filter]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
    for n, c, t, a, lo, hi in rows:
        if flt and flt not in n:
            continue
        short = n.split("(")[0][:110]
        print(f"\"{short}\",{c},{t/1e3:.1f},{a/1e3:.2f},{lo/1e3:.2f},{hi/1e3:.2f},{100*t/total:.2f}")


if __name__ == "__main__":
    main()
