#!/usr/bin/env python3
"""Where the side stream's kernels of the exchange pipeline run relative to K1, read off a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o x -- python3 bench.py --force-exchange [--fake-collective-us 15] ...
    python tools/exchange_timeline.py DIR [--last N]

For every side-stream kernel (gate_kernel, solve_publish_kernel, rehearsal_collective_kernel, RCCL's ncclDevKernel*) of the last N
steps: the K1 launch (srf_kernel / srf_u16_ring_kernel) it starts in, how far into that launch it starts, its duration, and whether it
ends before that K1 ends - plus the gaps between consecutive K1 launches on the caller's stream (an event or a stream wait there
would show as a bubble of ~5 us)."""
import argparse
import csv
import glob
import os
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last", type=int, default=30)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {a.dir}")
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    k1 = [r for r in rows if "srf_kernel" in r[2] or "srf_u16_ring_kernel" in r[2]]
    side = [r for r in rows if any(s in r[2] for s in ("gate_kernel", "solve_publish_kernel", "rehearsal_collective_kernel", "ncclDevKernel"))]
    k1 = k1[-a.last:]
    if not k1:
        raise SystemExit("no K1 launches in the trace")
    t_first = k1[0][0]
    gaps = [(k1[i + 1][0] - k1[i][1]) / 1e3 for i in range(len(k1) - 1)]
    durs = [(e - s) / 1e3 for s, e, *_ in k1]
    print(f"K1 launches looked at: {len(k1)}; duration us: mean {statistics.mean(durs):.1f} min {min(durs):.1f} max {max(durs):.1f}; "
          f"queue(s) {sorted(set(r[3] for r in k1))}")
    if gaps:
        print(f"gap between consecutive K1 launches (end -> next start) us: mean {statistics.mean(gaps):.2f} median {statistics.median(gaps):.2f} "
              f"max {max(gaps):.2f}")
    print("side-stream kernels (queues %s):" % sorted(set(r[3] for r in side if r[0] >= t_first)))
    print("| kernel | in K1 launch # | starts at (us into K1) | duration us | ends before that K1 ends | K1 duration us |")
    print("|---|---|---|---|---|---|")
    under = total = 0
    for s, e, name, q, _ in side:
        if s < t_first:
            continue
        host = [i for i, (ks, ke, *_r) in enumerate(k1) if ks <= s < ke]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:48]
        total += 1
        if host:
            i = host[0]
            under += 1
            print(f"| {short} | {i} | {(s - k1[i][0]) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {'yes' if e <= k1[i][1] else 'no (+%.1f us)' % ((e - k1[i][1]) / 1e3)} | {durs[i]:.1f} |")
        else:
            print(f"| {short} | between launches | - | {(e - s) / 1e3:.1f} | - | - |")
    print(f"{under} of {total} side-stream kernels START inside a K1 launch")


if __name__ == "__main__":
    main()
