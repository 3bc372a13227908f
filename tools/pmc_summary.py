#!/usr/bin/env python3
"""Aggregate rocprofv3 outputs (gpurun_out/...) into the committed summaries under profiles/.

    python tools/pmc_summary.py <round-tag> <kernel-trace-dir> [<pmc-dir>]

* kernel trace dir: output of `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...`
* pmc dir: one sub-directory per `--pmc` pass (FETCH_SIZE, WRITE_SIZE, SQ..., each collected in its own run,
  as MI355X_MICROARCH.md prescribes: TCC counters do not fit one pass).
HBM traffic per launch = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced stream (16 B per lane, global_load and LDS-DMA alike); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    tag, trace_dir = sys.argv[1], sys.argv[2]
    pmc_dir = sys.argv[3] if len(sys.argv) > 3 else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = [f"# {tag}: rocprofv3 summary (MI355X, bench.py, 1024x1024x285, deg 3)\n"]
    f = glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    hs = [r for r in rows if "hsr::" in r["Kernel_Name"]]
    dur = collections.defaultdict(list)
    for r in hs:
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out.append("## kernel trace (`rocprofv3 --kernel-trace --stats`), hsr kernels only\n")
    out.append("| kernel | calls | avg us | median us | min us | max us |\n|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        out.append(f"| `{k}` | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {statistics.median(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} |")
    srf = sorted([k for k in dur if "srf_kernel<3" in k], key=lambda k: -len(dur[k])) or [k for k in dur if "srf_u16_ring_kernel<3" in k]
    if srf:
        avg = sum(dur[srf[0]]) / len(dur[srf[0]])
        cube = 1024 * 1024 * 285 * (2 if "u16" in srf[0] else 4)
        fusedk = srf[0].rstrip().endswith("false, true>")          # the fused pipeline's launch also carries K3 of an older tile
        algo = cube + (1024 * 1024 * 8 * 12 if fusedk else 0)
        out.append(f"\nDominant kernel `{srf[0]}`: {avg/1e3:.2f} us average -> SURVEY 8(d) algorithmic bytes = the cube read once = {cube} B per launch "
                   f"-> {cube/avg:.0f} GB/s = **{cube/avg/8000:.4f} of 8 TB/s** (bench.py's roofline.frac)."
                   + (f"  With the K3 of an older tile the launch also carries (96 B per pixel: pseudo read, matched written; 12 channels) "
                      f"{algo} B -> {algo/avg:.0f} GB/s = {algo/avg/8000:.4f} (roofline.frac_launch_bytes)." if fusedk else "") + "\n")
    traffic = {}
    if pmc_dir:
        out.append("## PMC passes (`rocprofv3 --pmc <counters> --kernel-trace`), averages per launch\n")
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(pmc_dir, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "hsr::" in r["Kernel_Name"]:
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            out.append(f"### `{k}`\n\n| counter | avg per launch |\n|---|---|")
            for c, v in sorted(cs.items()):
                out.append(f"| {c} | {sum(v)/len(v):,.1f} |")
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                fe = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
                wr = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
                out.append(f"\nHBM traffic per launch: read {fe/1e6:.1f} MB (FETCH_SIZE x 1024 x 2, gfx950 correction) + "
                           f"write {wr/1e6:.1f} MB = {(fe+wr)/1e6:.1f} MB\n")
                traffic[k] = int(fe + wr)
            out.append("")
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    open(os.path.join(root, "profiles", f"{tag}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
    if traffic:
        tf = os.path.join(root, "profiles", "traffic.json")
        old = json.load(open(tf)) if os.path.isfile(tf) else {}
        per = dict(old.get("per_kernel", {}))
        per.update(traffic)
        fkey = [k for k in traffic if "srf_kernel<3" in k and k.rstrip().endswith("false, true>")]       # APPLY variant: the fused pipeline's launch
        key = [k for k in traffic if "srf_kernel<3" in k and "true, false>" in k and k not in fkey] or \
              [k for k in traffic if "srf_kernel<3" in k and k not in fkey]
        if fkey:
            old["srf_fused_kernel_hbm_bytes_per_launch"] = traffic[fkey[0]]
            old["source_fused"] = f"profiles/{tag}_rocprof_summary.md (the launch of the fused pipeline: K1+K2 + K3 of an older tile + tail fit)"
        ukey = [k for k in traffic if "srf_u16_ring_kernel<3" in k]
        old["per_kernel"] = per
        if key:
            old["srf_kernel_hbm_bytes_per_launch"] = traffic[key[0]]
            old["source"] = f"profiles/{tag}_rocprof_summary.md"
        if ukey:
            old["srf_u16_kernel_hbm_bytes_per_launch"] = traffic[ukey[0]]
            old["source_u16"] = f"profiles/{tag}_rocprof_summary.md (separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --cube u16)"
        json.dump(old, open(tf, "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main()
