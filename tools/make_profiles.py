#!/usr/bin/env python3
"""Turn a tools/collect_profiles.sh run (gpurun_out/<dir>) into the committed summaries under profiles/.

    python tools/make_profiles.py r02 gpurun_out/r02final
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def jload(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


def kstats(d, flt="hsr::"):
    f = glob.glob(os.path.join(d, "*kernel_stats.csv"))
    rows = []
    if f:
        for r in csv.DictReader(open(f[0])):
            if flt in r["Name"]:
                rows.append((r["Name"].split("(")[0].replace("void ", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                             int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
    return rows


def pmc_bytes(d, kern):
    """FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 per launch of kernels whose name contains `kern`."""
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(os.path.join(d, c, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            tot[c] = sum(vals) / len(vals)
    if len(tot) == 2:
        return tot["FETCH_SIZE"] * 2048, tot["WRITE_SIZE"] * 1024
    return None


def main():
    tag, src = sys.argv[1], sys.argv[2]
    P = os.path.join(ROOT, "profiles")
    os.makedirs(P, exist_ok=True)
    # ---- headline
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), tag, os.path.join(src, "trace"), os.path.join(src, "pmc")],
                   stdout=subprocess.DEVNULL, check=True)
    for name in ("bench_n1.json", "bench_under_rocprof.json", "bench_u16.json", "bench_u16_fast.json", "bench_mosaic8.json", "bench_mosaic8_batched.json",
                 "bench_force_exchange.json", "bench_launcher.json"):
        if os.path.isfile(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(P, f"{tag}_{name}.log"))
    f = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
    if f:
        rows = list(csv.DictReader(open(f[0])))
        with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=rows[0].keys())
            w.writeheader()
            for r in rows:
                if "hsr::" in r["Name"]:
                    r["Name"] = r["Name"].split("(")[0]
                    w.writerow(r)
    # ---- fresh processes
    out = [f"# {tag}: the headline over fresh processes (one box, back to back)\n",
           "`python bench.py --steps 100 --no-cpu-baseline` four times, then `--gpus 1 --steps 20 --warmup 5` (the driver's command) three times.  "
           "`cold` = the same step in the same process before the placement trials, on first allocations, without the settle phase.\n",
           "| run | steps | ms/step | K1+K2 ms (HIP events) | frac of 8 TB/s | cold ms/step | cold K1+K2 ms | placement trials (ms) |", "|---|---|---|---|---|---|---|---|"]
    for pat, st in (("fresh_%d.json", 100), ("s20_%d.json", 20)):
        for i in range(1, 7):
            d = jload(os.path.join(src, pat % i))
            if d:
                cold = d.get("cold") or {}
                out.append(f"| {i} | {st} | {d['ms_per_step']} | {d['roofline']['kernel_ms']} | {d['roofline']['frac']} | {cold.get('ms_per_step')} | {cold.get('kernel_ms')} | {d['config'].get('placement', {}).get('trials_ms')} |")
    d = jload(os.path.join(src, "s20_plain.json"))
    if d:
        out.append(f"\nThe driver's command with `--placement-trials 0 --settle-ms 0` (what a caller who just allocates and runs gets, timed region "
                   f"started from an idle GPU): {d['ms_per_step']} ms/step, K1+K2 {d['roofline']['kernel_ms']} ms.")
    open(os.path.join(P, f"{tag}_fresh_processes.md"), "w").write("\n".join(out) + "\n")
    # ---- exchange pipeline (r04)
    def read(name):
        pth = os.path.join(src, name)
        return open(pth, errors="replace").read() if os.path.isfile(pth) else ""
    sw = read("exchange_sweep.md")
    if sw:
        out = [f"# {tag}: the fused pipeline WITH the exchange, issued from C (hsr_pipeline_create_exchange; one-rank RCCL communicator through hsr_comm_*)\n",
               "`tools/bench_sweep.py`: one fresh `bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-probe <arguments>` process per row, one box, back to back.  "
               "`--force-exchange`: the multi-rank code path with a one-rank communicator (RCCL launches no kernel for one rank); `--fake-collective-us 15`: a stand-in "
               "kernel (2 or 8 workgroups of 256 threads, 48 KB LDS) resident for 15 us where the collective's kernel would run; `--pipeline on`: round 3's two-slot "
               "pipeline (collective through torch.distributed); `--height 128`: the row block of an 8-way strong-scaling run.\n", sw]
        tl = read("exchange_timeline.md")
        if tl:
            out += ["## Where the side stream's kernels run (`rocprofv3 --kernel-trace` of `bench.py --force-exchange --fake-collective-us 15`, `tools/exchange_timeline.py`)\n",
                    "```", tl.rstrip(), "```\n"]
        d = jload(os.path.join(src, "bench_launcher.json"))
        if d:
            out.append("`python bench.py --gpus 1 --launcher --force-exchange` (parent -> torchrun -> rank -> RCCL group from the launcher's rendezvous -> hsr_comm_init): " +
                       json.dumps({k: d[k] for k in ("value", "ms_per_step", "rccl_ranks", "world_size", "host_issue_us_per_step") if k in d} |
                                  {"exchange_transport": d["config"].get("exchange_transport")}) + "\n")
        open(os.path.join(P, f"{tag}_exchange_pipeline.md"), "w").write("\n".join(out) + "\n")
    # ---- mosaic (r04)
    out = [f"# {tag}: resident mosaic on one GPU (`bench.py --tiles-per-gpu T`: T x 1024 x 1024 x 285 tiles, ONE fit per step; BASELINE configs[3]/[4] per-GPU work)\n",
           "| run | form | ms/step | kernel ms | frac (cube bytes / kernel time) | step_frac_of_peak | placement (kept ms) |", "|---|---|---|---|---|---|---|"]
    anym = False
    for name in ("bench_mosaic8.json", "bench_mosaic8_batched.json", "bench_mosaic4.json"):
        d = jload(os.path.join(src, name))
        if d:
            anym = True
            mo = d["config"].get("mosaic") or {}
            out.append(f"| {name} | {mo.get('form')} | {d['ms_per_step']} | {d['roofline']['kernel_ms']} | {d['roofline']['frac']} | {d['roofline']['step_frac_of_peak']} | "
                       f"{(mo.get('placement') or {}).get('kept_ms')} |")
    d = jload(os.path.join(src, "bench_mosaic8.json"))
    if d and (d["config"].get("mosaic") or {}).get("placement"):
        out.append("\nDense map of device memory seen by K1 (ms per launch on 1.25 GB candidates allocated back to back; `place_mosaic`):\n\n```\n" +
                   json.dumps(d["config"]["mosaic"]["placement"].get("candidate_ms")) + "\n```")
    if anym:
        open(os.path.join(P, f"{tag}_mosaic.md"), "w").write("\n".join(out) + "\n")
    # ---- small select, 13 bands (r04)
    for name, dst, head in (("sel_small.log", f"{tag}_select_small.md", "hsr_percentile_limits through ctypes with a preallocated workspace (tools/dbg/sel_small_c.py): GPU time per call"),
                            ("nb13.log", f"{tag}_k1_13_bands.md", "K1+K2 with 12 bands (rows of 12) and with all 13 Sentinel-2 bands (rows of 16) - tools/dbg/k1_nb13.py")):
        txt = read(name)
        if txt:
            open(os.path.join(P, dst), "w").write(f"# {tag}: {head}\n\n```\n" + "".join(l for l in txt.splitlines(True) if "amdgpu.ids" not in l) + "```\n")
    # ---- batch
    out = [f"# {tag}: batched small tiles (tools/bench_batch.py; 100 x 100 x 285 tiles, deg 3)\n"]
    for name in ("batch_f32.json", "batch_u16.json", "batch_f32_t64.json"):
        d = jload(os.path.join(src, name))
        if d:
            out.append(f"`{name}`: " + json.dumps(d) + "\n")
    ks = kstats(os.path.join(src, "trace_batch"))
    if ks:
        out += ["Per kernel, `rocprofv3 --kernel-trace --stats -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 2`:\n",
                "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
        out += [f"| `{n}` | {c} | {a:.2f} | {lo:.2f} | {hi:.2f} |" for n, c, a, lo, hi in ks]
    b = pmc_bytes(os.path.join(src, "pmc_batch"), "srf_kernel<3, true, true, 64, true, true>")
    if b:
        algo = 256 * 10000 * (285 * 4 + 12 * 4 + 12 * 4)
        out.append(f"\nHBM traffic of the batched K1+K2 launch (T = 256): read {b[0]/1e6:.1f} MB + write {b[1]/1e6:.1f} MB = {(b[0]+b[1])/1e6:.1f} MB "
                   f"against {algo/1e6:.1f} MB algorithmic (cube + targets + planes) = {(b[0]+b[1])/algo:.3f}x.")
    open(os.path.join(P, f"{tag}_batch_tiles.md"), "w").write("\n".join(out) + "\n")
    # ---- u16
    out = [f"# {tag}: uint16 tiles (bench.py --cube u16 [--u16-fast])\n", "| run | ms/step | K1+K2 ms | frac of 8 TB/s on cube bytes |", "|---|---|---|---|"]
    for name in ("bench_u16.json", "bench_u16_fast.json", "bench_u16_off.json"):
        d = jload(os.path.join(src, name))
        if d:
            out.append(f"| {name} | {d['ms_per_step']} | {d['roofline']['kernel_ms']} | {d['roofline']['frac']} |")
    ks = kstats(os.path.join(src, "trace_u16"))
    if ks:
        out += ["\n| kernel (rocprofv3) | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
        out += [f"| `{n}` | {c} | {a:.2f} | {lo:.2f} | {hi:.2f} |" for n, c, a, lo, hi in ks]
    tf = os.path.join(P, "traffic.json")
    t = json.load(open(tf)) if os.path.isfile(tf) else {}
    b = pmc_bytes(os.path.join(src, "pmc_u16_plain"), "srf_u16_ring_kernel<3") or pmc_bytes(os.path.join(src, "pmc_u16"), "false, false>")
    if b:
        out.append(f"\nHBM traffic of the uint16 K1+K2 launch (`--pipeline off`): read {b[0]/1e6:.1f} MB + write {b[1]/1e6:.1f} MB = {(b[0]+b[1])/1e6:.1f} MB "
                   f"(algorithmic: 597.7 MB cube + 50.3 MB targets + 50.3 MB planes = 698.3 MB).")
        t["srf_u16_kernel_hbm_bytes_per_launch"] = int(b[0] + b[1])
        t["source_u16"] = f"profiles/{tag}_u16_tiles.md (separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --cube u16 --pipeline off)"
    b = pmc_bytes(os.path.join(src, "pmc_u16"), "false, true>")          # the APPLY variant: the fused pipeline's launch
    if b:
        out.append(f"\nHBM traffic of the fused uint16 launch (K1+K2 + K3 of an older tile + tail fit): read {b[0]/1e6:.1f} MB + write {b[1]/1e6:.1f} MB = "
                   f"{(b[0]+b[1])/1e6:.1f} MB (algorithmic: 698.3 MB + 100.7 MB of K3 = 799.0 MB).")
        t["srf_u16_fused_kernel_hbm_bytes_per_launch"] = int(b[0] + b[1])
        t["source_u16_fused"] = f"profiles/{tag}_u16_tiles.md (separate --pmc passes of bench.py --cube u16, fused pipeline)"
    json.dump(t, open(tf, "w"), indent=1)
    open(os.path.join(P, f"{tag}_u16_tiles.md"), "w").write("\n".join(out) + "\n")
    # ---- auxiliary kernels
    ax = os.path.join(src, "aux.log")
    if os.path.exists(ax):
        lines = [l.rstrip() for l in open(ax) if l.startswith("|")]
        open(os.path.join(P, f"{tag}_aux_kernels.md"), "w").write(
            f"# {tag}: auxiliary kernels of the path (`tools/bench_aux.py`, HIP events, MI355X) - resamplers (f1), exact masked percentiles (a4), "
            "validity mask, K3 alone\n\n" + "\n".join(lines) + "\n" +
            ("\nThe reference driver end to end (`tools/bench_match_pair.py`, device-resident inputs, host-timed with a synchronisation):\n\n```\n" +
             "".join(l for l in open(os.path.join(src, "match_pair.log")) if l.startswith("match_pair")) + "```\n"
             if os.path.exists(os.path.join(src, "match_pair.log")) else ""))
    # ---- rehearsals
    out = [f"# {tag}: multi-rank control flow rehearsed on one GPU - the JSON lines (bench.py started WITHOUT a launcher: it starts its own ranks)\n"]
    d = jload(os.path.join(src, "bench_gloo4.json"))
    if d:
        out.append("`python bench.py --gpus 4 --backend gloo --same-device --height 256 --width 256`: " + json.dumps({k: d[k] for k in ("value", "n_gpus", "world_size", "exchange_ranks", "rccl_ranks", "scaling", "ms_per_step", "degraded") if k in d}) + "\n")
    d = jload(os.path.join(src, "bench_strong2.json"))
    if d:
        out.append("`python bench.py --gpus 2 --backend gloo --same-device --scaling strong` (ONE 1024 x 1024 cube, two 512-row blocks): " +
                   json.dumps({k: d[k] for k in ("value", "n_gpus", "scaling", "ms_per_step", "degraded") if k in d} | {"rows_per_gpu": d["config"]["rows_per_gpu"]}) + "\n")
    e = os.path.join(src, "two_ranks_one_gpu.err")
    if os.path.isfile(e):
        msg = [l for l in open(e) if "[bench]" in l]
        out.append("2 ranks over RCCL on a 1-GPU box: " + (msg[0].strip() if msg else "(no message)") + "\n")
    d = jload(os.path.join(src, "bench_force_exchange.json"))
    if d:
        out.append("one-rank RCCL all-reduce in the loop, pipelined submit(): " + json.dumps({"ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms"], "pipeline": d["config"]["pipeline"]}) + "\n")
    d = jload(os.path.join(src, "bench_mosaic8.json"))
    if d:
        out.append("8 x 1024 x 1024 x 285 tiles resident on one GPU, one global fit per step (`--tiles-per-gpu 8`): " +
                   json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "step_frac_of_peak": d["roofline"]["step_frac_of_peak"]}) + "\n")
    open(os.path.join(P, f"{tag}_rehearsal_lines.md"), "w").write("\n".join(out) + "\n")
    # ---- a9
    out = [f"# {tag}: variant a9 (K4, polynomial-ridge fusion) - `tools/bench_ridge.py`, rocprofv3 kernel trace and PMC, MI355X\n"]
    rl = os.path.join(src, "ridge.log")
    if os.path.isfile(rl):
        out.append("```")
        out += [l.strip() for l in open(rl) if l.startswith("{")]
        out.append("```\n")
    ks = kstats(os.path.join(src, "trace_ridge"))
    if ks:
        out += ["| kernel (rocprofv3 --kernel-trace --stats, all three configurations of bench_ridge.py) | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
        out += [f"| `{n}` | {c} | {a:.2f} | {lo:.2f} | {hi:.2f} |" for n, c, a, lo, hi in ks]
    agg = {}
    for f in glob.glob(os.path.join(src, "pmc_ridge", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "hsr::" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if not any(x in k for x in ("predict103", "gram_f64", "chol_")):
                    continue
                a = agg.setdefault(k, {})
                a.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                a.setdefault("_dur_us", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if agg:
        out.append("\nPMC (separate passes; averages per launch over all launches of the kernel in `bench_ridge.py`).  Matrix-pipe busy = "
                   "`SQ_VALU_MFMA_BUSY_CYCLES` / (1024 SIMDs x `GRBM_GUI_ACTIVE` / 8); `SQ_*_CYCLES` / `SQ_WAIT_*` count quad-cycles.\n")
        for k, cs in agg.items():
            avg = {c: sum(v) / len(v) for c, v in cs.items()}
            line = f"* `{k}`: " + ", ".join(f"{c} {v:,.0f}" for c, v in sorted(avg.items()) if c != "_dur_us")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
                line += f" -> matrix pipe busy {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * avg['GRBM_GUI_ACTIVE'] / 8) * 100:.1f} %"
            out.append(line)
    cs_ = os.path.join(src, "chol_stamps.log")
    if os.path.isfile(cs_):
        out += ["\nWhere the one-workgroup Cholesky factorisation spends its cycles (`tools/chol_stamps`, n = 288):\n", "```"] + [l.rstrip() for l in open(cs_)] + ["```"]
    tp_ = os.path.join(src, "time_predict.log")
    if os.path.exists(tp_):
        out += ["\n`tools/time_predict.py` (predict_cube over 1024 x 1024 pixels, random degree-3 model):\n", "```"] + [l.rstrip() for l in open(tp_) if l.startswith("T=")] + ["```"]
    gs_ = os.path.join(src, "gram_stamps.log")
    if os.path.exists(gs_):
        out += ["\nPhase timeline of `gram_f64_lds_kernel` (`tools/gram_stamps.py`: s_memrealtime at the phase boundaries of every workgroup, "
                "s_memtime around the batch loop; 29 127 rows, 288 features + T targets):\n", "```"] + [l.rstrip() for l in open(gs_) if "amdgpu.ids" not in l] + ["```"]
    open(os.path.join(P, f"{tag}_k4_ridge.md"), "w").write("\n".join(out) + "\n")
    sc = os.path.join(src, "shard_curve.log")
    if os.path.isfile(sc):
        shutil.copy(sc, os.path.join(P, f"{tag}_shard_curve.log"))
    ks = kstats(os.path.join(src, "trace_rs"))
    if ks:
        out = [f"# {tag}: slot reduction + solve kernels (`rocprofv3 --kernel-trace --stats -- python3 tools/dbg/rs_time.py`)\n",
               "512 slots of a 512 x 512 tile (single-tile kernels), 256 tiles of 100 x 100 (batched kernel, 157 slots each = 54 MB of partials).\n",
               "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
        out += [f"| `{n}` | {c} | {a:.2f} | {lo:.2f} | {hi:.2f} |" for n, c, a, lo, hi in ks if "reduce" in n or "solve" in n]
        open(os.path.join(P, f"{tag}_reduce_solve.md"), "w").write("\n".join(out) + "\n")
    for name, dst in (("feed.log", f"{tag}_feed.md"), ("k1_stamps.log", f"{tag}_k1_phase_stamps.log"), ("placement_probe.log", f"{tag}_placement_probe.log"),
                      ("state_probe.log", f"{tag}_state_probe.log"), ("k1_stamps_u16.log", f"{tag}_k1_phase_stamps_u16.log"),
                      ("probe_modes.log", f"{tag}_probe_modes.log"), ("placement_map.log", f"{tag}_placement_map.log"), ("ramp.log", f"{tag}_ramp.log")):
        s = os.path.join(src, name)
        if os.path.isfile(s):
            txt = "".join(l for l in open(s, errors="replace") if "amdgpu.ids" not in l and "Warning" not in l)
            open(os.path.join(P, dst), "w").write(("```\n" + txt + "```\n") if dst.endswith(".md") else txt)
    print("profiles written:", sorted(x for x in os.listdir(P) if x.startswith(tag)))


if __name__ == "__main__":
    main()
