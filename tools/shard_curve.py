#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: what a rank of an N-GPU `bench.py --scaling strong` run executes
is the step over an H/N-row block of the 1024 x 1024 x 285 cube.  For N = 1, 2, 4, 8 this measures that step in one process
- the operator-by-operator Python path ("operators": three ctypes calls per step), step() on its prepared launches (one
call into the step executor), the pipelined submit() the multi-GPU path uses (native pipeline: fit on a side stream under
the next K1), and, with --graph, the three launches replayed from a hipGraph - and prints the time against the bytes-proportional share
of the full-tile step, i.e. the strong-scaling efficiency the compute side alone would allow (the exchange itself is
rehearsed by bench.py --force-exchange).

    python tools/shard_curve.py [--steps 400] [--graph]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--sync-every", type=int, default=0, help="synchronise the device every K calls inside the timed loop (bounds the host's run-ahead)")
    ap.add_argument("--modes", default="operators,step,submit,fused,exchange,exchange15",
                    help="exchange: the four-slot exchange pipeline over a one-rank RCCL communicator (collective issued from C, 8 CUs reserved); "
                         "exchange15: the same with a 15 us stand-in kernel where the collective of an N-rank run would be resident")
    ap.add_argument("--reserve", type=int, default=-1, help="CUs left free of K1 workgroups (-1: 8 in the pipelined modes, 0 otherwise)")
    ap.add_argument("--rows", default=None, help="comma-separated row counts instead of 1024 / ranks (pipeline resonance sweeps)")
    ap.add_argument("--host-cost", action="store_true", help="also time the ISSUE of 200 calls per mode on the 1024-row tile (GPU slower than host: the loop time is the host cost per call)")
    a = ap.parse_args()
    import torch
    from s2_emit import SpectralFusion
    from s2_emit.synthetic import device_problem
    dev = torch.device("cuda:0")
    if "exchange" in a.modes:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1, device_id=dev)
    rows = []
    full = None
    cases = [(1024 // int(x), int(x)) for x in a.ranks.split(",")] if a.rows is None else [(int(r), 1024 / int(r)) for r in a.rows.split(",")]
    for H, n in cases:
        p = device_problem(H, 1024, 285, deg=3, seed=H, device=dev)
        res = {"ranks": n, "rows": H}
        for mode in tuple(a.modes.split(",")) + (("graph",) if a.graph else ()):
            exch = mode.startswith("exchange")
            plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, min_valid=0.0, min_count=50, device=dev,
                                  reserved_cus=(8 if (mode == "submit" or exch) else 0) if a.reserve < 0 else a.reserve,
                                  fuse_apply=mode == "fused" or exch, coeff_sync="allreduce" if exch else "local", force_exchange=exch,
                                  rehearsal_collective=(15, 2) if mode == "exchange15" else None)
            if mode == "graph":
                plan.step(p.cube, p.real)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    plan.step(p.cube, p.real)
                run = g.replay
            elif mode in ("submit", "fused") or exch:
                run = lambda: plan.submit(p.cube, p.real)
            elif mode == "operators":            # the operator-by-operator Python path (what step() was before the executor)
                import torch as _t
                ev = (_t.cuda.Event(enable_timing=True), _t.cuda.Event(enable_timing=True))
                run = lambda: plan.step(p.cube, p.real, k1_events=ev)
            else:
                run = lambda: plan.step(p.cube, p.real)
            for _ in range(max(50, int(300 * n))):          # settle (profiles/r02_ramp.log) + warm-up
                run()
            if a.host_cost and H == 1024 and mode != "graph":
                if mode in ("submit", "fused") or exch:
                    plan.flush()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    run()
                res[mode + "_host"] = round((time.perf_counter() - t0) / 200 * 1e6, 2)      # issue only: no sync inside
            if mode in ("submit", "fused") or exch:
                plan.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                run()
                if a.sync_every and (i + 1) % a.sync_every == 0:
                    torch.cuda.synchronize()
            if mode in ("submit", "fused") or exch:
                plan.flush()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            plan.close()
            # host issue rate alone: the same loop without waiting for the GPU at the end is bounded by it
            res[mode + "_us"] = round(dt * 1e6, 2)
        rows.append(res)
        if H == 1024:
            full = res
    print("ranks rows   " + "  ".join(f"{m:>10s}" for m in rows[0] if m.endswith("_us")) + "   | ideal (full/N) | step/ideal | efficiency if compute-bound")
    for r in rows:
        keys = [k for k in r if k.endswith("_us")]
        best = min(r[k] for k in keys)
        ideal = min(full[k] for k in keys) / r["ranks"]
        print(f"{r['ranks']:5.2f} {r['rows']:4d}   " + "  ".join(f"{r[k]:10.2f}" for k in keys) + f"   | {ideal:10.2f}     | {best / ideal:6.2f}     | {ideal / best:5.2f}")
    if a.host_cost and full is not None:
        print("host cost per call (us):", {k: v for k, v in full.items() if k.endswith("_host")})
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
