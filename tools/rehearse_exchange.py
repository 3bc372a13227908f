#!/usr/bin/env python3
"""Rehearsal of the exchange steps of an N-rank run on ONE GPU: a one-rank RCCL group, the collectives
of the three sharded fits issued on a high-priority side stream WHILE the persistent K1 owns the caller's stream:

    moments all-reduce / reduce + broadcast   1 KB      (SpectralFusion.submit, every step)
    percentile histograms all-reduce          132 KB    (eng.percentile_limits(distributed=True), three per image)
    ridge statistics all-gather               168 B     (PolyRidge.fit(group=...))
    ridge Gram all-reduce                     1.4 MB    (PolyRidge.fit(group=...))
    a 16 MB all-reduce                                   (beyond anything the path sends: where the CU budget ends)

For each: the time from issue to completion on the side stream when K1 (8 CUs reserved, as in a pipelined multi-rank run) is
running, the same alone, and whether it finished before K1 did.  Run it under `rocprofv3 --kernel-trace` to see which RCCL
kernels a one-rank group launches (names, grids, queues).  A one-rank group is as far as one GPU goes: RCCL refuses two ranks
on one device, so ring / tree kernels over xGMI are not exercised here - what IS exercised is that RCCL's launches on the
high-priority stream are dispatched on the reserved CUs under K1, and the host-side call overhead of each collective.
"""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    from s2_emit import SpectralFusion, _engine as eng
    from s2_emit.synthetic import device_problem
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev, pg_options=opts)
    p = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
    plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, min_valid=0.0, min_count=50, device=dev, reserved_cus=8,
                          coeff_sync="allreduce", force_exchange=True)
    main_s = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev, priority=-1)
    ws = eng.MomentWorkspace(dev, plan.table.nb, 3)
    real2 = p.real.reshape(-1, p.real.shape[-1])
    out = eng.alloc_image(torch, plan.table.nb, 1024 * 1024, plan.layout, dev)

    def k1():
        eng.srf_integrate_moments(p.cube, plan.table, real2, 3, ws, None, 0.0, 0.0, out=out, reduce=False, layout=plan.layout,
                                  real_layout=plan.layout, opts=plan.opts)
    bufs = {"moments 1 KB": torch.zeros(12 * 11, dtype=torch.float64, device=dev),
            "histograms 132 KB": torch.zeros(33 * 1024, dtype=torch.int32, device=dev),
            "ridge Gram 1.4 MB": torch.zeros(288 * 608, dtype=torch.float64, device=dev),
            "16 MB": torch.zeros(4 << 20, dtype=torch.float32, device=dev)}
    stats = torch.zeros(21, dtype=torch.float64, device=dev)
    cases = [(name, (lambda t=t: dist.all_reduce(t))) for name, t in bufs.items()]
    cases += [("ridge stats all_gather 168 B", lambda: dist.all_gather([torch.empty_like(stats)], stats)),
              ("coeffs broadcast 384 B", lambda: dist.broadcast(bufs["moments 1 KB"][:48], src=0)),
              ("moments reduce 1 KB", lambda: dist.reduce(bufs["moments 1 KB"], dst=0))]
    for _ in range(300):
        k1()
    for _, fn in cases:
        with torch.cuda.stream(side):
            fn()
    torch.cuda.synchronize()
    rows = []
    for name, fn in cases:
        res = {"collective": name}
        for under in (True, False):
            ts = []
            for _ in range(7):
                e0, e1, k_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                torch.cuda.synchronize()
                if under:
                    k1()
                    k_end.record(main_s)
                t0 = time.perf_counter()
                with torch.cuda.stream(side):
                    e0.record(side)
                    fn()
                    e1.record(side)
                host_us = (time.perf_counter() - t0) * 1e6
                torch.cuda.synchronize()
                ts.append((e0.elapsed_time(e1) * 1e3, (e1.elapsed_time(k_end) * 1e3) if under else 0.0, host_us))
            ts.sort()
            m = ts[len(ts) // 2]
            if under:
                res["under_k1_us"], res["k1_ended_after_us"], res["host_call_us"] = round(m[0], 1), round(m[1], 1), round(m[2], 1)
            else:
                res["alone_us"] = round(m[0], 1)
        rows.append(res)
        print(json.dumps(res), flush=True)
    # the pipelined step with the exchange in the loop, both sync modes
    for mode in ("allreduce", "broadcast"):
        pl = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, min_valid=0.0, min_count=50, device=dev, coeff_sync=mode,
                            force_exchange=True)
        for _ in range(300):
            pl.submit(p.cube, p.real)
        pl.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            pl.submit(p.cube, p.real)
        pl.flush()
        torch.cuda.synchronize()
        print(json.dumps({"pipelined step with one-rank RCCL": mode, "us_per_step": round((time.perf_counter() - t0) / 200 * 1e6, 1),
                          "side_stream_candidates_us": pl.side_stream_log}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
