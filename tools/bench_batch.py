#!/usr/bin/env python3
"""Batched small tiles (the reference's own problem size: 100 x 100 x 285 EMIT tiles, one fit per tile -
tiles_helpers/utils.py:223-305, Spectral_matching.ipynb raw lines 293-294): step_batch() over T tiles in three launches
against T step() calls, in one process, interleaved rounds, HIP-event timing.

    python tools/bench_batch.py [--tiles 256] [--size 100] [--cube f32|u16] [--rounds 5]

Prints us per tile and TB/s of cube bytes for both, and the large-tile rate (one 1024 x 1024 tile) of the same box for
the "within 1.3x of the large-tile rate" target."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=256)
    ap.add_argument("--size", type=int, default=100)
    ap.add_argument("--cube", default="f32", choices=["f32", "u16"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-placement", action="store_true", help="leave the stacked inputs where the allocator put them")
    ap.add_argument("--no-loop", action="store_true", help="skip the T x step() arm (slow for large T)")
    a = ap.parse_args()
    import torch
    from s2_emit import SpectralFusion, _engine as eng
    from s2_emit.synthetic import device_problem
    torch.cuda.set_device(0)
    T, S, B = a.tiles, a.size, 285
    # one big synthetic scene cut into T tiles (cheaper than T generator calls; tiles stay independent fits)
    rows = S * T
    prob = device_problem(rows, S, B, deg=3, seed=0)
    cube = prob.cube if a.cube == "f32" else eng.tile_encode_u16(prob.cube)
    cubes = cube.reshape(T, S, S, B)
    reals = prob.real.reshape(T, S, S, -1)
    plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50)
    big = device_problem(1024, 1024, B, deg=3, seed=1)
    bcube = big.cube if a.cube == "f32" else eng.tile_encode_u16(big.cube)
    esz = 4 if a.cube == "f32" else 2

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def loop():
        for i in range(T):
            plan.step(cubes[i], reals[i])

    if not a.no_placement:
        cubes, reals, _ = plan.place_batch_inputs(cubes, reals)      # resident batch: inputs and outputs in a fast stretch
    k1 = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    res = {"batch": [], "loop": [], "large": [], "batch_k1": []}
    for _ in range(a.rounds):
        res["batch"].append(timed(lambda: plan.step_batch(cubes, reals), a.reps))
        plan.step_batch(cubes, reals, k1_events=k1)
        torch.cuda.synchronize()
        res["batch_k1"].append(k1[0].elapsed_time(k1[1]))
        if not a.no_loop:
            res["loop"].append(timed(loop, 1))
        res["large"].append(timed(lambda: plan.step(bcube, big.real), a.reps))
    tile_bytes = S * S * B * esz
    out = {"tiles": T, "tile": f"{S}x{S}x{B}", "cube": a.cube, "tile_MB": round(tile_bytes / 1e6, 2)}
    for k, v in res.items():
        if not v:
            continue
        ms = sorted(v)[len(v) // 2]
        byt = 1024 * 1024 * B * esz if k == "large" else T * tile_bytes
        out[k] = {"ms_median": round(ms, 4), "ms_min": round(min(v), 4), "TBps_cube_bytes": round(byt / ms / 1e9, 3)}
        if k != "large":
            out[k]["us_per_tile"] = round(ms * 1e3 / T, 3)
    out["placement_trials_ms"] = {"batch": getattr(next(iter(plan._batches.values())), "placement_log", None), "large": plan.placement_log.get(1024 * 1024)}
    out["batch_vs_large_rate"] = round(out["large"]["TBps_cube_bytes"] / out["batch"]["TBps_cube_bytes"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
