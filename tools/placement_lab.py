#!/usr/bin/env python3
"""Placement lab: what separates a fast from a slow allocation for K1, and can an allocation
recipe avoid the slow class without a search?

    python tools/placement_lab.py map   [--sets 14] [--pitch 16]          torch allocations, one K1 time per set
    python tools/placement_lab.py alloc [--sets 18] [--pitch 8]           allocator kinds interleaved in one process
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d D -o p -- python3 tools/placement_lab.py map --log D/map.json
        -> tools/placement_lab.py join D      joins the per-dispatch counters with the per-set K1 times of THAT process

`map` launches, per set, 1 untimed + `--reps` timed K1 (deg 3, the headline launch); the log lists the sets in launch
order, so the i-th group of (1 + reps) srf_kernel dispatches of the trace belongs to set i.
"""
import argparse
import collections
import csv
import ctypes as C
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))

KINDS = {"torch": -1, "hipmalloc": 0, "contig": 1, "vmm1": 2, "vmm2m": 3, "vmm64m": 3, "vmm1g": 3, "uncached": 4}
CHUNK = {"vmm2m": 2 << 20, "vmm64m": 64 << 20, "vmm1g": 1 << 30}


class _Raw:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class Allocator:
    def __init__(self, torch, dev):
        self.torch, self.dev = torch, dev
        self.lib = C.CDLL(os.path.join(ROOT, "tools", "libplacement_alloc.so"))
        self.lib.lab_alloc.restype = C.c_void_p
        self.lib.lab_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int]
        self.lib.lab_ptr.restype = C.c_void_p
        self.lib.lab_ptr.argtypes = [C.c_void_p]
        self.lib.lab_free.argtypes = [C.c_void_p]
        self.lib.lab_last_error.restype = C.c_char_p
        self.lib.lab_granularity.restype = C.c_size_t
        self.blocks = []

    def granularity(self):
        return self.lib.lab_granularity(0, 0), self.lib.lab_granularity(0, 1)

    def empty(self, kind, shape, dtype):
        torch = self.torch
        if kind == "torch":
            return torch.empty(shape, dtype=dtype, device=self.dev)
        n = 1
        for s in shape:
            n *= s
        esz = torch.empty((), dtype=dtype).element_size()
        b = self.lib.lab_alloc(KINDS[kind], n * esz, CHUNK.get(kind, 0), 0)
        if not b:
            raise RuntimeError(f"{kind}: {self.lib.lab_last_error().decode()}")
        self.blocks.append(b)
        ts = {torch.float32: "<f4", torch.uint8: "|u1"}[dtype]
        return torch.as_tensor(_Raw(self.lib.lab_ptr(b), shape, ts), device=self.dev)


def setup():
    import torch
    from s2_emit import SpectralFusion, _engine as eng
    from s2_emit.synthetic import device_problem
    dev = torch.device("cuda:0")
    p = device_problem(1024, 1024, 285, deg=3, seed=0)
    plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0)
    return torch, eng, dev, p, plan


def k1_timer(torch, eng, plan, reps):
    npix = 1024 * 1024
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def k1(c, r, o):
        rr, rl = plan._real_image(r, npix)
        eng.srf_integrate_moments(c, plan.table, rr, 3, plan.ws, None, 0.0, 0.0, out=o, reduce=False, layout=plan.layout,
                                  real_layout=rl, opts=plan.opts)

    def t(c, r, o):
        k1(c, r, o)
        ts = []
        for _ in range(reps):
            e0.record(); k1(c, r, o); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts), ts
    return t


def cmd_map(a):
    torch, eng, dev, p, plan = setup()
    t = k1_timer(torch, eng, plan, a.reps)
    nb, npix = plan.table.nb, 1024 * 1024
    sets, spacers = [], []
    for i in range(a.sets):
        if i:
            try:
                spacers.append(torch.empty(int(a.pitch * (1 << 30)), dtype=torch.uint8, device=dev))
            except RuntimeError:
                break
        sets.append((p.cube if i == 0 else p.cube.clone(), p.real if i == 0 else p.real.clone(),
                     eng.alloc_image(torch, nb, npix, plan.layout, dev)))
    # settle the power state first (profiles/r02_ramp.log) so that set 0 is not penalised
    rr0, rl0 = plan._real_image(sets[0][1], npix)
    for _ in range(a.settle):
        eng.srf_integrate_moments(sets[0][0], plan.table, rr0, 3, plan.ws, None, 0.0, 0.0, out=sets[0][2], reduce=False,
                                  layout=plan.layout, real_layout=rl0, opts=plan.opts)
    torch.cuda.synchronize()
    rec = []
    for i, s in enumerate(sets):
        best, ts = t(*s)
        rec.append({"set": i, "ms": round(best, 4), "all_ms": [round(x, 4) for x in ts], "cube_addr": s[0].data_ptr(),
                    "out_addr": s[2].data_ptr()})
        print(f"set {i:2d}  cube {s[0].data_ptr():#x}  K1 {best:.4f} ms  ({' '.join(f'{x:.4f}' for x in ts)})", flush=True)
    log = {"reps": a.reps, "launches_per_set": a.reps + 1, "settle_launches": a.settle, "sets": rec}
    if a.log:
        json.dump(log, open(a.log, "w"))


def cmd_alloc(a):
    torch, eng, dev, p, plan = setup()
    al = Allocator(torch, dev)
    print("VMM granularity (min, recommended):", al.granularity(), flush=True)
    t = k1_timer(torch, eng, plan, a.reps)
    nb, npix = plan.table.nb, 1024 * 1024
    kinds = a.kinds.split(",")
    row = (nb + 3) // 4 * 4
    res = collections.defaultdict(list)
    keep = []
    for _ in range(300):
        plan.step(p.cube, p.real)
    torch.cuda.synchronize()
    for i in range(a.sets):
        kind = kinds[i % len(kinds)]
        try:
            if i and a.pitch > 0:
                keep.append(torch.empty(int(a.pitch * (1 << 30)), dtype=torch.uint8, device=dev))
            if a.arena:      # all three operands in ONE allocation of this kind
                cb, rb, ob = p.cube.numel() * 4, p.real.numel() * 4, npix * row * 4
                pad = lambda x: (x + (2 << 20) - 1) // (2 << 20) * (2 << 20)
                arena = al.empty(kind, (pad(cb) + pad(rb) + pad(ob),), torch.uint8)
                c = arena[:cb].view(torch.float32).view(p.cube.shape)
                r = arena[pad(cb):pad(cb) + rb].view(torch.float32).view(p.real.shape)
                o = arena[pad(cb) + pad(rb):pad(cb) + pad(rb) + ob].view(torch.float32).view(npix, row)
                keep.append(arena)
            else:
                c = al.empty(kind, tuple(p.cube.shape), torch.float32)
                r = al.empty(kind, tuple(p.real.shape), torch.float32)
                o = al.empty(kind, (npix, row), torch.float32)
            c.copy_(p.cube); r.copy_(p.real)
        except RuntimeError as e:
            print(f"set {i} {kind}: {str(e)[:120]}", flush=True)
            continue
        keep.append((c, r, o))
        best, ts = t(c, r, o)
        res[kind].append(best)
        print(f"set {i:2d} {kind:9s} cube {c.data_ptr():#x}  K1 {best:.4f} ms", flush=True)
    print("\nkind       n  fast(<0.206)  min     median  max")
    for k in kinds:
        v = res[k]
        if v:
            print(f"{k:9s} {len(v):2d}  {sum(x < 0.206 for x in v):2d}            {min(v):.4f}  {statistics.median(v):.4f}  {max(v):.4f}")
    free, total = torch.cuda.mem_get_info()
    print(f"free {free / 2**30:.1f} of {total / 2**30:.1f} GB")


def cmd_join(a):
    d = a.dir
    log = json.load(open(os.path.join(d, "map.json")))
    per = log["launches_per_set"]
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "srf_kernel" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ds = [disp[k] for k in sorted(disp)][log["settle_launches"]:]
    names = [c for c in ds[0] if c != "dur"]
    print(f"{len(ds)} K1 dispatches after the settle launches, {per} per set; counters: {names}")
    print("set  event_ms  trace_us  " + "  ".join(f"{n[:28]:>28s}" for n in names))
    rows = []
    for s in log["sets"]:
        g = ds[s["set"] * per + 1: (s["set"] + 1) * per]          # skip the untimed first touch
        if len(g) < per - 1:
            break
        avg = {n: sum(x[n] for x in g) / len(g) for n in names}
        dur = sum(x["dur"] for x in g) / len(g)
        rows.append((s["ms"], dur, avg))
        print(f"{s['set']:3d}  {s['ms']:.4f}    {dur:7.1f}   " + "  ".join(f"{avg[n]:28.1f}" for n in names))
    # correlation of each counter with the traced duration across sets
    import numpy as np
    du = np.array([r[1] for r in rows])
    print("\ncounter                          corr(dur)   fast-class mean   slow-class mean   slow/fast")
    med = (du.min() + du.max()) / 2
    for n in names:
        v = np.array([r[2][n] for r in rows])
        cc = float(np.corrcoef(du, v)[0, 1]) if v.std() > 0 and du.std() > 0 else float("nan")
        fa, sl = v[du < med], v[du >= med]
        print(f"{n:32s} {cc:+.3f}      {fa.mean() if fa.size else float('nan'):14.1f}    {sl.mean() if sl.size else float('nan'):14.1f}"
              f"    {(sl.mean() / fa.mean()) if fa.size and sl.size and fa.mean() else float('nan'):.3f}")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("map"); m.add_argument("--sets", type=int, default=14); m.add_argument("--pitch", type=float, default=16.0)
    m.add_argument("--reps", type=int, default=3); m.add_argument("--log", default=None)
    m.add_argument("--settle", type=int, default=300)
    al = sub.add_parser("alloc"); al.add_argument("--sets", type=int, default=18); al.add_argument("--pitch", type=float, default=8.0)
    al.add_argument("--reps", type=int, default=3); al.add_argument("--kinds", default="torch,hipmalloc,contig,vmm1,vmm2m,vmm1g")
    al.add_argument("--arena", action="store_true")
    j = sub.add_parser("join"); j.add_argument("dir")
    a = ap.parse_args()
    {"map": cmd_map, "alloc": cmd_alloc, "join": cmd_join}[a.cmd](a)


if __name__ == "__main__":
    main()
