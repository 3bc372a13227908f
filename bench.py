#!/usr/bin/env python3
"""Headline benchmark: fused EMIT->S2 spectral matching throughput in Mpixel*bands/s.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W   (the wrapped form works too)

A step = one pass of the hot path over one batch of synthetic input: per GPU one 1024x1024x285
EMIT-like cube + matching real-S2 planes (BASELINE.json configs[2]; weak scaling: one tile per GPU,
configs[3]/[4]) -> K1+K2 (SRF integration + Vandermonde moments, one pass over the cube) ->
C1 (RCCL exchange, N>1) -> polynomial solve (degree 3, per band, all valid pixels) -> K3 (apply).
Inputs are resident in HBM before the timed region.  value = N*H*W*285*K / t / 1e6.
`--scaling strong` splits ONE HxW cube into N row blocks (H/N rows per rank, one global fit): value = H*W*285*K / t / 1e6.

The JSON line also carries:
  roofline     dominant kernel (K1+K2 fused): algorithmic bytes = H*W*285*4 per launch (the cube read
               exactly once) / its average duration measured with HIP events on the
               launch stream inside the timed region, against the 8 TB/s HBM3E peak.
  cold         the same step on the caller's first allocations, with no placement trials and no settle phase
               (what a caller who just allocates and runs gets), measured in the same process before the search.
  cpu_baseline the oracle (NumPy restatement in the reference's operation order) timed on this box's
               host cores on a bounded row-slab of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# dmabuf IPC only on this pool (RCCL / cross-process sharing fail with the legacy mode); must be in the
# environment before the HSA runtime initialises, i.e. before the first torch.cuda call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join("profiles", "traffic.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--bands", type=int, default=285)
    ap.add_argument("--deg", type=int, default=3)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: one HxW tile per GPU (per-GPU work fixed).  strong: ONE HxW cube split into N row blocks of "
                         "H/N rows, one per rank, and one global fit over all of them (total work fixed; SURVEY.md 8e "
                         "'row-blocks H/G')")
    ap.add_argument("--coeff-sync", default="allreduce", choices=["local", "allreduce", "broadcast"])
    ap.add_argument("--cpu-rows", type=int, default=256,
                    help="rows of the cube the single-thread CPU baseline processes (the all-cores run takes the whole cube)")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="processes of the row-sharded all-cores CPU run (0 = min(16, os.cpu_count()): a 1-GPU box's CPU share)")
    ap.add_argument("--tiles-per-gpu", type=int, default=1,
                    help="tiles resident per GPU; > 1 runs the mosaic step (K1+K2 per tile, ONE fit over all tiles of all "
                         "ranks, K3 per tile): BASELINE configs[3]/[4] (4 / 8 tiles) on however many GPUs there are")
    ap.add_argument("--k1-launches", type=int, default=24,
                    help="extra event-bracketed launches of the K1+K2 kernel after the timed region, so that roofline.frac "
                         "rests on >= 20 launches whatever --steps is")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl (= RCCL over xGMI; the measured configuration) or gloo "
                         "(CPU-side rehearsal of the multi-rank control flow)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off", "fused"],
                    help="software pipeline: on = one tile deep (fit of tile i on a side stream under K1 of tile i+1, K3(i) behind "
                         "K1(i+1)); fused = K3 of tile i-2 as a pre-phase of K1's launch for tile i (one kernel per tile on the "
                         "caller's stream; fit of tile i-1 as tail work of the same launch; needs no exchange); auto = on for N > 1, "
                         "fused at N = 1 if its self-check against step() passes, else off")
    ap.add_argument("--reserve-cus", type=int, default=8,
                    help="CUs left free of persistent K1 workgroups in pipelined mode (side-stream tail of the previous tile)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearsal only (N = 1): create a one-rank RCCL group and run the multi-GPU code path - "
                         "pipelined submit() with the collective on the side stream - to measure its launch cost on one GPU")
    ap.add_argument("--cube", default="f32", choices=["f32", "u16"],
                    help="f32: the headline workload (float32 cube, 4 B per pixel*band).  u16: the same cube in the "
                         "reference's on-disk tile format (uint16 x 10000, tiles_helpers/utils.py:362-374), decoded inside "
                         "K1 - a SURVEY 8-f2 measurement, 2 B per pixel*band, not the headline")
    ap.add_argument("--u16-fast", action="store_true",
                    help="with --cube u16: the opt-in fast arithmetic of the uint16 kernel (HSR_SRF_U16_FAST; 1e-6 relative "
                         "off the bit-exact path)")
    ap.add_argument("--fused-fit", action="store_true",
                    help="slot reduction + solve inside K1's launch (hsr_srf_integrate_fit); measured 3 us/step slower, off by default")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed load before the warm-up, to reach the GPU's settled power state: settle_ms / 0.25 steps per tile (0: none)")
    ap.add_argument("--no-input-placement", action="store_true",
                    help="keep the synthetic cube / target where the allocator first put them (no placement trials for inputs)")
    ap.add_argument("--placement-trials", type=int, default=96,
                    help="candidate allocations the placement search may time (0: none; the search is opt-in in the library).  r04: 96 "
                         "candidates BACK TO BACK (--placement-pitch-gb 0) - a dense map of ~140 GB of device memory, whose fast stretches "
                         "are 15-25 GB long - instead of 12 candidates 16 GB apart: the sparse search landed in a slow stretch in one of "
                         "two runs on the same box (0.2347 / 0.2136 ms per step against 0.2125 / 0.2142 dense)")
    ap.add_argument("--placement-pitch-gb", type=float, default=0.0,
                    help="spacer between consecutive candidate allocations of the placement search (0 = candidates back to back: a dense "
                         "map; the library's own default for sparse searches is 16 GB)")
    ap.add_argument("--placement-budget-gb", type=float, default=200.0,
                    help="device memory the placement search may hold (spacers + candidates); the library's own default is half "
                         "of the free memory")
    ap.add_argument("--cold-steps", type=int, default=20,
                    help="steps of the 'cold' region (first allocations, no placement trials, no settle), run before everything "
                         "else and reported as line['cold'] (0: skip)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket the K1+K2 kernel with HIP events on every n-th timed step (each pair of event "
                         "records costs ~12 us of launch gap, so not on every step)")
    ap.add_argument("--fake-collective-us", type=float, default=0.0,
                    help="rehearsal only (with --force-exchange): a one-rank RCCL all-reduce launches no kernel, so a stand-in kernel "
                         "(--fake-collective-blocks workgroups of 256 threads and 48 KB of LDS) stays resident on the side stream for "
                         "this many microseconds where the collective of an N-rank run would run")
    ap.add_argument("--fake-collective-blocks", type=int, default=2)
    ap.add_argument("--launcher", action="store_true",
                    help="take the parent -> torchrun -> rank path also at N = 1 (with --force-exchange the child then creates its "
                         "one-rank RCCL group exactly as the ranks of an N > 1 run do): rehearses the launcher on a one-GPU box")
    ap.add_argument("--dist-timeout", type=float, default=180.0,
                    help="seconds a rank waits in init_process_group or in any collective before the run is aborted (N > 1)")
    ap.add_argument("--deadline", type=float, default=900.0,
                    help="N > 1: hard limit per rank in seconds - a rank still alive then dumps its stacks and exits non-zero "
                         "(0: none).  The self-launching parent gives the whole job this long plus a minute.")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks from here
# ----------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count() -> int:
    """GPUs this node shows, counted WITHOUT loading HIP / HSA in this process: the self-launching parent goes on to start
    torchrun (fork + exec), which a process that has initialised the GPU must not do on this pool - and
    torch.cuda.device_count() only stays clear of hipGetDeviceCount while its amdsmi discovery works.  KFD's topology lists
    every node; GPU nodes are the ones with SIMDs.  HIP_/ROCR_/CUDA_VISIBLE_DEVICES narrow the set as they do for the runtime.
    Returns 0 without a KFD driver, -1 if the topology is there but cannot be read (the ranks' own check then decides)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):            # no amdgpu / KFD driver on this machine at all
        return 0
    try:
        nodes = os.listdir(base)
    except OSError:
        return -1
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split(None, 1) for ln in f.read().splitlines() if " " in ln)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass                                       # a GPU this container was not given reads as EPERM (device cgroup): not ours
    try:                                               # a container is usually handed only its own render nodes
        rn = len([d for d in os.listdir("/dev/dri") if d.startswith("renderD")])
        if rn > 0:
            n = min(n, rn)
    except OSError:
        pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process - which never touches the GPU and
    replaces nothing by exec - starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a
    child, relays rank 0's single JSON line to its own stdout and returns the child's exit status (non-zero if any rank
    failed, if the job outlived its deadline, or if no line was produced)."""
    import subprocess
    if not args.same_device:
        ndev = visible_gpu_count()                     # sysfs only: this process never loads HIP
        if 0 <= ndev < args.gpus:
            sys.stderr.write(f"[bench] --gpus {args.gpus} needs {args.gpus} visible GPUs on this node, {ndev} found "
                             f"(rehearse the control flow of more ranks than GPUs with --backend gloo --same-device)\n")
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + [a for a in argv if a != "--launcher"]
    limit = args.deadline + 60.0 if args.deadline > 0 else None
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)             # the launcher and every rank: exactly the group started above
        proc.wait()
        sys.stderr.write(f"[bench] {args.gpus}-rank job still running after {limit:.0f} s: killed\n")
        return 3
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if proc.returncode != 0:
        sys.stderr.write(f"[bench] {args.gpus}-rank job failed (exit status {proc.returncode})\n")
        return proc.returncode if 0 < proc.returncode < 256 else 1
    if not lines:
        sys.stderr.write("[bench] the ranks exited cleanly but rank 0 printed no JSON line\n")
        return 4
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    return 0


def _claim_stdout() -> int:
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner on stdout (through C stdio, at a moment
    of its own choosing - communicator creation is lazy), torch may warn, libraries may chat: so file descriptor 1
    is pointed at stderr for the whole run and the JSON line is written to the saved descriptor at the end."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle on host cores)
# ----------------------------------------------------------------------------------------------------------
def _cpu_slab(shared, job):
    """Worker of the all-cores CPU run: SRF integration (the reference's 13 float64 passes) of rows [r0, r1)."""
    import numpy as np
    from oracle import oracle_np as onp
    shape, r0, r1 = job
    R = np.frombuffer(shared, dtype=np.float32, count=shape[0] * shape[1] * shape[2]).reshape(shape)[r0:r1]
    srf = onp.synthetic_srf()
    w, good = onp.synthetic_wavelengths(shape[2])
    ps = onp.pseudo_s2_srf_integral(R, w, srf, good)
    return r0, np.stack([v for v in ps.values() if v is not None]).astype(np.float32)      # poly_regression.py:104


def _cpu_worker(shared, jobs, results):
    while True:
        job = jobs.get()
        if job is None:
            return
        try:
            results.put((True, _cpu_slab(shared, job)))
        except BaseException as e:          # report, do not die silently: the parent waits for one result per job
            results.put((False, repr(e)))


class CpuPool:
    """n worker processes forked at construction - BEFORE anything touches the GPU (a forked child of a process with a
    live HIP runtime is not something to rely on; an executor that forks on demand would do exactly that later) - which
    sleep on a queue until cpu_baseline() hands them row slabs.  The cube reaches them through an anonymous shared
    mapping created before the fork and inherited (no named shared-memory segment, hence nothing for a resource tracker
    to complain about at exit)."""

    def __init__(self, n: int, nbytes: int):
        import mmap
        import multiprocessing as mp
        ctx = mp.get_context("fork")
        self.shared = mmap.mmap(-1, max(int(nbytes), mmap.PAGESIZE))
        self.jobs, self.results = ctx.SimpleQueue(), ctx.SimpleQueue()
        self.procs = [ctx.Process(target=_cpu_worker, args=(self.shared, self.jobs, self.results), daemon=True)
                      for _ in range(n)]
        for p in self.procs:
            p.start()

    def map(self, jobs):
        jobs = list(jobs)
        for j in jobs:
            self.jobs.put(j)
        out = []
        for _ in jobs:
            ok, res = self.results.get()
            if not ok:
                raise RuntimeError(f"CPU baseline worker failed: {res}")
            out.append(res)
        return out

    def shutdown(self):
        for _ in self.procs:
            self.jobs.put(None)
        for p in self.procs:
            p.join(timeout=10)
        self.shared.close()


def start_cpu_pool(args):
    n = args.cpu_workers if args.cpu_workers > 0 else min(16, os.cpu_count() or 1)
    pool = CpuPool(n, args.height * args.width * args.bands * 4)
    alive = sum(p.is_alive() for p in pool.procs)
    if alive != n:
        raise RuntimeError(f"only {alive} of {n} CPU workers started")
    return pool, n, alive


def cpu_baseline(args, pool, nworkers, cube_host, real_host, gpu_pseudo, gpu_matched):
    """The oracle ('port' of the reference's NumPy path: 13 full-cube float64 SRF passes, np.polyfit, np.polyval) on the
    SAME cube the GPU processed:
      * single thread (NumPy elementwise is single-threaded: the reference's actual behaviour) on the first --cpu-rows rows;
      * row-sharded over the host cores on the whole cube, SURVEY.md 8(d) - its outputs are the
        full-size parity reference for the GPU's pseudo / matched images (max_rel_err)."""
    import numpy as np
    from oracle import oracle_np as onp
    H, W, B = cube_host.shape
    srf = onp.synthetic_srf()
    w, good = onp.synthetic_wavelengths(B)
    rows = min(args.cpu_rows, H)
    t0 = time.perf_counter()
    onp.fuse_lsq_reference(cube_host[:rows], w, srf, good, real_host[:, :rows], args.deg)
    dt1 = time.perf_counter() - t0
    out = {"value": round(rows * W * B / dt1 / 1e6, 2), "unit": "Mpixel*bands/s", "cores": 1, "kind": "port",
           "host_cores": os.cpu_count(),
           "sample": f"first {rows} rows of the {H}x{W}x{B} cube the GPU processed: SRF (13 float64 passes) + deg-{args.deg} "
                     f"np.polyfit per band + np.polyval apply, single-thread NumPy, {dt1:.1f} s"}
    # all cores, whole cube
    if cube_host.nbytes > len(pool.shared):
        raise RuntimeError("CPU pool was sized for a smaller cube")
    np.frombuffer(pool.shared, dtype=np.float32, count=cube_host.size).reshape(cube_host.shape)[...] = cube_host
    slab = max(8, min(64, H // max(1, 2 * nworkers)))
    jobs = [(tuple(cube_host.shape), r0, min(H, r0 + slab)) for r0 in range(0, H, slab)]
    t0 = time.perf_counter()
    parts = dict(pool.map(jobs))
    pseudo = np.concatenate([parts[r0] for r0 in sorted(parts)], axis=1)               # (nb, H, W) float32
    valid = np.ones(pseudo.shape[1:], dtype=bool)
    coeffs, _ = onp.fit_per_band_poly(pseudo, real_host, valid, args.deg, 0.0, 50)
    matched = onp.apply_poly_planes(pseudo, coeffs, None, clip=True)
    dtn = time.perf_counter() - t0
    out["all_cores"] = {"value": round(H * W * B / dtn / 1e6, 1), "cores": nworkers, "seconds": round(dtn, 2),
                        "sample": f"whole {H}x{W}x{B} cube, SRF row-sharded over {nworkers} processes ({slab}-row slabs), "
                                  f"polyfit + polyval in the parent"}

    def rel(got, ref):          # the tests' measure: |got - ref| / max(|ref|, 1e-3 max|ref|)
        ref = ref.astype(np.float64)
        scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max() + 1e-30)
        return float(np.max(np.abs(got.astype(np.float64) - ref) / scale))
    err = {"pseudo": rel(gpu_pseudo, pseudo), "matched": rel(gpu_matched, matched),
           "pixels_checked": int(H * W), "reference": "oracle (all-cores run above), full size"}
    return out, err


# ----------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if (args.gpus > 1 or args.launcher) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    real_stdout = _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.deadline > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.deadline, exit=True)     # a wedged rank ends non-zero, with its stacks on stderr
    strong = args.scaling == "strong"
    want_cpu = (world == 1 and rank == 0 and not args.no_cpu_baseline and args.tiles_per_gpu == 1 and args.cube == "f32")
    pool = nworkers = None
    if want_cpu:
        pool, nworkers, _ = start_cpu_pool(args)          # forked before the GPU is initialised
    import datetime
    import torch
    import torch.distributed as dist
    from s2_emit import SpectralFusion
    from s2_emit import _engine as eng
    from s2_emit.synthetic import device_problem

    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.same_device:
        if args.backend != "gloo":
            raise SystemExit("--same-device is a rehearsal mode and needs --backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    ndev = torch.cuda.device_count()           # does not initialise the GPU
    if local_rank >= ndev:
        raise SystemExit(f"[bench] rank {rank}: local rank {local_rank} needs cuda:{local_rank}, but only {ndev} device(s) are visible. "
                         f"--gpus {args.gpus} needs {args.gpus} visible GPUs on this node, {ndev} found (rehearse the control flow of more "
                         f"ranks than GPUs with --backend gloo --same-device).")
    if strong and (args.height % world or args.tiles_per_gpu != 1):
        raise SystemExit(f"--scaling strong needs --height ({args.height}) divisible by the number of ranks ({world}) and one tile per GPU")
    if strong and world > 1 and args.coeff_sync == "local":
        raise SystemExit("--scaling strong fits ONE polynomial over all row blocks: --coeff-sync must not be 'local'")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.force_exchange and world != 1:
        raise SystemExit("--force-exchange is a one-process rehearsal")
    exchange_ranks = None
    dist_timeout = datetime.timedelta(seconds=args.dist_timeout)
    if world > 1 or args.force_exchange:
        def rccl_options():
            # RCCL's internal stream must not share a hardware queue with the stream K1 runs on (streams of the default
            # priority did, on this stack): ask for the high-priority stream
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            return opts
        if args.force_exchange and "MASTER_PORT" in os.environ and "WORLD_SIZE" in os.environ:
            # started by the launcher (--launcher): the rendezvous the ranks of an N > 1 run use
            dist.init_process_group("nccl", device_id=device, pg_options=rccl_options(), timeout=dist_timeout)
        elif args.force_exchange:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=device,
                                    pg_options=rccl_options(), timeout=dist_timeout)
        elif args.backend == "nccl":
            # No fallback: a backend chosen per rank after a partial RCCL failure would leave some ranks inside an RCCL
            # collective and others in gloo (a hang), and a host-staged gloo number must not pass for an xGMI one.
            dist.init_process_group("nccl", device_id=device, pg_options=rccl_options(), timeout=dist_timeout)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo", timeout=dist_timeout)
        # the communicator really works, on every rank: an all-reduce of ones must count the ranks (reported as rccl_ranks)
        probe = torch.ones(1, device=device if args.backend == "nccl" or args.force_exchange else "cpu")
        dist.all_reduce(probe)
        if probe.is_cuda:
            torch.cuda.synchronize()
        exchange_ranks = int(probe.item())
        if exchange_ranks != world:
            raise SystemExit(f"[bench] all-reduce of ones gave {exchange_ranks} on {world} ranks")

    Hfull, W, B = args.height, args.width, args.bands
    H = Hfull // world if strong else Hfull            # rows this rank holds
    ntl = max(1, args.tiles_per_gpu)
    probs = [device_problem(H, W, B, deg=args.deg, seed=rank * ntl + i, device=device) for i in range(ntl)]
    prob = probs[0]
    exchanging = world > 1 or args.force_exchange
    # the library's own RCCL communicator (include/hsr.h hsr_comm_*): ONE for every plan of this run; the step executor issues the
    # per-step collective on it from C.  Under gloo (rehearsal) there is none and the exchange pipeline uses its host transport.
    comm, comm_error = None, None
    if exchanging and (args.backend == "nccl" or args.force_exchange) and args.coeff_sync != "local":
        try:
            comm = eng.Comm(None, device)
        except Exception as e:              # RCCL's C API not bindable / the communicator did not come up: round 3's path (two slots, torch.distributed)
            comm_error = f"{type(e).__name__}: {e}"
            sys.stderr.write(f"[bench] rank {rank}: hsr_comm not available ({comm_error}); the exchange goes through torch.distributed\n")
        if world > 1:                       # one decision for all ranks
            okc = torch.tensor([1.0 if comm is not None else 0.0], device=device)
            dist.all_reduce(okc, op=dist.ReduceOp.MIN)
            if okc.item() < 0.5 and comm is not None:
                comm.close()
                comm, comm_error = None, "another rank could not create its communicator"
    exchanging = exchanging and args.coeff_sync != "local"
    want_fused = ntl == 1 and not args.fused_fit and args.pipeline in ("fused", "auto")
    if exchanging and comm is None and (args.backend == "nccl" or args.force_exchange) and args.pipeline == "auto":
        want_fused = False                  # no communicator of our own: two-slot pipeline, collective from Python
    if exchanging and args.same_device and args.pipeline == "auto":
        # ranks SHARING a GPU (control-flow rehearsal): a K1 launch of the exchange pipeline that polls for its coefficients holds
        # the whole chip, the other rank's K1 - whose tail has to publish the moments everybody waits for - gets no CU, and only the
        # polls' 20 s limit ends it (seen in round 4).  One GPU per rank is the pipeline's premise; the rehearsal takes two slots.
        want_fused = False
    if exchanging and want_fused and args.reserve_cus < 8:
        raise SystemExit("[bench] the exchange pipeline needs --reserve-cus >= 8 (one free CU per XCD for the collective and the solve)")
    reserve = args.reserve_cus if exchanging else 0      # CUs kept free for the side stream's kernels (gate, collective, solve)
    fused_note = None
    if want_fused:
        # self-check on THIS tile before anything is timed: seven tiles through the fused pipeline must come out with the bits of
        # step() (pseudo, matched, moments, coefficients) - with an exchange: step()'s collective goes through torch.distributed,
        # the pipeline's through the C ABI.  auto falls back (to the plain sequence / the two-slot pipeline) if they do not.
        chk = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=args.deg, min_valid=0.0, min_count=50, clip=True,
                             device=device, coeff_sync=args.coeff_sync if exchanging else "local", fuse_apply=True,
                             force_exchange=args.force_exchange, reserved_cus=reserve, comm=comm, u16_fast=args.u16_fast)
        chk_cube = prob.cube if args.cube == "f32" else eng.tile_encode_u16(prob.cube)
        ref_out = chk.step(chk_cube, prob.real, reuse_buffers=False)
        ref_out = [t.clone() for t in (ref_out.pseudo, ref_out.matched, ref_out.moments, ref_out.coeffs)]
        got = [chk.submit(chk_cube, prob.real) for _ in range(7)]
        got = [o for o in got if o is not None] + chk.drain()
        same = chk._pipe["fused"] and len(got) == 7 and chk.pipeline_status() == 0 and all(
            torch.equal(o.pseudo.view(torch.int32), ref_out[0].view(torch.int32)) and
            torch.equal(o.matched.view(torch.int32), ref_out[1].view(torch.int32)) and
            (torch.equal(o.moments.view(torch.int64), ref_out[2].view(torch.int64)) or (args.coeff_sync == "broadcast" and rank != 0)) and
            torch.equal(o.coeffs.view(torch.int64), ref_out[3].view(torch.int64)) for o in got)
        if world > 1:                                  # one decision for all ranks
            flag = torch.tensor([1.0 if same else 0.0], device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            same = bool(flag.item() > 0.5)
        why = chk.fused_fallback
        chk.close()
        del chk, got, ref_out, chk_cube
        if not same:
            if args.pipeline == "fused":
                raise SystemExit(f"[bench] --pipeline fused: the fused pipeline did not reproduce step() on this tile ({why})")
            want_fused = False
            fused_note = f"auto: the fused pipeline's self-check failed ({why}), {'two-slot pipeline' if exchanging else 'plain sequence'} used"
        else:
            fused_note = "self-checked before timing: seven tiles through the fused pipeline carry the bits of step()"
    pipelined = ntl == 1 and (args.pipeline == "on" or want_fused or (args.pipeline == "auto" and exchanging))
    fused = pipelined and want_fused
    # A resident mosaic (--tiles-per-gpu T > 1) on ONE rank without an exchange goes through the fused per-tile launch with ONE fit per
    # step (SpectralFusion(group_tiles=T), hsr_pipeline_create_group): T kernels per step on the caller's stream and nothing else;
    # its coefficients are checked against fuse_mosaic() on the same tiles before anything is timed.  With an exchange (N > 1) the
    # mosaic keeps the batched five-launch form.
    mosaic_fused = ntl > 1 and not exchanging and not args.fused_fit and args.pipeline in ("auto", "fused") and ntl <= 64
    mosaic_note = None

    def make_plan(trials):
        return SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=args.deg, min_valid=0.0, min_count=50,
                              clip=True, device=device, group=None,
                              coeff_sync=args.coeff_sync if exchanging else "local",
                              force_exchange=args.force_exchange,
                              # CUs kept free for the side stream: with an exchange (gate, collective, solve) and for the two-slot
                              # pipeline's fit; the fused pipeline without an exchange has no side-stream work
                              reserved_cus=(args.reserve_cus if (exchanging or not fused) else 0) if pipelined else 0,
                              u16_fast=args.u16_fast, fused_fit=args.fused_fit, placement_trials=trials,
                              placement_budget_gb=args.placement_budget_gb, fuse_apply=fused or mosaic_fused, comm=comm,
                              placement_pitch_gb=args.placement_pitch_gb,
                              group_tiles=ntl if mosaic_fused else 1,
                              rehearsal_collective=(args.fake_collective_us, args.fake_collective_blocks)
                              if (args.force_exchange and args.fake_collective_us > 0) else None)
    real = prob.real            # (H, W, row) band-last, like the cube and the reference's (H, W, C) images
    cube = prob.cube
    if args.cube == "u16":      # quantise once, outside the timed region (the writer's arithmetic, on the device)
        for pr in probs:
            pr.cube_u16 = eng.tile_encode_u16(pr.cube)
            pr.cube = None
        cube = prob.cube_u16

    def barrier():
        if world > 1:
            # everything this rank has enqueued - the side stream's collectives on the library's own communicator included - is done
            # BEFORE torch's communicator starts its barrier kernel: two RCCL communicators never have kernels in flight together
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    def runner(plan_, cube_, real_, tiles_):
        def run_step(k1_events=None):
            if mosaic_fused:                   # one step = T submits: one kernel per tile, one fit per step (the events bracket tile 0's launch)
                o = None
                for ti, (c_, r_) in enumerate(tiles_):
                    o = plan_.submit(c_, r_, k1_events=k1_events if ti == 0 else None)
                return o
            if ntl > 1:
                return plan_.fuse_mosaic(tiles_, k1_events=k1_events, resident=True)
            if pipelined:
                return plan_.submit(cube_, real_, k1_events=k1_events)
            return plan_.step(cube_, real_, k1_events=k1_events)
        return run_step

    host_issue = [0.0]

    def timed_region(plan_, run_step, steps, warm, settle_steps):
        """settle (untimed load) -> W warm-up steps -> barrier -> exactly `steps` timed steps -> barrier."""
        for i in range(settle_steps):
            run_step()
            if (i + 1) % 50 == 0:
                torch.cuda.synchronize()     # keep the launch queue short
        for _ in range(max(warm, 1)):        # always one untimed pass: code-object load and LDS attributes are setup, not a step
            run_step()
        if pipelined or mosaic_fused:
            plan_.flush()
        barrier()
        every = max(1, args.event_every)
        ev = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for i in range(0, steps, every)}
        t0 = time.perf_counter()
        for i in range(steps):
            run_step(ev.get(i))
        host_issue[0] = (time.perf_counter() - t0) / max(1, steps)      # host time to ISSUE a step (the GPU runs behind)
        if pipelined or mosaic_fused:
            plan_.flush()            # the last tile's apply belongs to the timed region
        barrier()
        return time.perf_counter() - t0, ev

    def max_over_ranks(x):
        tt = torch.tensor([x], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # A generation-2 pass of Python's cycle collector takes ~40 ms with torch imported - several times the whole
    # timed region - and stalls the launch thread (seen in rocprof traces as a 37-50 ms idle gap in front of one
    # kernel).  As timeit does: collect now, keep the collector off while timing.  (Before the warm-up, not between
    # warm-up and timed region: 40 ms of idle GPU there put the timed steps into the transient described below.)
    import gc
    gc.collect()
    gc.disable()

    # ---- cold region: what a caller gets who allocates, builds a plan and runs (no placement trials, no settle) ----
    cold = None
    if args.cold_steps > 0 and ntl == 1:
        cold_plan = make_plan(0)
        dtc, evc = timed_region(cold_plan, runner(cold_plan, cube, real, None), args.cold_steps, 1, 0)
        dtc = max_over_ranks(dtc)
        kc = [a.elapsed_time(b) for a, b in evc.values()]
        cold = {"ms_per_step": round(dtc / args.cold_steps * 1e3, 4), "kernel_ms": round(sum(kc) / max(1, len(kc)), 4),
                "steps": args.cold_steps,
                "note": "same step, same process, BEFORE the placement trials: inputs and images where the allocator first put "
                        "them, one untimed step (code-object load), no settle phase - the speed a caller's own tensors get"}
        del cold_plan, evc

    plan = make_plan(0 if args.same_device else args.placement_trials)
    input_log = None
    if ntl == 1 and not args.no_input_placement and not args.same_device and args.placement_trials > 1:
        # where the resident inputs lie in HBM is the benchmark's to choose: the same slow stretches of device memory that
        # the plan avoids for its own images (profiles/r02_two_speeds.md) slow K1's read streams too, so the cube and the
        # target are cloned into a few regions before the warm-up and the fastest copies kept (same bytes, same results)
        t_place = time.perf_counter()
        cube, real, input_log = plan.place_inputs(cube, real)
        input_log["seconds"] = round(time.perf_counter() - t_place, 2)
        if args.cube == "u16":
            prob.cube_u16 = cube
        else:
            prob.cube = cube
        prob.real = real
    tiles = [((pr.cube_u16 if args.cube == "u16" else pr.cube), pr.real) for pr in probs]
    if mosaic_fused:
        # self-check: two steps through the group pipeline against fuse_mosaic() (per-tile launches, per-tile moments added in the fixed
        # order) on the same tiles - coefficients and moments bit for bit, the last tile's matched image too
        chk = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=args.deg, min_valid=0.0, min_count=50, clip=True, device=device,
                             coeff_sync="local", fuse_apply=True, group_tiles=ntl, u16_fast=args.u16_fast)
        co_ref, tot_ref, outs_ref = chk.fuse_mosaic(tiles)
        got = []
        for _ in range(2):
            got += [o for o in (chk.submit(c_, r_) for c_, r_ in tiles) if o is not None]
        got += chk.drain()
        same = len(got) == 2 * ntl and chk._pipe["group"] is not None and all(
            torch.equal(o.coeffs.view(torch.int64), co_ref.view(torch.int64)) and torch.equal(o.moments.view(torch.int64), tot_ref.view(torch.int64))
            for o in got) and torch.equal(got[-1].matched.view(torch.int32), outs_ref[-1].matched.view(torch.int32))
        chk.close()
        del chk, got, outs_ref
        if not same:
            if args.pipeline == "fused":
                raise SystemExit("[bench] the group pipeline did not reproduce fuse_mosaic() on these tiles")
            mosaic_fused = False
            mosaic_note = "auto: the group pipeline's self-check against fuse_mosaic() failed, batched five-launch form used"
            plan = make_plan(0)
        else:
            mosaic_note = "self-checked before timing: two steps through the group pipeline carry the bits of fuse_mosaic()"
    mosaic_log = None
    if mosaic_fused and not args.no_input_placement and not args.same_device and args.placement_trials > 1:
        t_place = time.perf_counter()
        tiles, mosaic_log = plan.place_mosaic(tiles)
        mosaic_log["seconds"] = round(time.perf_counter() - t_place, 2)
    run_step = runner(plan, cube, real, tiles)

    # Steady state before timing (tools/dbg/ramp.py, profiles/r02_ramp.log): started from a GPU that idled for >= 10 ms,
    # the first ~10 steps run at full speed, the next ~100 run 6-10 % slower (0.243-0.254 against 0.221 ms) and only
    # then the step time settles - a power-management transient, longer than a 20-step timed region.  A pipeline that
    # processes tiles continuously lives in the settled state, so the bench loads the GPU with untimed steps for
    # --settle-ms before the W warm-up steps; the timed region follows the warm-up with no host work in between.
    # The number of settle steps is fixed by the arguments, NOT by a clock: with more than one rank every step holds a
    # collective, and ranks that looped "until 250 ms have passed" would issue different numbers of them.
    per_step_ms = 0.25 * ntl * H / Hfull
    settle_steps = int(args.settle_ms / max(per_step_ms, 0.02) + 0.5) if args.settle_ms > 0 else 0
    dt, ev = timed_region(plan, run_step, args.steps, args.warmup, settle_steps)
    gc.enable()

    # the same K1+K2 launch, event-bracketed every time, outside the timed region: >= 20 launches for roofline.frac
    ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(0, args.k1_launches))]
    for pair in ev2:
        out = run_step(pair)
    if pipelined or mosaic_fused:
        out = plan.flush()
    else:
        out = run_step()
    barrier()
    in_region = [a.elapsed_time(b) for a, b in ev.values()]
    extra = [a.elapsed_time(b) for a, b in ev2]
    k1_all = in_region + extra
    k1_ms = sum(k1_all) / max(1, len(k1_all))
    dt_max = max_over_ranks(dt)

    if rank == 0:
        nb = len(prob.names)
        npb = H * W * B                              # pixel*bands one rank processes per step and tile
        value = world * ntl * npb * args.steps / dt_max / 1e6          # strong: world * (Hfull / world) rows = the one cube
        esz = 4 if args.cube == "f32" else 2
        cube_bytes = npb * esz
        full_bytes = H * W * (esz * B + 16 * nb)        # + pseudo write, real read, apply read + write (no mask in this run)
        # the algorithmic bytes of the dominant kernel are the cube read ONCE (H*W*285*4 per tile) - roofline.achieved
        # and .frac use exactly that over the kernel's time.  What the launch moves beyond it (targets, planes, and in the fused
        # pipelines K3 of an older tile: + 8 x row bytes per pixel) is reported beside it as frac_launch_bytes.
        alg_bytes = cube_bytes * (1 if mosaic_fused else ntl)      # the batched mosaic form runs K1+K2 of all its tiles in one launch
        launch_bytes = alg_bytes + (H * W * 8 * prob.real.shape[-1] if (fused or mosaic_fused) else 0)
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
        carried = ("K1+K2 of tile i + K3 of tile i-3 (pre-phase) + slot reduction of tile i-1 (tail) in one launch, exchange on the side stream"
                   if (fused and exchanging) else "K1+K2 of tile i + K3 of tile i-2 (pre-phase) + fit of tile i-1 (tail) in one launch")
        if mosaic_fused:
            carried = (f"K1+K2 of tile n + K3 of tile n-{ntl + 1} (pre-phase) + slot reduction of tile n-1 (tail; behind a step's last tile also "
                       f"the sum over the {ntl} tiles and the solve) in one launch, {ntl} launches per step")
        roof = {"bound": "hbm", "kernel": (f"srf_kernel<deg,fast,...,APPLY>: {carried}" if (fused or mosaic_fused) else "srf_kernel<deg,fast> (K1+K2 fused)")
                if args.cube == "f32" else
                "srf_u16_ring_kernel<deg%s> (K1+K2 fused, uint16 tile decode%s)%s" % (
                    ",...,APPLY" if (fused or mosaic_fused) else "", ", fast arithmetic" if args.u16_fast else "", (": " + carried) if (fused or mosaic_fused) else ""),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "traffic_source": None, "algorithmic_bytes": alg_bytes,
                "algorithmic_bytes_note": "SURVEY 8(d): H*W*285*4 per tile, the cube read exactly once (x tiles per launch)",
                "kernel_ms": round(k1_ms, 4),
                "kernel_launches_timed": len(k1_all),
                "kernel_ms_in_timed_region": round(sum(in_region) / max(1, len(in_region)), 4), "launches_in_timed_region": len(in_region),
                "kernel_ms_after_region": round(sum(extra) / max(1, len(extra)), 4) if extra else None,
                "agrees_with": "profiles/r04_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command): average duration of "
                               "the srf_kernel<..., true> (fused pipeline) / srf_kernel / srf_u16_ring_kernel row",
                "launch_bytes": launch_bytes,                 # everything of the cube + the carried K3 the launch must move
                "frac_launch_bytes": round(launch_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "step_frac_of_peak": round(ntl * cube_bytes * args.steps / dt_max / 1e9 / HBM_PEAK_GBS, 4),
                "total_fraction": round(ntl * full_bytes * args.steps / dt_max / 1e9 / HBM_PEAK_GBS, 4),
                "total_bytes_per_step": ntl * full_bytes}
        tf = os.path.join(ROOT, TRAFFIC_FILE)
        if ntl == 1 and (H, W, B) == (1024, 1024, 285) and os.path.isfile(tf):       # the PMC figure is per single-tile launch of this shape
            try:
                roof["traffic"] = json.load(open(tf)).get(("srf_fused_kernel_hbm_bytes_per_launch" if fused else "srf_kernel_hbm_bytes_per_launch")
                                                          if args.cube == "f32" else
                                                          ("srf_u16_fused_kernel_hbm_bytes_per_launch" if fused else "srf_u16_kernel_hbm_bytes_per_launch"))
                roof["traffic_source"] = (f"{TRAFFIC_FILE}: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 per launch from separate "
                                          f"rocprofv3 --pmc passes over this command (committed file, not measured in this run)")
            except Exception:
                pass
        if not args.no_probe:
            # ceiling of K1's own load shape on this box (non-temporal LDS-DMA, nothing computed) and a plain stream
            roof["measured_read_peak"] = round(eng.probe_read_bandwidth(1 << 30, 10, device, mode=0) / 1e9, 1)
            roof["measured_plain_read"] = round(eng.probe_read_bandwidth(1 << 30, 10, device, mode=1) / 1e9, 1)
        degraded = world > 1 and args.backend != "nccl"
        if strong:
            wl = (f"ONE {Hfull}x{W}x{B} EMIT-like cube + {nb} real-S2 planes split into {world} row block(s) of {H} rows, one per "
                  f"GPU, deg-{args.deg} per-band least squares over all valid pixels of the whole cube: one global fit per step "
                  f"(BASELINE.json configs[2] at N GPUs; SURVEY.md 8e row blocks)")
        else:
            wl = (f"{H}x{W}x{B} EMIT-like cube{' stored as uint16 x 10000 tiles (decode fused into K1)' if args.cube == 'u16' else ''} + {nb} real-S2 planes, "
                  f"{ntl} tile{'s' if ntl > 1 else ''} per GPU, deg-{args.deg} per-band least squares over all valid pixels "
                  + ("(BASELINE.json configs[2]; one tile per GPU for N>1)" if ntl == 1 else
                     f"of all {world * ntl} tiles: ONE global fit per step (BASELINE.json configs[3]/[4] mosaic)"))
        line = {"metric": "Mpixel*bands/s fused (SRF + deg-%d per-band LSQ fit + apply)" % args.deg,
                "value": round(value, 1), "unit": "Mpixel*bands/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt_max / args.steps * 1e3, 4),
                "host_issue_us_per_step": round(host_issue[0] * 1e6, 2),
                "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                "dtype": "f32" if args.cube == "f32" else "u16->f32",
                "data": "synthetic",
                "world_size": world, "rccl_ranks": exchange_ranks if (args.backend == "nccl" or args.force_exchange) else None,
                "exchange_ranks": exchange_ranks,
                "config": {"workload": wl,
                           "tiles_per_gpu": ntl, "rows_per_gpu": H, "coeff_sync": (args.coeff_sync if world > 1 else "none") +
                           (f" (rehearsal: one-rank RCCL {args.coeff_sync} forced)" if args.force_exchange else ""),
                           "pipeline": (("one kernel per tile with the exchange issued from C: K3 of tile i-3 as a pre-phase, K1+K2 of tile i, slot "
                                         "reduction of tile i-1 as tail work; gate -> " +
                                         (("RCCL " + args.coeff_sync + " through the library's own communicator") if comm is not None else
                                          "host transport (pinned round trip + torch.distributed on the CPU)") +
                                         f" -> solve on the side stream, no event or stream wait on the caller's stream, {args.reserve_cus} CUs "
                                         "reserved; " + str(fused_note)) if (fused and exchanging) else
                                        ("one kernel per tile: K3 of tile i-2 as a pre-phase, K1+K2 of tile i, fit of tile i-1 as tail work "
                                         "(no side stream, no events, no reserved CUs); " + str(fused_note)) if fused else
                                        f"one tile deep (two slots, fit on a side stream), {args.reserve_cus} CUs reserved") if pipelined else "off",
                           "exchange_transport": (plan._pipe or {}).get("transport") if pipelined else None,
                           "hsr_comm_error": comm_error,
                           "mosaic": ({"form": "group pipeline: one kernel per tile, one fit per step" if mosaic_fused else "batched: five launches per step",
                                       "note": mosaic_note, "placement": mosaic_log} if ntl > 1 else None),
                           "fake_collective": ({"us": args.fake_collective_us, "blocks": args.fake_collective_blocks}
                                               if (args.force_exchange and args.fake_collective_us > 0) else None),
                           "settle": {"ms": args.settle_ms, "untimed_steps": settle_steps,
                                      "note": "untimed load before the W warm-up steps: from idle a 20-step region sits in a "
                                              "power-management transient 6-10 % slower than the continuous-load state"},
                           "pipeline_note": fused_note,
                           "launches_per_step": ntl if mosaic_fused else 1 if fused else (2 if (args.fused_fit and world == 1 and not args.force_exchange and ntl == 1) else None),
                           "placement": {"trials": plan.placement_trials, "trials_ms": plan.placement_log.get(H * W),
                                         "joint_with_inputs": input_log is not None,
                                         "search_seconds": (input_log or {}).get("seconds"),
                                         "pitch_gb": plan.placement_pitch_gb, "budget_gb": args.placement_budget_gb,
                                         "held_gb": round(plan.placement_held_gb, 1),
                                         "note": f"before the warm-up K1 is timed on {plan.placement_trials} candidate allocations, " +
                                                 (f"{plan.placement_pitch_gb:g} GB apart" if plan.placement_pitch_gb > 0 else "back to back (a dense map)") +
                                                 f", of (cube copy, target copy, output images); the fastest set is kept "
                                                 f"(profiles/r02_two_speeds.md, r03_placement_mechanism.md); same bytes, bit-identical "
                                                 f"results; line['cold'] is the step without any of this"},
                           "backend": (args.backend if world > 1 else "none") +
                           (" (rehearsal: all ranks on cuda:0)" if args.same_device else "")},
                "roofline": roof}
        if cold is not None:
            line["cold"] = cold
        if degraded:
            line["degraded"] = True         # exchange over gloo (host staged): a rehearsal, not an RCCL/xGMI measurement
        if want_cpu:
            fo = out if not isinstance(out, tuple) else out[2][0]
            cube_host = prob.cube.cpu().numpy()
            real_host = prob.real_planes.cpu().numpy()
            gp = fo.planes("pseudo").cpu().numpy().reshape(nb, H, W)
            gm = fo.planes("matched").cpu().numpy().reshape(nb, H, W)
            line["cpu_baseline"], line["max_rel_err"] = cpu_baseline(args, pool, nworkers, cube_host, real_host, gp, gm)
            pool.shutdown()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1 or args.force_exchange:
        dist.barrier()
        plan.close()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
