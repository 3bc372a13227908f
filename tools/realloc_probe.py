"""Is the per-process K1 speed (0.205 vs 0.224 ms) tied to where the buffers landed?  Re-create everything several
times inside one process (empty_cache in between, padding allocations to shift placement) and time K1 each time."""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
def measure(tag, pad_mb):
    pad = torch.empty(pad_mb << 18, dtype=torch.float32, device=dev) if pad_mb else None
    prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
    plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
    for _ in range(150):
        plan.step(prob.cube, prob.real)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for e in evs:
        plan.step(prob.cube, prob.real, k1_events=e)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print("%s: cube %x real %x  K1 %.4f ms" % (tag, prob.cube.data_ptr(), prob.real.data_ptr(), ts[20]), flush=True)
    del prob, plan, pad
    gc.collect(); torch.cuda.empty_cache()
for i, pad in enumerate((0, 0, 37, 0, 513, 1, 1025, 0)):
    measure("round %d (pad %4d MB)" % (i, pad), pad)
