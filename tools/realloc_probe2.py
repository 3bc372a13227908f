"""Which buffer's placement flips K1 between 0.206 and 0.224 ms?  Cube and targets stay; only the plan's own buffers
(pseudo / matched / partial sums) are re-created, with padding allocations of various sizes in front."""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
def measure(tag, pad_mb, real):
    pad = torch.empty(max(pad_mb, 1) << 18, dtype=torch.float32, device=dev)
    plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
    for _ in range(150):
        out = plan.step(prob.cube, real)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for e in evs:
        plan.step(prob.cube, real, k1_events=e)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print("%s: pseudo %x  real %x  K1 %.4f ms" % (tag, out.pseudo.data_ptr(), real.data_ptr(), ts[20]), flush=True)
    del plan, pad, out
    gc.collect(); torch.cuda.empty_cache()
print("cube %x" % prob.cube.data_ptr())
for i, pad in enumerate((0, 3, 37, 70, 129, 513, 0, 3)):
    measure("plan buffers, pad %4d MB" % pad, pad, prob.real)
for i in range(4):
    r2 = prob.real.clone()
    measure("new copy of real #%d     " % i, 0, r2)
    keep = r2  # keep it allocated so that the next clone lands elsewhere
