"""Does the placement of the cube in memory decide between the two K1 speeds seen from run to run (0.212 vs 0.229 ms)?
One process, one big allocation, the cube copied to different offsets inside it; K1+K2 timed at each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
n = prob.cube.numel()
big = torch.empty(n + (64 << 20) // 4, dtype=torch.float32, device=dev)
print("cube ptr %x  big ptr %x  real ptr %x" % (prob.cube.data_ptr(), big.data_ptr(), prob.real.data_ptr()), flush=True)
def k1_time(cube):
    for _ in range(5):
        plan.step(cube, prob.real)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e in evs:
        plan.step(cube, prob.real, k1_events=e)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]
print("original cube : %.4f ms" % k1_time(prob.cube), flush=True)
for off_bytes in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 8 << 20, 17 << 20, 33 << 20, 63 << 20):
    view = big[off_bytes // 4: off_bytes // 4 + n].view(1024, 1024, 285)
    view.copy_(prob.cube)
    print("offset %9d B: %.4f ms" % (off_bytes, k1_time(view)), flush=True)
print("original again: %.4f ms" % k1_time(prob.cube), flush=True)
