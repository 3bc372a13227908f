import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
u = eng.tile_encode_u16(prob.cube)
torch.cuda.synchronize()
def run(cube, n, tag):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e in evs:
        plan.step(cube, prob.real, k1_events=e)
    torch.cuda.synchronize()
    print(tag, " ".join(f"{a.elapsed_time(b):.3f}" for a, b in evs), flush=True)
run(u, 24, "u16 first:")
run(u, 12, "u16 again:")
run(prob.cube, 12, "f32      :")
run(u, 12, "u16 after f32:")
u2 = u.clone()
run(u2, 12, "u16 clone:")
