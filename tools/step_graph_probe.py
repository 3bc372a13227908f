"""Diagnostic: is the 1-GPU step launch-bound?  Eager step() loop vs a captured hipGraph replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion
from s2_emit.synthetic import device_problem

dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
K = 200
for _ in range(10):
    plan.step(prob.cube, prob.real)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        plan.step(prob.cube, prob.real)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager: issue {1e3*(t1-t0)/K:.4f} ms/step, total {1e3*(t2-t0)/K:.4f} ms/step", flush=True)

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plan.step(prob.cube, prob.real)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    out = plan.step(prob.cube, prob.real)
torch.cuda.synchronize()
ref = plan.step(prob.cube, prob.real)
c0 = ref.coeffs.clone(); m0 = ref.matched.clone()
g.replay(); torch.cuda.synchronize()
print("graph replay bit-identical:", bool(torch.equal(out.coeffs, c0)) and bool(torch.equal(out.matched.view(torch.int32), m0.view(torch.int32))), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graph: issue {1e3*(t1-t0)/K:.4f} ms/step, total {1e3*(t2-t0)/K:.4f} ms/step", flush=True)
