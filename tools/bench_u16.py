"""K1+K2 on a uint16 tile cube vs the float32 cube (1024 x 1024 x 285, deg 3): kernel time by HIP events."""
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
npb = 1024 * 1024 * 285
for name, cube, bytes_per in (("f32", prob.cube, 4), ("u16", u, 2), ("f32", prob.cube, 4), ("u16", u, 2)):
    for _ in range(10):
        plan.step(cube, prob.real)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    for e in evs:
        plan.step(cube, prob.real, k1_events=e)
    t1.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    if ts[-1] > 3 * ts[len(ts) // 2]:
        print("  outliers (ms):", " ".join(f"{a.elapsed_time(b):.3f}" for a, b in evs), flush=True)
    k1 = sum(ts) / len(ts)
    step = t0.elapsed_time(t1) / len(evs)
    print(f"{name}: K1+K2 {k1:.4f} ms ({npb*bytes_per/k1/1e6:.0f} GB/s of cube bytes), step {step:.4f} ms = {npb/step/1e3:.0f} Mpix*bands/s", flush=True)
# decode / encode alone
for name, fn, nbytes in (("encode", lambda: eng.tile_encode_u16(prob.cube), npb * 6), ("decode", lambda: eng.tile_decode_u16(u), npb * 6)):
    for _ in range(3):
        fn()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        fn()
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 10
    print(f"{name}: {ms:.4f} ms, {nbytes/ms/1e6:.0f} GB/s", flush=True)
