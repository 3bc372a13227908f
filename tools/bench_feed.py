"""PCIe-inclusive rate of the hot path: tiles in pinned host memory -> SpectralFusion.stream() -> results in
pinned host memory (1024 x 1024 x 285 tiles, deg 3).  H2D of tile i+1 overlaps K1..K3 of tile i."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion, _engine as eng
from s2_emit.synthetic import device_problem

dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
npb = 1024 * 1024 * 285
real_h = prob.real.cpu().pin_memory()
src = {"f32": [prob.cube.cpu().pin_memory(), prob.cube.flip(0).contiguous().cpu().pin_memory()]}
u = eng.tile_encode_u16(prob.cube)
src["u16"] = [u.cpu().pin_memory(), u.flip(0).contiguous().cpu().pin_memory()]
ntiles = 16
for kind in ("f32", "u16", "f32", "u16"):
    tiles = [(src[kind][i % 2], real_h) for i in range(ntiles)]
    for depth in (2, 3):
        list(plan.stream(tiles[:3], depth=depth))           # warm-up (allocations, pinned result buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = sum(1 for _ in plan.stream(tiles, depth=depth))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gb = (src[kind][0].numel() * src[kind][0].element_size() + real_h.numel() * 4) / 1e9
        print(f"{kind} depth {depth}: {n} tiles in {dt*1e3:.1f} ms = {dt/n*1e3:.2f} ms/tile, {n*npb/dt/1e6:.0f} Mpix*bands/s, "
              f"H2D {gb*n/dt:.1f} GB/s", flush=True)
