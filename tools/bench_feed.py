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

# ENVI files in their native interleave: file (page cache) -> pinned staging in file order -> H2D -> GPU
# transpose -> step.  The host's pass over the samples (memmap -> pinned) is a CPU memcpy and is what bounds this path.
import tempfile
import numpy as np
from s2_emit.emit_io import EnviCubeFile
tmp = tempfile.mkdtemp(prefix="hsr_feed_")
cube_h = prob.cube.cpu().numpy()
files = {}
for inter, lay in (("bil", cube_h.transpose(0, 2, 1)), ("bsq", cube_h.transpose(2, 0, 1)), ("bip", cube_h)):
    np.ascontiguousarray(lay).tofile(os.path.join(tmp, f"c_{inter}.bin"))
    open(os.path.join(tmp, f"c_{inter}.hdr"), "w").write(
        f"ENVI\nsamples = 1024\nlines = 1024\nbands = 285\nheader offset = 0\ndata type = 4\ninterleave = {inter}\nbyte order = 0\n")
    files[inter] = EnviCubeFile(os.path.join(tmp, f"c_{inter}.hdr"), os.path.join(tmp, f"c_{inter}.bin"))
for inter in ("bil", "bsq", "bip"):
    tiles = [(files[inter], real_h) for _ in range(8)]
    list(plan.stream(tiles[:3], depth=2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(1 for _ in plan.stream(tiles, depth=2))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"ENVI {inter} file feed: {n} tiles in {dt*1e3:.1f} ms = {dt/n*1e3:.2f} ms/tile, {n*npb/dt/1e6:.0f} Mpix*bands/s "
          f"(file -> pinned staging is a host memcpy: {1.195/(dt/n):.1f} GB/s end to end)", flush=True)
# the transpose alone
raw = torch.from_numpy(np.fromfile(os.path.join(tmp, "c_bil.bin"), dtype=np.float32)).cuda()
for inter in ("bil", "bsq"):
    f = files[inter]
    out = torch.empty((1024, 1024, 285), dtype=torch.float32, device="cuda")
    f.to_bip(raw, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f.to_bip(raw, out=out)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"hsr_interleave_to_bip {inter}: {ms:.3f} ms per 1024x1024x285 float32 cube = {2*1.195e9/ms/1e9:.2f} TB/s of read+write traffic")
import shutil
shutil.rmtree(tmp, ignore_errors=True)
