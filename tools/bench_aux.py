"""Auxiliary kernels of the path at the sizes of a 1024 x 1024 EMIT grid <-> 6144 x 6144 Sentinel-2 grid (factor 6):
resamplers (f1), masked percentiles + stretch (a4), validity mask, K3 alone.  HIP-event times, algorithmic GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import _engine as eng

dev = "cuda"
torch.manual_seed(0)
Hc = Wc = 1024
f = 6
Hf, Wf = Hc * f, Wc * f


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def row(name, ms, nbytes, note=""):
    print(f"| {name} | {ms*1e3:.1f} us | {nbytes/1e6:.1f} MB | {nbytes/ms/1e6:.0f} GB/s | {note} |", flush=True)


print("| kernel (shape) | time | algorithmic bytes | rate | note |\n|---|---|---|---|---|")
# f1: S2 visual uint8 RGB 6144^2 x 3 (band-last, as the file holds it) -> 1024^2 x 3 float32 /255, and back up
s2_u8 = torch.randint(0, 256, (Hf * Wf, 3), dtype=torch.uint8, device=dev)
ms = timed(lambda: eng.block_mean(s2_u8, Hc, Wc, f, 1.0 / 255.0, "pixmajor", "pixmajor", nb=3))
row("block_mean uint8 6144^2x3 -> 1024^2x3 (pixel-major)", ms, Hf * Wf * 3 + Hc * Wc * 4 * 4)
s2_f = torch.rand((3, Hf * Wf), device=dev)
ms = timed(lambda: eng.block_mean(s2_f, Hc, Wc, f, 1.0, "planar"))
row("block_mean float32 3 planes 6144^2 -> 1024^2", ms, 3 * (Hf * Wf + Hc * Wc) * 4)
lo = torch.rand((3, Hc * Wc), device=dev)
ms = timed(lambda: eng.bilinear_upsample(lo, Hc, Wc, f, "planar"))
row("bilinear_upsample 3 planes 1024^2 -> 6144^2", ms, 3 * (Hf * Wf + Hc * Wc) * 4)
lo_pm = torch.rand((Hc * Wc, 4), device=dev)
ms = timed(lambda: eng.bilinear_upsample(lo_pm, Hc, Wc, f, "pixmajor", "pixmajor", nb=3))
row("bilinear_upsample pixel-major 1024^2x3 -> 6144^2x3(4)", ms, (Hf * Wf + Hc * Wc) * 16)
# a4: exact masked percentiles (3 radix passes over the masked samples) and the stretch
for name, n in (("1024^2", Hc * Wc), ("6144^2", Hf * Wf)):
    x = torch.rand((3, n), device=dev)
    m = (torch.rand(n, device=dev) > 0.1).to(torch.uint8)
    ms = timed(lambda: eng.percentile_limits(x, m, 2, 98))
    row(f"percentile_limits 3 planes {name}, 90 % masked in", ms, 3 * 3 * n * 4 + 3 * n, "3 passes x (4 B sample + 1 B mask)")
    lohi = eng.percentile_limits(x, m, 2, 98)
    ms = timed(lambda: eng.poly_apply_stretch_only(x, lohi))
    row(f"stretch (K3 without polynomial) 3 planes {name}", ms, 2 * 3 * n * 4)
for name, n in (("1024^2", Hc * Wc), ("6144^2", Hf * Wf)):
    xr = torch.rand((n, 4), device=dev)
    m = (torch.rand(n, device=dev) > 0.1).to(torch.uint8)
    ms = timed(lambda: eng.percentile_limits(xr, m, 2, 98, "pixmajor", nb=3))
    row(f"percentile_limits band-last rows of 4 (3 channels) {name}", ms, 3 * n * 17, "3 passes x (16 B row + 1 B mask)")
x12 = torch.rand((Hc * Wc, 12), device=dev)
y3 = torch.rand((Hc * Wc, 3), device=dev)
ms = timed(lambda: eng.valid_mask(x12, 0, y3, None, "pixmajor"))
row("valid_mask 1024^2 x (12 + 3) pixel-major", ms, Hc * Wc * (15 * 4 + 1))
co = torch.rand((12, 4), dtype=torch.float64, device=dev)
ms = timed(lambda: eng.poly_apply(x12, co, None, None, True, "pixmajor"))
row("poly_apply deg 3, 1024^2 x 12 pixel-major (K3 of the headline)", ms, 2 * Hc * Wc * 12 * 4)
xp = torch.rand((12, Hc * Wc), device=dev)
ms = timed(lambda: eng.poly_apply(xp, co, None, None, True, "planar"))
row("poly_apply deg 3, 12 planes 1024^2", ms, 2 * Hc * Wc * 12 * 4)
