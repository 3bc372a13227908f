"""hsr_gram_f64 on the fit's shapes: [1|Phi]^T [1|Phi|Y] for 29 127 pixels, 288 padded features, T targets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import _native as nat
from s2_emit._engine import _ptr, _stream
lib = nat.load()
n, na = 29127, 288
for T in (32, 285):
    nb = na + (T + 15) // 16 * 16
    Q = torch.rand((n, nb), device="cuda", dtype=torch.float64)
    work = torch.empty(lib.hsr_gram_work_bytes(na, nb, n) // 8, dtype=torch.float64, device="cuda")
    G = torch.empty((na, nb), dtype=torch.float64, device="cuda")
    def run():
        nat.check(lib.hsr_gram_f64(_ptr(Q), nb, na, _ptr(Q), nb, nb, n, _ptr(work), _ptr(G), nb, _stream(torch)))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = Q[:, :na].T @ Q
    err = float((G - ref).abs().max() / ref.abs().max())
    full = 2.0 * n * na * nb
    print(f"T={T}: {us:.1f} us (gram + reduce), {full/us/1e6:.1f} TFLOP/s of the full {na}x{nb} Gram, max rel diff vs torch {err:.1e}", flush=True)
