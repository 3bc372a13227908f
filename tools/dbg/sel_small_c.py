#!/usr/bin/env python3
"""hsr_percentile_limits called straight through ctypes with a preallocated workspace (no Python allocation per call): GPU time per
call at the reference's tile sizes."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import _native as nat
lib = nat.load()
torch.manual_seed(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for side in (100, 300, 600, 1024, 1448):
    n = side * side
    for layout in ("planes", "rows4"):
        x = (torch.rand((3, n), device="cuda") ** 2) if layout == "planes" else (torch.rand((n, 4), device="cuda") ** 2)
        m = (torch.rand(n, device="cuda") > 0.1).to(torch.uint8)
        work = torch.empty(lib.hsr_percentile_work_bytes(3) // 8 + 1, dtype=torch.int64, device="cuda")
        lohi = torch.empty((3, 2), dtype=torch.float64, device="cuda")
        bs, ps = (n, 1) if layout == "planes" else (1, 4)
        def call():
            nat.check(lib.hsr_percentile_limits(x.data_ptr(), bs, ps, m.data_ptr(), n, 3, 2.0, 98.0, work.data_ptr(), lohi.data_ptr(), st))
        for _ in range(10): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): call()
        e1.record(); torch.cuda.synchronize()
        print(f"{os.environ.get('HSR_LIBRARY', 'prod')[-14:]:>14} | {side} x {side} x 3 {layout} | {e0.elapsed_time(e1) / 200 * 1e3:.1f} us", flush=True)
