"""Phase timeline of gram_f64_lds_kernel (library built with -DHSR_GRAM_STAMPS: tools/dbg/build_variants.sh
"gstamp:-DHSR_GRAM_STAMPS:hsr_ridge"; HSR_LIBRARY=tools/dbg/libhsr_gstamp.so).  s_memrealtime (100 MHz) at: entry, end of
the prologue, end of the batch loop, end of the group combine, partial sums stored."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import numpy as np, torch
from s2_emit import _native as nat
from s2_emit._engine import _ptr, _stream
lib = nat.load()
raw = C.CDLL(os.environ["HSR_LIBRARY"])
raw.hsr_dbg_gram_stamps.argtypes = [C.c_void_p]
n, na = 29127, 288
for T in (32, 285):
    nb = na + (T + 15) // 16 * 16
    Q = torch.rand((n, nb), device="cuda", dtype=torch.float64)
    work = torch.empty(lib.hsr_gram_work_bytes(na, nb, n) // 8, dtype=torch.float64, device="cuda")
    G = torch.empty((na, nb), dtype=torch.float64, device="cuda")
    st = torch.zeros((8192, 8), dtype=torch.int64, device="cuda")
    def run():
        nat.check(lib.hsr_gram_f64(_ptr(Q), nb, na, _ptr(Q), nb, nb, n, _ptr(work), _ptr(G), nb, _stream(torch)))
    for _ in range(3): run()
    raw.hsr_dbg_gram_stamps(C.c_void_p(st.data_ptr()))
    torch.cuda.synchronize(); run(); torch.cuda.synchronize()
    raw.hsr_dbg_gram_stamps(None)
    s = st.cpu().numpy()
    s = s[s[:, 4] > 0]
    t0 = s[:, 0].min()
    us = (s[:, :5] - t0) / 100.0
    kind = s[:, 5] // 1000000
    print(f"T={T}: {len(s)} workgroups with stamps; kernel span {us[:, 4].max():.1f} us from the first entry")
    for k in np.unique(kind):
        u = us[kind == k]
        nbt = s[kind == k, 5] % 1000000
        print(f"  kind {k} (3 = wide 3 x 3, 1 = narrow 3 x 1, 2 = diagonal): {len(u)} workgroups, local batches {nbt.min()}..{nbt.max()}")
        for i, name in enumerate(("entry", "prologue done", "loop done", "combine done", "stored")):
            print(f"    {name:14s} min {u[:, i].min():7.1f}  median {np.median(u[:, i]):7.1f}  max {u[:, i].max():7.1f} us")
        d = u[:, 2] - u[:, 1]
        cyc = (s[kind == k, 7] - s[kind == k, 6]).astype(np.float64)
        print(f"    loop in shader-clock cycles (s_memtime): median {np.median(cyc):.0f} = {np.median(cyc) / np.median(nbt):.0f} per local batch; "
              f"cycles / wall time = {np.median(cyc / d) / 1e3:.3f} GHz")
        print(f"    loop duration  min {d.min():7.1f}  median {np.median(d):7.1f}  max {d.max():7.1f} us; per local batch {np.median(d) / np.median(nbt) * 1e3:.0f} ns")
