import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "hyperspectral_super-resolution_amd"))
import numpy as np, torch
import s2_emit
rng = np.random.default_rng(0)
H = W = 1024
X = torch.rand((10, H, W), device="cuda") * 0.5 + 0.1
for T in (32, 24):
    m = s2_emit.PolyRidge.from_params(np.full(10, 0.35), np.full(10, 0.15), rng.normal(0, 0.05, (T, 285)), rng.normal(0, 0.1, T))
    for _ in range(3): out = m.predict_cube(X)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): out = m.predict_cube(X)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{os.environ.get('HSR_LIBRARY','prod')[-12:]} T={T:4d}: {ms:.4f} ms per Mpixel, {2 * 286 * T * H * W / ms / 1e9:.1f} TFLOP/s, checksum {float(out.float().sum()):.6e}", flush=True)
