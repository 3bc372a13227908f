import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import _engine as eng
torch.cuda.set_device(0)
x = torch.rand((1024 * 1024, 12), device="cuda")
o = torch.empty_like(x)
co = torch.tensor([[0.3, -0.5, 1.1, 0.01]] * 12, dtype=torch.float64, device="cuda")
def run():
    eng.poly_apply(x, co, None, None, True, "pixmajor", out=o, nb=12)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(20): run()
    e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(os.environ.get("HSR_LIBRARY", "prod").split("_")[-1], "K3 back-to-back us:", " ".join(f"{t*1000:.1f}" for t in ts))
