import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
os.environ["HSR_LIBRARY"] = os.path.join(ROOT, "tools", "dbg", "libhsr_cholprof.so")
import torch, s2_emit
from s2_emit import _native as nat
g = torch.Generator(device="cuda").manual_seed(0)
X = (600 + 4600 * torch.rand((29127, 10), generator=g, device="cuda")).float()
Y = torch.logit((0.02 + 0.5 * torch.rand((29127, 32), generator=g, device="cuda")).double())
m = s2_emit.PolyRidge(3, 1.0)
for _ in range(5): m.fit(X, Y)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["HSR_LIBRARY"])
out = (ctypes.c_ulonglong * 4)()
lib.hsr_chol_prof(out)
tot = sum(out)
print("factor kernel cycles (thread 0): load D %d, diag block %d, panel %d, update %d  (total %d)" % (out[0], out[1], out[2], out[3], tot))
print("shares: load %.1f%% diag %.1f%% panel %.1f%% update %.1f%%" % tuple(100.0 * o / tot for o in out))
