import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch, s2_emit
g = torch.Generator(device="cuda").manual_seed(0)
X = (600 + 4600 * torch.rand((29127, 10), generator=g, device="cuda")).float()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Y = torch.logit((0.02 + 0.5 * torch.rand((29127, T), generator=g, device="cuda")).double())
m = s2_emit.PolyRidge(3, 1.0)
for _ in range(3): m.fit(X, Y)
torch.cuda.synchronize()
for _ in range(10): m.fit(X, Y)
torch.cuda.synchronize()
