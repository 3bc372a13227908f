import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hyperspectral_super-resolution_amd")):
    sys.path.insert(0, p)
import torch
from s2_emit import _engine as eng
torch.cuda.set_device(0)
for nbytes in (1024 * 1024 * 285 * 2, 1024 * 1024 * 285 * 4):
    for mode in (0, 2, 3, 1):
        r = [round(eng.probe_read_bandwidth(nbytes, 10, "cuda:0", mode) / 1e12, 2) for _ in range(3)]
        print(f"{nbytes / 1e6:.0f} MB mode {mode}: TB/s", r)
