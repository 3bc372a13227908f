"""Map of K1's speed over the device address space: N candidate sets (cube clone, target clone, output image), allocated
back to back, `gap` GB of spacer between sets; one K1 time per set.  Also: the same sets re-timed in reverse order (is
the speed a property of the set, or of time?)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion, _engine as eng, _native as nat
from s2_emit.synthetic import device_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device("cuda:0")
p = device_problem(1024, 1024, 285, deg=3, seed=0)
plan = SpectralFusion(p.emit_w, p.srf, p.good_mask, deg=3, placement_trials=0)
npix = 1024 * 1024
nb = plan.table.nb
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def k1(c, r, o):
    rr, rl = plan._real_image(r, npix)
    eng.srf_integrate_moments(c, plan.table, rr, 3, plan.ws, None, 0.0, 0.0, out=o, reduce=False, layout=plan.layout,
                              real_layout=rl, opts=plan.opts)


def t(c, r, o):
    k1(c, r, o)
    best = 9
    for _ in range(3):
        e0.record(); k1(c, r, o); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


sets, spacers = [], []
for i in range(N):
    if i:
        spacers.append(torch.empty(int(gap * (1 << 30)), dtype=torch.uint8, device=dev))
    c = p.cube if i == 0 else p.cube.clone()
    r = p.real if i == 0 else p.real.clone()
    o = eng.alloc_image(torch, nb, npix, plan.layout, dev)
    sets.append((c, r, o))
fw = [t(*s) for s in sets]
bw = [t(*s) for s in reversed(sets)][::-1]
print("addr GB  ", " ".join(f"{(s[0].data_ptr() >> 30) & 0xfff:5d}" for s in sets))
print("forward  ", " ".join(f"{x:.3f}" for x in fw))
print("backward ", " ".join(f"{x:.3f}" for x in bw))
# which operand carries it: best set's cube with worst set's out etc.
b, w = min(range(N), key=fw.__getitem__), max(range(N), key=fw.__getitem__)
print(f"best set {b} {fw[b]:.3f}  worst set {w} {fw[w]:.3f}")
for nm, (ci, ri, oi) in {"cube worst": (w, b, b), "real worst": (b, w, b), "out worst": (b, b, w), "cube best": (b, w, w),
                          "real best": (w, b, w), "out best": (w, w, b)}.items():
    print(f"  {nm:11s} {t(sets[ci][0], sets[ri][1], sets[oi][2]):.3f}")
free, total = torch.cuda.mem_get_info()
print(f"free {free / 2**30:.1f} of {total / 2**30:.1f} GB")
