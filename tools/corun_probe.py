"""Can a small kernel on a high-priority stream run while the persistent K1 owns the chip?
For several numbers of CUs left without K1 workgroups: time of a tiny kernel launched ~30 us after K1 started."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
import torch
from s2_emit import SpectralFusion, _native as nat, _engine as eng
from s2_emit.synthetic import device_problem
dev = torch.device("cuda", 0)
prob = device_problem(1024, 1024, 285, deg=3, seed=0, device=dev)
plan = SpectralFusion(prob.emit_w, prob.srf, prob.good_mask, deg=3, min_valid=0.0, min_count=50, clip=True, device=dev)
side = torch.cuda.Stream(device=dev, priority=-1)
main = torch.cuda.current_stream()
small = torch.zeros(4096, device=dev)
ws = eng.MomentWorkspace(dev, plan.table.nb, 3)
out = None
def tiny_add():
    small.add_(1.0)                  # a 16-workgroup kernel
def fit_reduce():
    eng.moments_reduce(ws)           # 132 workgroups of one wave
def fit_solve():
    eng.poly_solve(ws.moments, 3, 50, out=ws.coeffs)   # one workgroup, uses scratch
def fit_both():
    eng.moments_reduce(ws)
    eng.poly_solve(ws.moments, 3, 50, out=ws.coeffs)
for reserve, delay, fn in ((0, 0.0, fit_both), (4, 0.0, fit_both), (6, 0.0, fit_both), (8, 0.0, fit_both), (10, 0.0, fit_both),
                           (12, 0.0, fit_both), (16, 0.0, fit_both), (8, 0.0, tiny_add), (8, 0.0, fit_reduce), (8, 0.0, fit_solve)):
    nat.check(nat.load().hsr_set_srf_reserved_cus(reserve))
    for _ in range(3):
        plan.step(prob.cube, prob.real)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        e0, e1, k0, k1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        start = torch.cuda.Event()
        k0.record(main)
        pseudo, _ = eng.srf_integrate_moments(prob.cube, plan.table, prob.real.reshape(-1, prob.real.shape[-1]), 3, ws, None, 0.0, 0.0,
                                              out=out, reduce=False, layout="pixmajor", real_layout="pixmajor")
        out = pseudo
        k1.record(main)
        if delay:
            time.sleep(delay)               # let K1 get going (the CPU is ~instant, K1 takes 0.22 ms)
        with torch.cuda.stream(side):
            e0.record(side)
            fn()
            e1.record(side)
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) * 1e3, k0.elapsed_time(k1) * 1e3, k0.elapsed_time(e1) * 1e3))
    r = sorted(res)[len(res) // 2]
    print(f"reserved CUs {reserve:3d}, delay {delay*1e6:3.0f} us, {fn.__name__:10s}: side kernel {r[0]:7.1f} us (K1 {r[1]:6.1f} us; tiny finished {r[2]:6.1f} us after K1 was enqueued)", flush=True)
