"""Synthetic EMIT / Sentinel-2 pairs of the shape the benchmark is quoted on.

There is no network on the GPU nodes, so benchmarks and large-size tests use seeded synthetic data:
a low-rank reflectance cube (3 smooth endmembers x random abundances + noise, clipped to the EMIT
value range), EMIT-like wavelengths 381-2493 nm with the two water-vapour windows masked, 13
Gaussian SRFs at the nominal S2A centres/widths on a 1-nm grid (entries > 1e-3 kept, so B10 has no
support), and "real S2" planes that are a monotone polynomial-like function of the pseudo planes.
The big arrays are generated directly in HBM with torch (plumbing); the small tables are NumPy.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from . import _native as nat
from .srf import S2_BANDS_13

S2A_CENTRES = [443, 490, 560, 665, 705, 740, 783, 842, 865, 945, 1375, 1610, 2190]
S2A_WIDTHS = [20, 65, 35, 30, 15, 15, 20, 115, 20, 20, 30, 90, 180]


def gaussian_srf(threshold: float = 1e-3) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    lam = np.arange(300.0, 2600.0, 1.0)
    srf = {}
    for name, c, fwhm in zip(S2_BANDS_13, S2A_CENTRES, S2A_WIDTHS):
        r = np.exp(-0.5 * ((lam - c) / (fwhm / 2.3548200450309493)) ** 2)
        m = r > threshold
        srf[name] = (lam[m].copy(), r[m].copy())
    return srf


def emit_wavelengths(B: int = 285):
    w = np.linspace(381.00558, 2492.9, B, dtype=np.float32)
    wf = w.astype(float)
    good = ~(((wf > 1320) & (wf < 1440)) | ((wf > 1770) & (wf < 1970)))
    return w, good


def endmembers(B: int = 285) -> np.ndarray:
    w, _ = emit_wavelengths(B)
    t = (w.astype(float) - 381.0) / (2493.0 - 381.0)
    return np.stack([0.08 + 0.35 * np.exp(-((t - 0.25) / 0.3) ** 2),
                     0.05 + 0.45 * t * np.exp(-t * 1.5) * 2.0,
                     0.30 - 0.2 * t + 0.05 * np.sin(6 * t)])


@dataclass
class DeviceProblem:
    emit_w: np.ndarray
    good_mask: np.ndarray
    srf: Dict[str, Tuple[np.ndarray, np.ndarray]]
    cube: object        # (H, W, B) float32 on the GPU
    real: object        # (H, W, row) float32 on the GPU, band-last, row = nb padded to a multiple of 4
    names: list
    real_planes: object = None   # (nb, H, W) band-major copy (tests)


def device_problem(H: int, W: int, B: int = 285, deg: int = 3, seed: int = 0, device="cuda") -> DeviceProblem:
    torch = nat.require_gpu()
    from . import _engine as eng
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    w, good = emit_wavelengths(B)
    srf = gaussian_srf()
    E = torch.from_numpy(endmembers(B).astype(np.float32)).to(device)
    A = torch.rand((H * W, 3), generator=gen, device=device)
    A /= A.sum(dim=1, keepdim=True)
    cube = A @ E
    del A
    # noise added in row blocks to bound the temporary
    step = max(1, (1 << 26) // B)
    for r0 in range(0, H * W, step):
        blk = cube[r0:r0 + step]
        blk.add_(torch.randn(blk.shape, generator=gen, device=device, dtype=torch.float32), alpha=0.005)
    cube.clamp_(-0.01, 0.6)
    cube = cube.reshape(H, W, B).contiguous()
    table = eng.build_srf_table(w, srf, good)
    pseudo = eng.srf_integrate(cube, table, layout=nat.PLANAR)
    nb = table.nb
    rs = np.random.default_rng(seed + 1)
    g = torch.from_numpy((0.8 + 0.4 * rs.random(nb)).astype(np.float32)).to(device)[:, None]
    gam = torch.from_numpy((0.8 + 0.4 * rs.random(nb)).astype(np.float32)).to(device)[:, None]
    o = torch.from_numpy((0.02 * rs.random(nb)).astype(np.float32)).to(device)[:, None]
    real = g * pseudo.clamp(min=1e-6) ** gam + o
    real.add_(torch.randn(real.shape, generator=gen, device=device, dtype=torch.float32), alpha=0.01)
    real.clamp_(0.0, 1.0)
    row = eng.padded_row(nb)
    real_bl = torch.zeros((H * W, row), dtype=torch.float32, device=device)
    real_bl[:, :nb] = real.t()
    return DeviceProblem(w, good, srf, cube, real_bl.reshape(H, W, row), list(table.supported),
                         real.reshape(nb, H, W).contiguous())
