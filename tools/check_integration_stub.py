"""Runs the ctypes stub printed in INTEGRATION.md section 2 exactly as a maintainer of the reference would write it
(no import from this package besides the shared library), against the package's own result."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "hyperspectral_super-resolution_amd", "lib", "libhsr_mi355x.so"))
_lib.hsr_srf_integrate.restype = C.c_int
_lib.hsr_srf_integrate.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                   C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
_lib.hsr_last_error.restype = C.c_char_p


def srf_integrate(cube, Wn, k0, klen):
    npix, B, nb = cube.shape[0] * cube.shape[1], cube.shape[2], Wn.shape[0]
    out = torch.empty((nb, npix), dtype=torch.float32, device=cube.device)
    rc = _lib.hsr_srf_integrate(cube.data_ptr(), npix, B, Wn.data_ptr(),
                                k0.ctypes.data_as(C.POINTER(C.c_int32)),
                                klen.ctypes.data_as(C.POINTER(C.c_int32)), nb,
                                out.data_ptr(), npix, 1,
                                torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError(_lib.hsr_last_error().decode())
    return out


sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
from oracle import oracle_np as onp
import s2_emit
srf = onp.synthetic_srf(); w, good = onp.synthetic_wavelengths()
R = onp.synthetic_cube(40, 30, seed=3)
# weights as the stub's docstring says: row b = r_b * (d_{k-1} + d_k)/2 / (trapz(r_b) + 1e-32)
Wn64, names = onp.srf_weight_matrix(w, srf, good)
Wn = np.ascontiguousarray(Wn64, dtype=np.float32)
nz = [np.flatnonzero(r) for r in Wn]
k0 = np.array([z[0] for z in nz], np.int32); klen = np.array([z[-1] - z[0] + 1 for z in nz], np.int32)
planes = srf_integrate(torch.from_numpy(R).cuda(), torch.from_numpy(Wn).cuda(), k0, klen).cpu().numpy()
ref = s2_emit.pseudo_s2_srf_integral(R, w, srf, good)
for i, k in enumerate(names):
    assert np.array_equal(planes[i].reshape(40, 30), ref[k].astype(np.float32)), k
print("INTEGRATION.md stub: identical to the package result for", len(names), "bands")
