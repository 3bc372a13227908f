"""pytest configuration: marker registration and import paths.

``-m "not gpu"``: oracle vs golden fixtures, host logic, C-ABI symbol checks, gloo multi-process.
``-m gpu``: parity tests proper (HIP path through the C-ABI vs oracle / fixtures) on an MI355X.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hyperspectral_super-resolution_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack_srf(g):
    names = [str(n) for n in g["srf_names"]]
    lens = g["srf_lens"]
    offs = np.concatenate([[0], np.cumsum(lens)])
    return {n: (g["srf_lam"][offs[i]:offs[i + 1]], g["srf_rsp"][offs[i]:offs[i + 1]])
            for i, n in enumerate(names)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def g11_cube(g):
    """The 10 x 600 x 600 input cube of fixture g11 (stored as its 100 x 100 coarse cube; gen_golden.py builds it with the
    same two statements): repeat 6 x 6, add the fixed dither, NaN / nodata at the listed positions."""
    coarse = g["cube_coarse"]
    C = coarse.shape[0]
    ci, ii, jj = np.meshgrid(np.arange(C), np.arange(600), np.arange(600), indexing="ij")
    cube = np.repeat(np.repeat(coarse, 6, axis=1), 6, axis=2).astype(np.float32) + ((7 * ii + 13 * jj + 5 * ci) % 17 - 8).astype(np.float32)
    for c_, i_, j_ in g["cube_nan"]:
        cube[c_, i_, j_] = np.nan
    for c_, i_, j_ in g["cube_nd"]:
        cube[c_, i_, j_] = g["nodata"]
    return cube
