"""The N>1 path on CPU: gloo processes (world size 2, and 8 - the size of the driver's scaling run) exchange Vandermonde moments exactly as the GPU
ranks do over RCCL (s2_emit.fusion.exchange_moments) and must fit identical polynomials, equal to a
single-process fit of the concatenated tiles.  No GPU: the moments are formed with NumPy here, the
solve is the library's host twin (same C code as the device solve)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

DEG = 3
NB = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tile(rank):
    rng = np.random.default_rng(100 + rank)
    x = (rng.random((NB, 5000)) * 0.6).astype(np.float32)
    y = np.clip(0.8 * x.astype(np.float64) ** 1.1 + 0.05 * rank + 0.01 * rng.standard_normal(x.shape), 0, 1).astype(np.float32)
    return x, y


def _moments(x, y):
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    S = [np.sum(xd ** k, axis=1) for k in range(2 * DEG + 1)]
    T = [np.sum(xd ** j * yd, axis=1) for j in range(DEG + 1)]
    return np.stack(S + T, axis=1)


def _worker(rank, port, mode, q, world):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "hyperspectral_super-resolution_amd"))
    import torch
    import torch.distributed as dist
    from s2_emit import _engine as eng
    from s2_emit.fusion import exchange_moments
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y = _tile(rank)
        mom = torch.from_numpy(_moments(x, y))

        def solver(m):
            return torch.from_numpy(eng.poly_solve_host(m.numpy(), DEG, 50))

        mom_out, coeffs = exchange_moments(mom, solver, None, mode)
        q.put((rank, mom_out.numpy().copy(), coeffs.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("allreduce", 2), ("broadcast", 2), ("local", 2), ("allreduce", 8), ("broadcast", 8)])
def test_exchange_moments_world(mode, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, mode, q, world)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r, m, c = q.get(timeout=240)
        res[r] = (m, c)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from s2_emit import _engine as eng
    tiles = [_tile(r) for r in range(world)]
    if mode == "local":
        for r in range(world):
            ref = np.stack([np.polyfit(tiles[r][0][b].astype(np.float64), tiles[r][1][b].astype(np.float64), DEG) for b in range(NB)])
            np.testing.assert_allclose(res[r][1], ref, rtol=1e-7, atol=1e-9)
        assert not np.array_equal(res[0][1], res[1][1])
        return
    # every rank holds bit-identical coefficients
    for r in range(1, world):
        np.testing.assert_array_equal(res[0][1], res[r][1])
    xa = np.concatenate([t[0] for t in tiles], axis=1).astype(np.float64)
    ya = np.concatenate([t[1] for t in tiles], axis=1).astype(np.float64)
    ref = np.stack([np.polyfit(xa[b], ya[b], DEG) for b in range(NB)])
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-7, atol=1e-9)
    total = _moments(*tiles[0])
    for r in range(1, world):
        total = total + _moments(*tiles[r])
    if world == 2:
        # two addends: the global moments equal the tile-ordered sum of the per-tile moments bitwise, and the
        # coefficients equal a single-process solve of that sum
        np.testing.assert_array_equal(res[0][0], total)
        np.testing.assert_array_equal(res[0][1], eng.poly_solve_host(total, DEG, 50))
    else:
        # eight addends: the collective's own (fixed) association order, equal to the tile-ordered sum to rounding
        np.testing.assert_allclose(res[0][0], total, rtol=1e-14)
