"""Multivariate polynomial-ridge fusion S2 -> EMIT on MI355X (variant a9 of the path).

Mirrors the notebook-resident pipeline of the reference, legacy_notebooks/Spectral_matching.ipynb:
``flatten_pixels`` (raw lines 108-126), ``logit`` / ``sigmoid`` (174-181), the scikit-learn pipeline
``StandardScaler -> PolynomialFeatures(degree, include_bias=False) -> Ridge(alpha)`` (475-490) and
``predict_cube_logit`` (192-213).  The fit is the float64 normal-equation form of that pipeline
(centre features and targets, solve (Phi^T Phi + alpha I) W = Phi^T Y): the Gram contraction runs on the
float64 matrix cores (``hsr_gram_f64``), the 285 x 285 ridge system by the library's own one-workgroup blocked
Cholesky (``hsr_chol_solve_f64``), and
the prediction is one fused expand + float32-MFMA + sigmoid kernel (``hsr_polyfeat_predict``).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _native as nat
from ._engine import _ptr, _stream


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def subsample_bands_evenly(num_bands_total: int, num_keep: int = 32) -> np.ndarray:
    """Evenly spaced band indices in [0, num_bands_total) (notebook raw lines 83-97)."""
    idx = np.unique(np.linspace(0, num_bands_total - 1, num_keep).round().astype(int))
    while len(idx) < num_keep:
        missing = num_keep - len(idx)
        add = []
        for i in range(len(idx) - 1):
            if len(add) >= missing:
                break
            add.append(int((idx[i] + idx[i + 1]) // 2))
        idx = np.unique(np.concatenate([idx, np.array(add, dtype=int)]))
    return idx[:num_keep]


def flatten_pixels(X_bhw, Y_bhw, x_nodata=None, y_nodata=None):
    """(Bx,H,W), (By,H,W) -> X (N,Bx), Y (N,By) keeping pixels finite in both (and not nodata)."""
    Bx, H, W = X_bhw.shape
    By, Hy, Wy = Y_bhw.shape
    assert (H, W) == (Hy, Wy)
    X = X_bhw.reshape(Bx, -1).T
    Y = Y_bhw.reshape(By, -1).T
    mask = np.isfinite(X).all(axis=1) & np.isfinite(Y).all(axis=1)
    if x_nodata is not None:
        mask &= ~(np.isclose(X, x_nodata).any(axis=1))
    if y_nodata is not None:
        mask &= ~(np.isclose(Y, y_nodata).any(axis=1))
    return X[mask], Y[mask]


def logit(x, eps: float = 1e-4):
    x = np.clip(x, eps, 1.0 - eps)
    return np.log(x / (1.0 - x))


def sigmoid(z):
    z = np.clip(z, -50, 50)
    return 1.0 / (1.0 + np.exp(-z))


class PolyRidge:
    """StandardScaler -> PolynomialFeatures(degree, no bias) -> Ridge(alpha, intercept) on the GPU."""

    def __init__(self, degree: int = 3, alpha: float = 1.0):
        self.degree, self.alpha = int(degree), float(alpha)
        self.n_in = self.n_feat = self.n_targets = 0
        self._dev = {}
        self._fit64 = None     # float64 device tensors of the last fit: (mean, scale, W (nf, T), intercept)
        self._host = None      # their host copies, made on first access of mean_ / scale_ / coef_ / intercept_

    # scikit-learn's attribute names, float64 NumPy arrays.  They are copied from the device on first use, so that
    # fit() itself never synchronises with the host (it is ~0.4 ms of enqueued GPU work).
    def _host_copy(self):
        if self._fit64 is None:
            return (None, None, None, None)
        if self._host is None:
            mean, scale, Wm, b = self._fit64
            self._host = (mean.cpu().numpy(), scale.cpu().numpy(), Wm.t().contiguous().cpu().numpy(), b.cpu().numpy())
        return self._host

    mean_ = property(lambda self: self._host_copy()[0])
    scale_ = property(lambda self: self._host_copy()[1])
    coef_ = property(lambda self: self._host_copy()[2])
    intercept_ = property(lambda self: self._host_copy()[3])

    # ---- fit --------------------------------------------------------------------------------------
    # The fit is a sum over pixels twice over (scaler statistics, then the Gram matrix), so it shards by
    # pixels exactly like the per-band fit (variant a9): ranks exchange (count, mean, M2) of
    # their inputs - combined in rank order with the pairwise update of Chan et al., identically on every
    # rank - and then all-reduce the [1|Phi]^T [1|Phi|Y] Gram (<= 288 x 608 float64 = 1.4 MB); every rank
    # solves the same system and holds a bit-identical model.  The stages are separate methods so that the
    # exchange can also be driven by hand (tests emulate ranks with shards on one GPU).
    @staticmethod
    def _to_dev(X, Y):
        torch = nat.require_gpu()
        Xd = (X if _is_torch(X) else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))).to("cuda", torch.float32).contiguous()
        Yd = (Y if _is_torch(Y) else torch.from_numpy(np.ascontiguousarray(Y))).to("cuda", torch.float64).contiguous()
        if Yd.dim() == 1:
            Yd = Yd[:, None]
        return Xd, Yd

    @staticmethod
    def _stats_dev(Xd):
        """(stats, mean, scale) of this shard's inputs on the device: stats = (1 + 2 n_in,) float64 [n, mean..., M2...],
        mean / scale = StandardScaler's for THIS shard alone (hsr_ridge_stats: two launches)."""
        torch = nat.require_gpu()
        lib = nat.load()
        n, n_in = Xd.shape
        dev = Xd.device
        if n == 0:
            z = torch.zeros(1 + 2 * n_in, dtype=torch.float64, device=dev)
            return z, z[1:1 + n_in].clone(), torch.ones(n_in, dtype=torch.float64, device=dev)
        buf = torch.empty(1 + 4 * n_in + lib.hsr_ridge_stats_work_bytes(n_in) // 8, dtype=torch.float64, device=dev)
        stats, mean, scale, work = buf[:1 + 2 * n_in], buf[1 + 2 * n_in:1 + 3 * n_in], buf[1 + 3 * n_in:1 + 4 * n_in], buf[1 + 4 * n_in:]
        nat.check(lib.hsr_ridge_stats(_ptr(Xd), Xd.stride(0), Xd.stride(1) if n_in > 1 else 1, n, n_in, _ptr(work),
                                      _ptr(stats), _ptr(mean), _ptr(scale), _stream(torch, Xd)), "hsr_ridge_stats")
        return stats, mean, scale

    @staticmethod
    def local_stats(Xd):
        """(1 + 2 n_in,) float64 device tensor [n, mean..., M2...] of this shard's inputs."""
        return PolyRidge._stats_dev(Xd)[0]

    @staticmethod
    def combine_stats(stats):
        """(world, 1 + 2 n_in) stacked local_stats -> (mean, scale) of the union; fixed rank order."""
        torch = nat.require_gpu()
        n_in = (stats.shape[1] - 1) // 2
        n = stats[0, 0].clone()
        mean = stats[0, 1:1 + n_in].clone()
        m2 = stats[0, 1 + n_in:].clone()
        for r in range(1, stats.shape[0]):
            nb_, mb, m2b = stats[r, 0], stats[r, 1:1 + n_in], stats[r, 1 + n_in:]
            tot = n + nb_
            delta = mb - mean
            safe = torch.where(tot > 0, tot, torch.ones_like(tot))
            mean = mean + delta * (nb_ / safe)
            m2 = m2 + m2b + delta * delta * (n * nb_ / safe)
            n = tot
        scale = (m2 / n).sqrt()
        scale = torch.where(scale == 0, torch.ones_like(scale), scale)          # StandardScaler: zero variance -> 1
        return mean, scale

    def local_gram(self, Xd, Yd, mean, scale):
        """[1 | Phi(z)]^T [1 | Phi(z) | Y] of this shard on the float64 matrix cores -> (na, na + tp) device tensor."""
        torch = nat.require_gpu()
        lib = nat.load()
        n, n_in = Xd.shape
        T = Yd.shape[1]
        nf = lib.hsr_polyfeat_count(n_in, self.degree)
        if nf <= 0:
            raise ValueError(f"unsupported polynomial features: n_in={n_in}, degree={self.degree}")
        nat.check(lib.hsr_polyfeat_prepare(n_in, self.degree), "hsr_polyfeat_prepare")
        na = (nf + 1 + 15) // 16 * 16                                          # [1 | features] padded
        tp = (T + 15) // 16 * 16
        if n == 0:
            return torch.zeros((na, na + tp), dtype=torch.float64, device=Xd.device)
        G = torch.empty((na, na + tp), dtype=torch.float64, device=Xd.device)   # hsr_gram_f64 writes every element
        Q = torch.empty((n, na + tp), dtype=torch.float64, device=Xd.device)    # [P | Y | 0]; expand fills [0, na)
        Q[:, na:na + T] = Yd
        if tp > T:
            Q[:, na + T:].zero_()
        nat.check(lib.hsr_polyfeat_expand_f64(_ptr(Xd), Xd.stride(0), Xd.stride(1) if n_in > 1 else 1, _ptr(mean),
                                              _ptr(scale), n, n_in, self.degree, _ptr(Q), Q.stride(0), na,
                                              _stream(torch, Xd)), "hsr_polyfeat_expand_f64")
        work = torch.empty(max(1, lib.hsr_gram_work_bytes(na, na + tp, n) // 8), dtype=torch.float64, device=Xd.device)
        nat.check(lib.hsr_gram_f64(_ptr(Q), Q.stride(0), na, _ptr(Q), Q.stride(0), na + tp, n, _ptr(work), _ptr(G),
                                   G.stride(0), _stream(torch, Q)), "hsr_gram_f64")
        return G

    def solve_gram(self, G, mean, scale, n_in: int, T: int):
        """Centre, add alpha I, Cholesky-solve; stores the model (host float64 copies + device float32 operands)."""
        torch = nat.require_gpu()
        lib = nat.load()
        nf = lib.hsr_polyfeat_count(n_in, self.degree)
        na = (nf + 1 + 15) // 16 * 16
        # (Phi_c^T Phi_c + alpha I) W = Phi_c^T (Y - ybar) by the library's Cholesky (csrc/hsr_chol.hip), the system padded to
        # a multiple of 32 with an identity block and zero right-hand-side rows, which leaves the solution untouched;
        # assembly and model read-out are one launch each (hsr_ridge_assemble / hsr_ridge_finish)
        npad = (nf + 31) // 32 * 32
        dev = G.device
        Gp = torch.empty((npad, npad), dtype=torch.float64, device=dev)
        Bp = torch.empty((npad, T), dtype=torch.float64, device=dev)
        self._chol_info = torch.empty(1, dtype=torch.int32, device=dev)        # 0, or the first non-positive pivot
        st = _stream(torch, G)
        nat.check(lib.hsr_ridge_assemble(_ptr(G), G.stride(0), na, nf, T, float(self.alpha), _ptr(Gp), npad, _ptr(Bp),
                                         Bp.stride(0), _ptr(self._chol_info), st), "hsr_ridge_assemble")
        cwork = torch.empty(lib.hsr_chol_work_bytes(npad) // 8, dtype=torch.float64, device=dev)
        nat.check(lib.hsr_chol_solve_f64(_ptr(Gp), Gp.stride(0), npad, _ptr(Bp), Bp.stride(0), T, _ptr(cwork),
                                         _ptr(self._chol_info), st), "hsr_chol_solve_f64")
        Wm = Bp[:nf]                                     # (nf, T)
        kpad = (nf + 1) // 2 * 2
        b = torch.empty(T, dtype=torch.float64, device=dev)
        f32 = torch.empty(kpad * T + T + 2 * n_in, dtype=torch.float32, device=dev)
        Wf, b32 = f32[:kpad * T].view(kpad, T), f32[kpad * T:kpad * T + T]
        mean32, inv32 = f32[kpad * T + T:kpad * T + T + n_in], f32[kpad * T + T + n_in:]
        nat.check(lib.hsr_ridge_finish(_ptr(G), na, nf, T, _ptr(Wm), Bp.stride(0), _ptr(mean), _ptr(scale), n_in, kpad,
                                       _ptr(b), _ptr(b32), _ptr(Wf), _ptr(mean32), _ptr(inv32), st), "hsr_ridge_finish")
        self.n_in, self.n_feat, self.n_targets = n_in, nf, T
        self._fit64, self._host = (mean, scale, Wm, b), None
        self._dev = dict(W=Wf, b=b32, mean=mean32, inv=inv32)
        return self

    def fit(self, X, Y, group=None, distributed: Optional[bool] = None):
        """X (N, n_in), Y (N, T) - NumPy or GPU tensors; rows must be finite (see flatten_pixels).
        Inside an initialised torch.distributed job of more than one rank (or distributed=True) X, Y are this
        rank's pixels and the fit is over the union of all ranks' pixels (two small collectives, see above)."""
        torch = nat.require_gpu()
        Xd, Yd = self._to_dev(X, Y)
        if distributed is None:
            import torch.distributed as dist
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        stats, mean, scale = self._stats_dev(Xd)
        if distributed:
            import torch.distributed as dist
            world = dist.get_world_size(group)
            gathered = [torch.empty_like(stats) for _ in range(world)]
            dist.all_gather(gathered, stats, group=group)
            mean, scale = self.combine_stats(torch.stack(gathered))
        G = self.local_gram(Xd, Yd, mean, scale)
        if distributed:
            dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
        return self.solve_gram(G, mean, scale, Xd.shape[1], Yd.shape[1])

    @classmethod
    def from_params(cls, mean, scale, coef, intercept, degree: int = 3, alpha: float = 1.0):
        """A model from given parameters - e.g. a scikit-learn pipeline fitted elsewhere (``scaler.mean_``, ``scaler.scale_``,
        ``ridge.coef_`` (T, n_features), ``ridge.intercept_``) - ready for predict() / predict_cube() on the GPU."""
        torch = nat.require_gpu()
        lib = nat.load()
        mean = np.ascontiguousarray(mean, dtype=np.float64).reshape(-1)
        scale = np.ascontiguousarray(scale, dtype=np.float64).reshape(-1)
        coef = np.atleast_2d(np.asarray(coef, dtype=np.float64))
        b = np.ascontiguousarray(intercept, dtype=np.float64).reshape(-1)
        n_in, T = mean.shape[0], coef.shape[0]
        nf = lib.hsr_polyfeat_count(n_in, int(degree))
        if nf <= 0 or coef.shape[1] != nf or scale.shape[0] != n_in or b.shape[0] != T:
            raise ValueError(f"parameters do not describe a degree-{degree} model of {n_in} inputs: coef {coef.shape}, "
                             f"expected ({T}, {nf})")
        m = cls(degree, alpha)
        kpad = (nf + 1) // 2 * 2
        W = np.zeros((kpad, T), dtype=np.float32)
        W[:nf] = coef.T
        dev = torch.device("cuda", torch.cuda.current_device())
        to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
        m.n_in, m.n_feat, m.n_targets = n_in, nf, T
        m._fit64 = (to(mean, torch.float64), to(scale, torch.float64), to(coef.T, torch.float64), to(b, torch.float64))
        m._dev = dict(W=to(W, torch.float32), b=to(b, torch.float32), mean=to(mean, torch.float32), inv=to(1.0 / scale, torch.float32))
        return m

    # ---- predict ------------------------------------------------------------------------------------
    def _predict_dev(self, x, x_ps: int, x_cs: int, npix: int, activation: int, nan_bad: bool = False, nodata=None):
        torch = nat.require_gpu()
        lib = nat.load()
        if not self._dev:
            raise RuntimeError("PolyRidge is not fitted")
        nat.check(lib.hsr_polyfeat_prepare(self.n_in, self.degree), "hsr_polyfeat_prepare")
        d = self._dev
        out = torch.empty((self.n_targets, npix), dtype=torch.float32, device=x.device)
        nat.check(lib.hsr_polyfeat_predict_cube(_ptr(x), x_ps, x_cs, _ptr(d["mean"]), _ptr(d["inv"]), npix, self.n_in,
                                                self.degree, _ptr(d["W"]), d["W"].stride(0), _ptr(d["b"]), self.n_targets,
                                                activation, 1 if nan_bad else 0, 0.0 if nodata is None else float(nodata),
                                                0 if nodata is None else 1, _ptr(out), out.stride(0), _stream(torch, out)),
                  "hsr_polyfeat_predict_cube")
        return out

    def predict(self, X):
        """X (N, n_in) -> (N, T) float32 raw model output (logit space in the notebook's use)."""
        torch = nat.require_gpu()
        Xd = (X if _is_torch(X) else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))).to("cuda", torch.float32).contiguous()
        out = self._predict_dev(Xd, Xd.stride(0), 1, Xd.shape[0], 0).t()
        return out if _is_torch(X) else out.cpu().numpy()

    def predict_cube(self, X_bhw, nodata=None):
        """(n_in, H, W) S2 cube -> (T, H, W) float32 = sigmoid(clip(model, +-50)); pixels with a non-finite
        (or nodata) input come out NaN, as predict_cube_logit leaves them (notebook raw lines 197-203)."""
        torch = nat.require_gpu()
        Xd = (X_bhw if _is_torch(X_bhw) else torch.from_numpy(np.ascontiguousarray(X_bhw, dtype=np.float32)))
        Xd = Xd.to("cuda", torch.float32).contiguous()
        Cc, H, W = Xd.shape
        # the rule for unusable pixels (non-finite or nodata input -> NaN in every target) is part of the kernel's epilogue
        out = self._predict_dev(Xd, 1, H * W, H * W, 1, nan_bad=True, nodata=nodata)
        out = out.reshape(self.n_targets, H, W)
        return out if _is_torch(X_bhw) else out.cpu().numpy()


def predict_cube_logit(model: PolyRidge, X_bhw, nodata=None, batch_pixels: int = 200_000):
    """Notebook signature (raw lines 192-213); ``batch_pixels`` is accepted and ignored (one fused launch)."""
    return model.predict_cube(X_bhw, nodata=nodata)
