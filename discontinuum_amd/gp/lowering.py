"""Lower a kernel tree onto one of the fused HIP evaluators (``csrc/dgp_models.h``).

The hot path implements exactly the two composite covariance functions the reference ships
(SURVEY.md section 8a rows a1-a9).  ``lower()`` checks the tree against those shapes and returns the
model id plus a differentiable builder of the constrained hyperparameter vector in the order the
device code expects.  Anything else raises ``NotImplementedError`` -- there is no generic slow path.
"""
from __future__ import annotations

import torch

from . import kernels as K


class UnsupportedKernelError(NotImplementedError):
    pass


def _dims(k):
    ad = getattr(k, "active_dims", None)
    return None if ad is None else tuple(int(v) for v in ad.tolist())


def _need(cond, what):
    if not cond:
        raise UnsupportedKernelError(f"kernel structure not supported by the HIP engine: expected {what}")


def _scale(k, what):
    _need(isinstance(k, K.ScaleKernel), f"ScaleKernel for {what}")
    return k, k.base_kernel


def _matern(k, nu, dims, what, ard=None):
    _need(isinstance(k, K.MaternKernel) and k.nu == nu, f"MaternKernel(nu={nu}) for {what}")
    _need(_dims(k) == tuple(int(v) for v in dims), f"{what} on columns {tuple(dims)}")
    _need(k.raw_lengthscale.shape[-1] == (ard or 1), f"{what} with {ard or 1} lengthscale(s)")
    return k


def _product(k, n, what):
    _need(isinstance(k, K.ProductKernel) and len(k.kernels) == n, f"product of {n} kernels for {what}")
    return list(k.kernels)


def _flat(*vals):
    return torch.cat([v.reshape(-1) for v in vals])


def _lower_loadest(cov, d):
    parts = list(cov.kernels)
    _need(len(parts) == 3, "seasonal + covariates + residual")
    s0, b0 = _scale(parts[0], "seasonal")
    per, m52 = _product(b0, 2, "Periodic x Matern52 on time")
    _need(isinstance(per, K.PeriodicKernel) and _dims(per) == (0,), "PeriodicKernel on column 0")
    _matern(m52, 2.5, (0,), "seasonal Matern52")
    s1, rbf = _scale(parts[1], "covariates")
    _need(isinstance(rbf, K.RBFKernel) and _dims(rbf) == tuple(range(1, d)), "ARD RBF on columns 1..d-1")
    _need(rbf.raw_lengthscale.shape[-1] == d - 1, "RBF with d-1 lengthscales")
    s2, m32 = _scale(parts[2], "residual")
    _matern(m32, 1.5, range(d), "residual Matern32", ard=d)

    def theta():
        return _flat(s0.outputscale, per.lengthscale, per.period_length, m52.lengthscale,
                     s1.outputscale, rbf.lengthscale, s2.outputscale, m32.lengthscale)

    return "loadest", theta


def _shift_like(k, nu_time, what):
    s, b = _scale(k, what)
    ms, mt = _product(b, 2, f"{what}: Matern(stage) x Matern(time)")
    _matern(ms, 2.5, (1,), f"{what} Matern52(stage)")
    _matern(mt, nu_time, (0,), f"{what} Matern(time)")
    return s, ms, mt


def _lower_rating(cov, d):
    _need(d == 2, "two input columns (time, stage)")
    parts = list(cov.kernels)
    _need(len(parts) == 3, "gated lower + gated upper + ungated")
    sig, lw_lower = _product(parts[0], 2, "SigmoidKernel x LogWarp(lower)")
    inv, lw_upper = _product(parts[1], 2, "InvertedSigmoidKernel x LogWarp(upper)")
    lw_rest = parts[2]
    _need(isinstance(sig, K.SigmoidKernel) and _dims(sig) == (1,) and sig.a == 20, "SigmoidKernel(a=20) on stage")
    _need(isinstance(inv, K.InvertedSigmoidKernel) and inv.sigmoid_kernel is sig, "inverted gate sharing b")
    for lw in (lw_lower, lw_upper, lw_rest):
        _need(isinstance(lw, K.LogWarpKernel) and lw.dim == 1 and lw.eps == 1e-6, "LogWarpKernel(dim=1, eps=1e-6)")
    lower = lw_lower.base_kernel
    _need(isinstance(lower, K.AdditiveKernel) and len(lower.kernels) == 2, "two cov_shift kernels")
    sh = [_shift_like(k, 1.5, f"cov_shift #{i + 1}") for i, k in enumerate(lower.kernels)]
    bend = _shift_like(lw_upper.base_kernel, 2.5, "cov_bend")
    rest = lw_rest.base_kernel
    _need(isinstance(rest, K.AdditiveKernel) and len(rest.kernels) == 2, "cov_base + cov_periodic")
    sb, mb = _scale(rest.kernels[0], "cov_base")
    _matern(mb, 2.5, (1,), "cov_base Matern52(stage)")
    sp, bp = _scale(rest.kernels[1], "cov_periodic")
    per, pm = _product(bp, 2, "Periodic x Matern52 on time")
    _need(isinstance(per, K.PeriodicKernel) and _dims(per) == (0,), "PeriodicKernel on column 0")
    _matern(pm, 2.5, (0,), "cov_periodic Matern52(time)")

    def theta():
        vals = [sig.b]
        for s, ms, mt in sh + [bend]:
            vals += [s.outputscale, ms.lengthscale, mt.lengthscale]
        vals += [sb.outputscale, mb.lengthscale, sp.outputscale, per.lengthscale, per.period_length, pm.lengthscale]
        return _flat(*vals)

    return "rating", theta


def lower(covar_module, d):
    """-> (model name for ``backend.GPPlan``, zero-argument callable returning theta (P,) float64 with grad)."""
    _need(isinstance(covar_module, K.AdditiveKernel), "a sum of scaled kernels")
    first = covar_module.kernels[0]
    if isinstance(first, K.ProductKernel) and any(isinstance(k, K.SigmoidKernel) for k in first.kernels):
        return _lower_rating(covar_module, d)
    return _lower_loadest(covar_module, d)
