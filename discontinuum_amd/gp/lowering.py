"""Lower a kernel tree onto one of the fused HIP evaluators (``csrc/dgp_models.h``).

The hot path implements exactly the two composite covariance functions the reference ships
(SURVEY.md section 8a rows a1-a9).  ``lower()`` checks the tree against those shapes and returns the
model id plus a differentiable builder of the constrained hyperparameter vector in the order the
device code expects.  A tree that is neither of them but is a sum of (optionally scaled) products of RBF / Matern /
Periodic factors -- e.g. the reference's covariance with its unused trend term switched on
(``src/loadest_gp/models/gpytorch.py:78-88``) -- lowers onto the GENERIC interpreted evaluator
(``dgp_composite_define``, ``csrc/dgp_models.h::Composite``): slower per matrix entry, same kernels and ABI otherwise.
Anything else (gates, warps, nested sums inside products) raises ``UnsupportedKernelError``.
"""
from __future__ import annotations

import warnings

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


_FACTOR_TYPES = {K.RBFKernel: 0, K.MaternKernel: 1, K.PeriodicKernel: 2}


def composite_spec(covar_module, d):
    """-> (spec ints for ``dgp_composite_define``, [modules whose constrained values make theta, in order])."""
    terms = list(covar_module.kernels) if isinstance(covar_module, K.AdditiveKernel) else [covar_module]
    _need(1 <= len(terms) <= 6, "at most 6 additive terms")
    spec, parts = [int(d), len(terms)], []
    for term in terms:
        scaled = isinstance(term, K.ScaleKernel)
        base = term.base_kernel if scaled else term
        factors = list(base.kernels) if isinstance(base, K.ProductKernel) else [base]
        _need(1 <= len(factors) <= 3, "at most 3 factors per term")
        spec += [int(scaled), len(factors)]
        if scaled:
            parts.append((term, "outputscale"))
        for fac in factors:
            _need(type(fac) in _FACTOR_TYPES, "RBF / Matern / Periodic factors")
            dims = _dims(fac)
            dims = tuple(range(d)) if dims is None else dims
            _need(1 <= len(dims) <= 6 and all(0 <= c < d for c in dims), "active_dims inside the design matrix")
            nls = fac.raw_lengthscale.shape[-1]
            _need(nls in (1, len(dims)), "one lengthscale, or one per active column")
            ard = int(nls == len(dims) and len(dims) > 1)
            kind = _FACTOR_TYPES[type(fac)]
            if kind == 2:
                _need(len(dims) == 1 and fac.raw_period_length.numel() == 1, "PeriodicKernel on one column")
            spec += [kind, int(round(2 * fac.nu)) if kind == 1 else 0, ard, len(dims), *dims]
            parts.append((fac, "lengthscale"))
            if kind == 2:
                parts.append((fac, "period_length"))
    return spec, parts


def _lower_composite(cov, d):
    import ctypes as C

    from .. import _lib

    spec, parts = composite_spec(cov, d)
    arr = (C.c_int * len(spec))(*spec)
    model = C.c_int()
    try:
        _lib.check(_lib.load().dgp_composite_define(arr, len(spec), C.byref(model)), "dgp_composite_define")
    except _lib.DGPError as e:
        raise UnsupportedKernelError(str(e)) from None

    def theta():
        return _flat(*[getattr(mod, attr) for mod, attr in parts])

    return f"composite:{model.value}", theta


_WARNED = set()


def lower(covar_module, d):
    """-> (model name for ``backend.GPPlan``, zero-argument callable returning theta (P,) float64 with grad)."""
    first = covar_module.kernels[0] if isinstance(covar_module, K.AdditiveKernel) else None
    try:
        _need(first is not None, "a sum of scaled kernels")
        if isinstance(first, K.ProductKernel) and any(isinstance(k, K.SigmoidKernel) for k in first.kernels):
            return _lower_rating(covar_module, d)
        return _lower_loadest(covar_module, d)
    except UnsupportedKernelError as fused:
        try:
            name, theta = _lower_composite(covar_module, d)
            if name not in _WARNED:  # once per structure: the interpreted evaluator is several times slower per entry
                _WARNED.add(name)
                warnings.warn(f"covariance lowered to the generic interpreted evaluator {name!r} -- same kernels and C ABI, "
                              f"slower Gram / gradient assembly -- because it is not one of the fused models: {fused}",
                              RuntimeWarning, stacklevel=2)
            return name, theta
        except UnsupportedKernelError as generic:
            raise UnsupportedKernelError(f"{fused}; and not a generic composite either: {generic}") from None
