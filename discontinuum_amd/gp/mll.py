"""Exact marginal log likelihood on the MI355X backend.

``ExactMarginalLogLikelihood(likelihood, model)(engine_state)`` returns what gpytorch's class of the
same name returns at ``src/discontinuum/engines/gpytorch.py:353``:
    ( log N(y | m(X), K + Sigma) + sum_priors log p(theta) ) / n          
The Gaussian term and ALL its gradients come from one ``dgp_fit_step`` call (Gram build, blocked
Cholesky, L^-1, K^^-1, fused gradient contraction in HIP); the O(P) prior / constraint algebra is host
torch.  Non-positive-definite matrices follow linear_operator's ``psd_safe_cholesky`` policy
(SURVEY A.7): retry with diagonal jitter 1e-8, 1e-7, 1e-6 (fp64) / 1e-6, 1e-5, 1e-4 (fp32), then raise
``NotPSDError`` -- which the fit loop counts like the reference does (engines/gpytorch.py:352-358).
"""
from __future__ import annotations

import warnings

import torch

from .. import _lib
from .kernels import prior_closures


class NotPSDError(RuntimeError):
    pass


class _ExactGPNLL(torch.autograd.Function):
    """nll_data(theta, r, noise) = 1/2 r^T K^^-1 r + 1/2 log|K^| + n/2 log 2 pi, K^ = K(theta) + diag(noise)."""

    @staticmethod
    def forward(ctx, plan, theta, r, noise, pending=None):
        # `pending` = the (out, dr, dnoise) of a fit_step already launched with these arguments
        out, dr, dnoise = pending if pending is not None else plan.fit_step(theta, r, noise)
        host = out.to("cpu", torch.float64)  # the one device->host sync of a fit step
        info = int(host[_lib.OUT_INFO].item())
        if info != 0:
            raise NotPSDError(f"Matrix not positive definite: Cholesky pivot {info} is not positive")
        ctx.save_for_backward(host[_lib.OUT_DTHETA:_lib.OUT_DTHETA + plan.ntheta].clone(), dr, dnoise)
        ctx.theta_dtype = theta.dtype
        return host[_lib.OUT_NLL].clone()

    @staticmethod
    def backward(ctx, g):
        dtheta, dr, dnoise = ctx.saved_tensors
        need = ctx.needs_input_grad  # (plan, theta, r, noise, pending)
        gd = g.to(dr.device, dr.dtype) if (need[2] or need[3]) else None
        return (None, (dtheta * g).to(ctx.theta_dtype) if need[1] else None, dr * gd if need[2] else None,
                dnoise * gd if need[3] else None, None)


class _RowNLL(torch.autograd.Function):
    """The same data term for a prior mean / noise model whose few parameters live on the host (``shortcut``, see
    ``engines/hip.py::MeanShortcut``): the residual and noise vectors were built on the device without autograd, and
    the parameters' gradients are read off the reductions the fit step leaves in its result row
    (``DGP_OUT_SUM_DR``, ``DGP_OUT_DR_W0..``, ``DGP_OUT_SUM_DNOISE``) -- no device-side autograd, one sync."""

    @staticmethod
    def forward(ctx, plan, theta, pending, shortcut, *params):
        host = pending[0].to("cpu", torch.float64)
        info = int(host[_lib.OUT_INFO].item())
        if info != 0:
            raise NotPSDError(f"Matrix not positive definite: Cholesky pivot {info} is not positive")
        ctx.save_for_backward(host[_lib.OUT_DTHETA:_lib.OUT_DTHETA + plan.ntheta].clone())
        ctx.param_grads = shortcut.grads(host)
        ctx.dtypes = (theta.dtype,) + tuple(p.dtype for p in params)
        ctx.shapes = tuple(p.shape for p in params)
        return host[_lib.OUT_NLL].clone()

    @staticmethod
    def backward(ctx, g):
        (dtheta,) = ctx.saved_tensors
        grads = tuple((torch.as_tensor(v, dtype=torch.float64) * g).to(dt).reshape(shape)
                      for v, dt, shape in zip(ctx.param_grads, ctx.dtypes[1:], ctx.shapes))
        return (None, (dtheta * g).to(ctx.dtypes[0]), None, None) + grads


class _PredictiveMean(torch.autograd.Function):
    """K(X*, X; theta) K^^-1 r as a differentiable function of (theta, r, noise): forward is
    ``dgp_predict_mean``, backward ``dgp_mean_vjp``.  Both read the factorisation the plan holds from the
    fit step at the same hyperparameters -- which is how the reference's penalty callback is used: it runs
    inside the training iteration, right after the marginal likelihood (engines/gpytorch.py:350-373)."""

    @staticmethod
    def forward(ctx, plan, theta, r, noise, Xs):
        ctx.plan, ctx.theta_dtype = plan, theta.dtype
        ctx.save_for_backward(theta.detach(), Xs)
        return plan.predict_mean(theta, Xs)

    @staticmethod
    def backward(ctx, g):
        theta, Xs = ctx.saved_tensors
        dtheta, dr, dnoise = ctx.plan.mean_vjp(theta, Xs, g.contiguous())
        return None, dtheta.to("cpu", ctx.theta_dtype), dr, dnoise, None


def predictive_mean(plan, theta, r, noise, Xs):
    return _PredictiveMean.apply(plan, theta, r, noise, Xs)


def _with_jitter_retries(plan, first, again):
    """gpytorch's psd_safe_cholesky policy around a data-term evaluation: ``first()``, then ``again(jitter)`` thrice."""
    jitter0 = 1e-8 if plan.dtype == torch.float64 else 1e-6
    try:
        return first()
    except NotPSDError:
        for i in range(3):
            jitter = jitter0 * 10 ** i
            try:
                val = again(jitter)
            except NotPSDError:
                continue
            warnings.warn(f"A not p.d., added jitter of {jitter:.1e} to the diagonal", RuntimeWarning, stacklevel=3)
            return val
        raise


def exact_gp_nll(plan, theta, r, noise, pending=None):
    """Differentiable data term with gpytorch's jitter-retry policy. Returns a 0-dim CPU float64 tensor."""
    return _with_jitter_retries(plan, lambda: _ExactGPNLL.apply(plan, theta, r, noise, pending),
                                lambda jitter: _ExactGPNLL.apply(plan, theta, r, noise + jitter))


def exact_gp_nll_row(plan, theta, r, noise, pending, shortcut):
    """``exact_gp_nll`` for a host-side mean / noise model: gradients from the result row (``_RowNLL``)."""
    params = shortcut.params
    return _with_jitter_retries(
        plan, lambda: _RowNLL.apply(plan, theta, pending, shortcut, *params),
        lambda jitter: _RowNLL.apply(plan, theta, plan.fit_step(theta, r, noise + jitter), shortcut, *params))


class ExactMarginalLogLikelihood:
    def __init__(self, likelihood, model):
        self.likelihood = likelihood
        self.model = model
        # the module tree is fixed for the lifetime of an mll object (one fit): walk it once
        self._priors = prior_closures(model)
        if likelihood is not None and not any(likelihood is m for m in model.modules()):
            self._priors += prior_closures(likelihood)
        self._group_priors()

    def _group_priors(self):
        """Priors of one family are evaluated together: their constrained values are concatenated and the family's
        log-density is a handful of vector operations with the hyperparameters (constants during a fit) expanded per
        element -- a dozen autograd nodes per iteration instead of six per prior.  Other prior classes are evaluated
        one by one."""
        import math

        from .priors import GammaPrior, HalfNormalPrior, NormalPrior

        with torch.no_grad():
            sizes = [int(closure(mod).numel()) for _prior, closure, mod in self._priors]
        self._groups, self._loose, self._prior_const = {}, [], 0.0
        half_log_2pi = 0.5 * math.log(2.0 * math.pi)
        for kind in (GammaPrior, HalfNormalPrior, NormalPrior):
            idx = [i for i, (prior, _c, _m) in enumerate(self._priors) if type(prior) is kind]
            if not idx:
                continue

            def per_element(name):
                return torch.cat([getattr(self._priors[i][0], name).detach().to(torch.float64).reshape(1).expand(sizes[i])
                                  for i in idx])

            if kind is GammaPrior:
                a, b = per_element("concentration"), per_element("rate")
                self._prior_const += float((a * torch.log(b) - torch.lgamma(a)).sum())
                self._groups["gamma"] = (idx, a - 1.0, b)
            elif kind is HalfNormalPrior:
                scale = per_element("scale")
                self._prior_const += float((math.log(2.0) - torch.log(scale) - half_log_2pi).sum())
                self._groups["half_normal"] = (idx, 1.0 / scale, None)
            else:
                loc, scale = per_element("loc"), per_element("scale")
                self._prior_const += float((-torch.log(scale) - half_log_2pi).sum())
                self._groups["normal"] = (idx, 1.0 / scale, loc)
        grouped = {i for idx, _a, _b in self._groups.values() for i in idx}
        self._loose = [i for i in range(len(self._priors)) if i not in grouped]

    def log_prior(self):
        values = [closure(mod).reshape(-1) for _prior, closure, mod in self._priors]
        lp = torch.full((), self._prior_const, dtype=torch.float64)
        for name, (idx, p, q) in self._groups.items():
            x = values[idx[0]] if len(idx) == 1 else torch.cat([values[i] for i in idx])
            if name == "gamma":          # (a - 1) log x - b x
                lp = lp + (p * torch.log(x) - q * x).sum()
            elif name == "half_normal":  # -1/2 (x / s)^2
                lp = lp - 0.5 * ((x * p) ** 2).sum()
            else:                        # -1/2 ((x - loc) / s)^2
                lp = lp - 0.5 * (((x - q) * p) ** 2).sum()
        for i in self._loose:
            lp = lp + self._priors[i][0].log_prob(values[i]).sum()
        return lp

    def __call__(self, output, target):
        """``output`` is the engine's prior spec (plan, theta, mean on device, noise on device)."""
        n = target.shape[0]
        shortcut = getattr(output, "shortcut", None)
        if shortcut is not None:  # mean / noise parameters on the host: device vectors without autograd
            r, noise = shortcut.residual_and_noise(output.plan, target)
        else:
            r, noise = (target - output.mean).contiguous(), output.noise.contiguous()
        pending = output.plan.fit_step(output.theta, r, noise)  # asynchronous: the device starts now ...
        lp = self.log_prior()                                   # ... and the O(P) host algebra runs under it
        if shortcut is not None:
            nll = exact_gp_nll_row(output.plan, output.theta, r, noise, pending, shortcut)
        else:
            nll = exact_gp_nll(output.plan, output.theta, r, noise, pending)
        return ((-nll + lp) / n).reshape(1)  # shape (1,) like the reference's (1, n)-noise batch
