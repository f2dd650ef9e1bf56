"""Declarative kernel tree (the "kernel-spec IR") mirroring the gpytorch classes the reference composes.

The model packages compose these classes into the reference's covariance functions
(``src/loadest_gp/models/gpytorch.py:61-128``, ``src/rating_gp/models/gpytorch.py:205-372``,
``src/rating_gp/models/kernels.py:242-382``): same class names, constructor arguments, ``+`` / ``*``
composition, raw-parameter names and shapes, constraints and priors -- hence the same ``state_dict`` keys.  The tree holds NO arithmetic:
``discontinuum_amd.gp.lowering`` maps it onto one of the fused HIP evaluators
(``csrc/dgp_models.h``) and fails loudly for any structure the hardware path does not implement.
"""
from __future__ import annotations

import torch
from torch import nn

from .constraints import Interval, Positive
from .priors import NormalPrior

DT = torch.float64


class Kernel(nn.Module):
    has_lengthscale = False

    def __init__(self, active_dims=None, ard_num_dims=None, lengthscale_prior=None, lengthscale_constraint=None):
        super().__init__()
        if active_dims is not None:  # gpytorch registers active_dims as a buffer (it is part of state_dict)
            self.register_buffer("active_dims", torch.as_tensor([int(v) for v in active_dims], dtype=torch.long))
        else:
            self.active_dims = None
        self.ard_num_dims = ard_num_dims
        self._priors = {}
        if self.has_lengthscale:
            k = 1 if ard_num_dims is None else int(ard_num_dims)
            self.raw_lengthscale = nn.Parameter(torch.zeros(1, k, dtype=DT))
            self.raw_lengthscale_constraint = lengthscale_constraint or Positive()
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, lambda m: m.lengthscale)

    # -- gpytorch-style plumbing -----------------------------------------------------------
    def register_prior(self, name, prior, closure):
        self.add_module(name, prior)
        self._priors[name] = (prior, closure)

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale)

    def __add__(self, other):
        return AdditiveKernel(self, other)

    def __mul__(self, other):
        return ProductKernel(self, other)


def named_priors(module: nn.Module):
    """(qualified name, prior, constrained value) for every registered prior, each module once."""
    for mod_name, mod in module.named_modules():
        for name, (prior, closure) in getattr(mod, "_priors", {}).items():
            yield (f"{mod_name}.{name}" if mod_name else name), prior, closure(mod)


def prior_closures(module: nn.Module):
    """[(prior, closure, owner module)] for every registered prior: the walk of ``named_priors`` done once."""
    return [(prior, closure, mod) for _n, mod in module.named_modules()
            for prior, closure in getattr(mod, "_priors", {}).values()]


class RBFKernel(Kernel):
    has_lengthscale = True


class MaternKernel(Kernel):
    has_lengthscale = True

    def __init__(self, nu=2.5, **kwargs):
        if nu not in (0.5, 1.5, 2.5):
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu


class PeriodicKernel(Kernel):
    has_lengthscale = True

    def __init__(self, period_length_prior=None, period_length_constraint=None, **kwargs):
        super().__init__(**kwargs)
        k = 1 if self.ard_num_dims is None else int(self.ard_num_dims)
        self.raw_period_length = nn.Parameter(torch.zeros(1, k, dtype=DT))
        self.raw_period_length_constraint = period_length_constraint or Positive()
        if period_length_prior is not None:
            self.register_prior("period_length_prior", period_length_prior, lambda m: m.period_length)

    @property
    def period_length(self):
        return self.raw_period_length_constraint.transform(self.raw_period_length)


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.raw_outputscale = nn.Parameter(torch.zeros((), dtype=DT))
        self.raw_outputscale_constraint = outputscale_constraint or Positive()
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, lambda m: m.outputscale)

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)


class _Composite(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        flat = []
        for k in kernels:  # gpytorch flattens nested sums / products of the same kind
            flat.extend(k.kernels if type(k) is type(self) else [k])
        self.kernels = nn.ModuleList(flat)


class AdditiveKernel(_Composite):
    pass


class ProductKernel(_Composite):
    pass


class SigmoidKernel(Kernel):
    """Rank-one gate g(x) g(x')^T, g = 1 / (1 + exp(a (x - b))), a = 20 fixed
    (``src/rating_gp/models/kernels.py:242-319``)."""

    def __init__(self, b_constraint, b_prior=None, **kwargs):
        super().__init__(**kwargs)
        self.a = 20
        lo, hi = b_constraint.lower_bound, b_constraint.upper_bound
        init_b = lo + torch.rand(1, 1, dtype=DT) * (hi - lo)  # kernels.py:276
        self.raw_b_constraint = b_constraint
        self.raw_b = nn.Parameter(b_constraint.inverse_transform(init_b))
        self.register_prior("b_prior", NormalPrior(0, 1), lambda m: m.b)  # kernels.py:280 overrides the argument

    @property
    def b(self):
        return self.raw_b_constraint.transform(self.raw_b)


class InvertedSigmoidKernel(Kernel):
    """(1 - g(x)) (1 - g(x'))^T sharing the switch point of ``sigmoid_kernel`` (kernels.py:323-360)."""

    def __init__(self, sigmoid_kernel, active_dims=None, b_constraint=None):
        super().__init__(active_dims=active_dims)
        self.sigmoid_kernel = sigmoid_kernel

    @property
    def a(self):
        return self.sigmoid_kernel.a

    @property
    def b(self):
        return self.sigmoid_kernel.b


class LogWarpKernel(Kernel):
    """Applies log(x + eps) to one input column before the wrapped kernel (kernels.py:363-382)."""

    def __init__(self, base_kernel, dim, eps=1e-6):
        super().__init__()
        self.base_kernel = base_kernel
        self.dim = int(dim)
        self.eps = eps


__all__ = [
    "Kernel", "RBFKernel", "MaternKernel", "PeriodicKernel", "ScaleKernel", "AdditiveKernel", "ProductKernel",
    "SigmoidKernel", "InvertedSigmoidKernel", "LogWarpKernel", "Interval", "Positive", "named_priors", "prior_closures",
]
