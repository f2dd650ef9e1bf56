"""Hyperparameter priors with gpytorch's log-densities.

Each prior is evaluated on the CONSTRAINED value and summed over ARD dimensions, exactly like the
``Sum_priors log p(theta)`` term that ``ExactMarginalLogLikelihood`` adds
(reference call site: ``src/discontinuum/engines/gpytorch.py:318, 353``).
"""
from __future__ import annotations

import math

import torch
from torch import nn

_LOG2PI = math.log(2.0 * math.pi)


class Prior(nn.Module):
    def log_prob(self, x):  # pragma: no cover - interface
        raise NotImplementedError


def _buffers(mod, **vals):
    """gpytorch keeps prior hyperparameters as module buffers, so they show up in ``state_dict()``
    (e.g. ``...outputscale_prior.scale``); mirror that for checkpoint-key compatibility."""
    for k, v in vals.items():
        mod.register_buffer(k, torch.as_tensor(float(v), dtype=torch.float64))


class NormalPrior(Prior):
    def __init__(self, loc, scale):
        super().__init__()
        _buffers(self, loc=loc, scale=scale)

    def log_prob(self, x):
        return -0.5 * ((x - self.loc) / self.scale) ** 2 - torch.log(self.scale) - 0.5 * _LOG2PI


class HalfNormalPrior(Prior):
    def __init__(self, scale):
        super().__init__()
        _buffers(self, scale=scale)

    def log_prob(self, x):
        return math.log(2.0) - 0.5 * (x / self.scale) ** 2 - torch.log(self.scale) - 0.5 * _LOG2PI


class GammaPrior(Prior):
    def __init__(self, concentration, rate):
        super().__init__()
        _buffers(self, concentration=concentration, rate=rate)

    def log_prob(self, x):
        a, b = self.concentration, self.rate
        return a * torch.log(b) - torch.lgamma(a) + (a - 1.0) * torch.log(x) - b * x
