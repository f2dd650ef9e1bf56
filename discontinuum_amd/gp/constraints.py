"""Parameter constraints with gpytorch's semantics.

The reference builds its kernels out of ``gpytorch.constraints`` objects implicitly (``Positive`` is
the default for lengthscale / outputscale / period_length) and explicitly
(``Interval(b_min, b_max)`` at ``src/rating_gp/models/gpytorch.py:228``); the learned likelihood noise
uses ``GreaterThan(1e-4)``.  Raw parameters live on the host; these transforms stay in torch so that
autograd carries d theta / d raw.
"""
from __future__ import annotations

import torch
from torch import nn


def _inv_softplus(y):
    return y + torch.log(-torch.expm1(-y))


class Interval(nn.Module):
    """theta = lo + (hi - lo) * sigmoid(raw)."""

    def __init__(self, lower_bound, upper_bound):
        super().__init__()
        self.register_buffer("lower_bound", torch.as_tensor(float(lower_bound), dtype=torch.float64))
        self.register_buffer("upper_bound", torch.as_tensor(float(upper_bound), dtype=torch.float64))

    def transform(self, raw):
        return self.lower_bound + (self.upper_bound - self.lower_bound) * torch.sigmoid(raw)

    def inverse_transform(self, value):
        u = (torch.as_tensor(value, dtype=torch.float64) - self.lower_bound) / (self.upper_bound - self.lower_bound)
        return torch.log(u) - torch.log1p(-u)


class GreaterThan(Interval):
    """theta = softplus(raw) + lower_bound."""

    def __init__(self, lower_bound):
        super().__init__(lower_bound, float("inf"))

    def transform(self, raw):
        return torch.nn.functional.softplus(raw) + self.lower_bound

    def inverse_transform(self, value):
        return _inv_softplus(torch.as_tensor(value, dtype=torch.float64) - self.lower_bound)


class Positive(GreaterThan):
    """theta = softplus(raw)."""

    def __init__(self):
        super().__init__(0.0)

    def transform(self, raw):
        return torch.nn.functional.softplus(raw)  # without GreaterThan's "+ 0": one autograd node less per value

    def inverse_transform(self, value):
        return _inv_softplus(torch.as_tensor(value, dtype=torch.float64))
