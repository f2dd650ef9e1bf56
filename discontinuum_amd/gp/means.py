"""Mean functions (host-side torch; O(n) work, evaluated on the GPU tensors with autograd)."""
from __future__ import annotations

import torch
from torch import nn


class Mean(nn.Module):
    pass


class ConstantMean(Mean):
    """gpytorch.means.ConstantMean: m(x) = c, raw_constant initialised to 0."""

    def __init__(self):
        super().__init__()
        self.raw_constant = nn.Parameter(torch.zeros((), dtype=torch.float64))

    @property
    def constant(self):
        return self.raw_constant

    def forward(self, x):
        return self.raw_constant.to(x.device, x.dtype).expand(x.shape[0])


class NoOpMean(Mean):
    """``NoOpMean`` of the reference engine (src/discontinuum/engines/gpytorch.py:31-33)."""

    def forward(self, x):
        return x.squeeze(-1)
