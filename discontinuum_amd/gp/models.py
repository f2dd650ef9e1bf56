"""``ExactGP`` container: training data + likelihood + the user's mean / covariance modules."""
from __future__ import annotations

from torch import nn


class ExactGP(nn.Module):
    """Shape of ``gpytorch.models.ExactGP``: subclasses define ``mean_module``, ``covar_module`` and
    ``prior_mean(x)`` (the mean half of the reference's ``forward``); the covariance half is never
    evaluated in Python -- it is lowered to the fused HIP evaluator."""

    def __init__(self, train_x, train_y, likelihood):
        super().__init__()
        self.train_inputs = (train_x,)
        self.train_targets = train_y
        self.likelihood = likelihood

    def set_train_data(self, inputs=None, targets=None, strict=True):
        if inputs is not None:
            if strict and inputs.shape != self.train_inputs[0].shape:
                raise RuntimeError("Cannot modify shape of inputs (expected strict=False)")
            self.train_inputs = (inputs,)
        if targets is not None:
            if strict and targets.shape != self.train_targets.shape:
                raise RuntimeError("Cannot modify shape of targets (expected strict=False)")
            self.train_targets = targets

    def prior_mean(self, x):
        return self.mean_module(x)
