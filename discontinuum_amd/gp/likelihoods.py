"""Gaussian likelihood with fixed heteroskedastic noise (+ optional learned homoskedastic part).

Semantics of ``gpytorch.likelihoods.FixedNoiseGaussianLikelihood`` as the reference uses it
(``src/loadest_gp/models/gpytorch.py:50-54``, ``src/rating_gp/models/gpytorch.py:71-75``).
"""
from __future__ import annotations

import warnings

import torch
from torch import nn

from .constraints import GreaterThan


class _HomoskedasticNoise(nn.Module):
    def __init__(self, noise_prior=None, noise_constraint=None):
        super().__init__()
        self.raw_noise = nn.Parameter(torch.zeros(1, dtype=torch.float64))
        self.raw_noise_constraint = noise_constraint or GreaterThan(1e-4)
        self._priors = {}
        if noise_prior is not None:
            self.add_module("noise_prior", noise_prior)
            self._priors["noise_prior"] = (noise_prior, lambda m: m.noise)

    @property
    def noise(self):
        return self.raw_noise_constraint.transform(self.raw_noise)


class FixedNoiseGaussianLikelihood(nn.Module):
    def __init__(self, noise, learn_additional_noise=False, noise_prior=None):
        super().__init__()
        self.noise = torch.as_tensor(noise).detach().reshape(-1).to(torch.float64)  # (1, n) and (n,) alike
        self.second_noise_covar = _HomoskedasticNoise(noise_prior=noise_prior) if learn_additional_noise else None

    @property
    def second_noise(self):
        if self.second_noise_covar is None:
            return None
        return self.second_noise_covar.noise

    def train_noise_fixed(self, device, dtype):
        """The fixed part of the training noise on the device (a buffer: its device copy is kept across iterations
        instead of re-uploading n values)."""
        key = (str(device), dtype, self.noise.data_ptr(), self.noise._version)
        if getattr(self, "_noise_dev_key", None) != key:
            self._noise_dev, self._noise_dev_key = self.noise.to(device, dtype), key
        return self._noise_dev

    def train_noise(self, device, dtype):
        """Diagonal of Sigma for the n training points (differentiable w.r.t. second_noise)."""
        sigma = self.train_noise_fixed(device, dtype)
        if self.second_noise_covar is not None:
            sigma = sigma + self.second_noise.to(device, dtype)
        return sigma

    def predictive_noise(self, m, device, dtype):
        """What ``likelihood(model(x))`` adds to the latent variance at m test points: the fixed part
        only when m equals the training size (else a zero no-op -- with gpytorch's warning when there is no learned
        second noise term either), plus second_noise."""
        add = torch.zeros(m, device=device, dtype=dtype)
        if m == self.noise.numel():
            add = add + self.noise.to(device, dtype)
        elif self.second_noise_covar is None:  # gpytorch warns only when there is no second noise term either
            warnings.warn(
                "You have passed data through a FixedNoiseGaussianLikelihood that did not match the size "
                "of the fixed noise, *and* you did not specify noise. This is treated as a no-op.",
                stacklevel=2,
            )
        if self.second_noise_covar is not None:
            add = add + self.second_noise.detach().to(device, dtype)
        return add
