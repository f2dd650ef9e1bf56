"""loadest-gp model declaration -- reads like ``src/loadest_gp/models/gpytorch.py:24-128`` with the
gpytorch classes replaced by ``discontinuum_amd.gp`` and the engine by ``MarginalHIP``."""
from __future__ import annotations

import numpy as np
import torch

from .. import gp
from ..engines.base import DataMixin, ModelConfig
from ..engines.hip import MarginalHIP
from ..gp.kernels import MaternKernel, PeriodicKernel, RBFKernel, ScaleKernel
from ..gp.priors import GammaPrior, HalfNormalPrior, NormalPrior
from ..pipeline import LogStandardPipeline, TimePipeline


class LoadestDataMixin(DataMixin):
    """Column order (time, flow) -- ``src/loadest_gp/models/base.py:14-17``."""

    def build_datamanager(self, model_config: ModelConfig | None = None):
        self._build_datamanager({"time": TimePipeline, "flow": LogStandardPipeline}, model_config)


class LoadestGPMarginalHIP(LoadestDataMixin, MarginalHIP):
    """Gaussian-process LOAD ESTimation model, marginal likelihood, MI355X engine.

    (The reference also mixes in ``LoadestPlotMixin``; plotting is outside the hot path -- the mixin
    composes in the same MRO slot, before the engine.)"""

    def __init__(self, model_config: ModelConfig | None = None):
        if model_config is None:
            model_config = ModelConfig()
        super().__init__(model_config=model_config)
        self.build_datamanager(model_config)

    def build_model(self, X, y):
        noise = 0.1 ** 2 * torch.ones(y.shape[0], dtype=y.dtype).reshape(1, -1)
        self.likelihood = gp.likelihoods.FixedNoiseGaussianLikelihood(noise=noise, learn_additional_noise=False)
        return ExactGPModel(X, y, self.likelihood)


class ExactGPModel(gp.ExactGP):
    def __init__(self, train_x, train_y, likelihood):
        super().__init__(train_x, train_y, likelihood)
        n_d = train_x.shape[1]
        self.dims = np.arange(n_d)
        self.time_dim = [self.dims[0]]
        self.cov_dims = self.dims[1:]
        self.mean_module = gp.means.ConstantMean()
        self.covar_module = self.cov_seasonal() + self.cov_covariates() + self.cov_residual()

    def cov_seasonal(self):
        eta = HalfNormalPrior(scale=1)
        period = NormalPrior(loc=1, scale=0.01)
        return ScaleKernel(
            PeriodicKernel(period_length_prior=period, active_dims=self.time_dim)
            * MaternKernel(nu=2.5, active_dims=self.time_dim),
            outputscale_prior=eta,
        )

    def cov_covariates(self):
        eta = HalfNormalPrior(scale=2)
        ls = GammaPrior(concentration=2, rate=3)
        return ScaleKernel(
            RBFKernel(ard_num_dims=self.cov_dims.shape[0], lengthscale_prior=ls, active_dims=self.cov_dims),
            outputscale_prior=eta,
        )

    def cov_residual(self):
        eta = HalfNormalPrior(scale=0.2)
        ls = GammaPrior(concentration=2, rate=10)
        return ScaleKernel(
            MaternKernel(ard_num_dims=self.dims.shape[0], nu=1.5, active_dims=self.dims, lengthscale_prior=ls),
            outputscale_prior=eta,
        )
