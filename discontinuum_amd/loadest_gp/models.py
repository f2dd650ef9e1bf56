"""loadest-gp (concentration from time and streamflow) on the MI355X engine.

The model is the reference's (``src/loadest_gp/models/gpytorch.py:24-128``): constant mean, fixed observation noise
0.1^2 in model space, and a covariance of three scaled terms over the design-matrix columns (time first):

    seasonal    sigma^2 * Periodic(t) * Matern52(t)          sigma^2 ~ HalfNormal(1),   period ~ N(1, 0.01)
    covariates  sigma^2 * RBF_ARD(x_1 .. x_{d-1})            sigma^2 ~ HalfNormal(2),   l ~ Gamma(2, 3)
    residual    sigma^2 * Matern32_ARD(t, x_1 .. x_{d-1})    sigma^2 ~ HalfNormal(0.2), l ~ Gamma(2, 10)

The module tree (and so every ``state_dict`` key) is the reference's: ``covar_module`` is the sum of three
``ScaleKernel``s in that order, which is also what ``gp.lowering`` recognises and maps onto the fused device kernel.
"""
from __future__ import annotations

import torch

from .. import gp
from ..engines.base import DataMixin, ModelConfig
from ..engines.hip import MarginalHIP
from ..gp import kernels as K
from ..gp.priors import GammaPrior, HalfNormalPrior, NormalPrior
from ..pipeline import LogStandardPipeline, TimePipeline

MODEL_SPACE_NOISE = 0.1 ** 2  # fixed observation variance (gpytorch.py:51-54)


def loadest_covariance(n_columns: int):
    """seasonal + covariates + residual for a design matrix with ``n_columns`` columns, time in column 0."""
    time, others, everything = [0], list(range(1, n_columns)), list(range(n_columns))
    seasonal = K.ScaleKernel(
        K.PeriodicKernel(active_dims=time, period_length_prior=NormalPrior(loc=1, scale=0.01))
        * K.MaternKernel(nu=2.5, active_dims=time),
        outputscale_prior=HalfNormalPrior(scale=1))
    covariates = K.ScaleKernel(
        K.RBFKernel(active_dims=others, ard_num_dims=len(others), lengthscale_prior=GammaPrior(concentration=2, rate=3)),
        outputscale_prior=HalfNormalPrior(scale=2))
    residual = K.ScaleKernel(
        K.MaternKernel(nu=1.5, active_dims=everything, ard_num_dims=n_columns,
                       lengthscale_prior=GammaPrior(concentration=2, rate=10)),
        outputscale_prior=HalfNormalPrior(scale=0.2))
    return seasonal + covariates + residual


class ExactGPModel(gp.ExactGP):
    def __init__(self, train_x, train_y, likelihood):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = gp.means.ConstantMean()
        self.covar_module = loadest_covariance(train_x.shape[1])


class LoadestDataMixin(DataMixin):
    """Design matrix columns (time, flow) -- ``src/loadest_gp/models/base.py:14-17``."""

    def build_datamanager(self, model_config: ModelConfig | None = None):
        self._build_datamanager({"time": TimePipeline, "flow": LogStandardPipeline}, model_config)


class LoadestGPMarginalHIP(LoadestDataMixin, MarginalHIP):
    """LOAD ESTimation as an exact GP (marginal likelihood) on the MI355X engine.  The reference class also mixes in
    its plotting helpers; they sit outside the hot path and would compose in the same MRO slot, before the engine."""

    def __init__(self, model_config: ModelConfig | None = None):
        config = model_config or ModelConfig()
        super().__init__(model_config=config)
        self.build_datamanager(config)

    def build_model(self, X, y):
        fixed = torch.full((1, y.shape[0]), MODEL_SPACE_NOISE, dtype=y.dtype)
        self.likelihood = gp.likelihoods.FixedNoiseGaussianLikelihood(noise=fixed, learn_additional_noise=False)
        return ExactGPModel(X, y, self.likelihood)
