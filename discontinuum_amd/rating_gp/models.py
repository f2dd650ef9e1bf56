"""rating-gp model declaration -- reads like ``src/rating_gp/models/gpytorch.py:28-372`` with the
gpytorch classes replaced by ``discontinuum_amd.gp`` and the engine by ``MarginalHIP``."""
from __future__ import annotations

import numpy as np
import torch

from .. import gp
from ..engines.base import DataMixin, ModelConfig
from ..engines.hip import MarginalHIP
from ..gp.constraints import Interval
from ..gp.kernels import (
    InvertedSigmoidKernel,
    LogWarpKernel,
    MaternKernel,
    PeriodicKernel,
    ScaleKernel,
    SigmoidKernel,
)
from ..gp.means import NoOpMean
from ..gp.priors import GammaPrior, HalfNormalPrior, NormalPrior
from ..pipeline import TimePipeline, UnitPipeline


class PowerLawTransform(torch.nn.Module):
    """a + b log(x - c) with the reference's initialisation (gpytorch.py:28-40)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.randn(1, dtype=torch.float64))
        self.b = torch.nn.Parameter(torch.randn(1, dtype=torch.float64) + 1.3)
        self.c = torch.nn.Parameter(torch.rand(1, dtype=torch.float64))

    def clamp_c(self, stage_min):
        self.c.data = torch.clamp(self.c.data, max=float(stage_min) - 1e-6)

    def forward(self, x):
        self.clamp_c(x.min())
        a, b, c = (p.to(x.device, x.dtype) for p in (self.a, self.b, self.c))
        return a + (b * torch.log(x - c))


class RatingDataMixin(DataMixin):
    """Column order (time, stage), stage rescaled to [1, 2] -- ``src/rating_gp/models/base.py:14-17``."""

    def build_datamanager(self, model_config: ModelConfig | None = None):
        self._build_datamanager({"time": TimePipeline, "stage": UnitPipeline}, model_config)


class RatingGPMarginalHIP(RatingDataMixin, MarginalHIP):
    """Gaussian-process stage-discharge rating model, marginal likelihood, MI355X engine."""

    def __init__(self, model_config: ModelConfig | None = None):
        if model_config is None:
            model_config = ModelConfig()
        super().__init__(model_config=model_config)
        self.build_datamanager(model_config)

    def build_model(self, X, y, y_unc=None):
        noise = y_unc if y_unc is not None else 0.1 ** 2 * torch.ones(y.shape[0], dtype=y.dtype).reshape(1, -1)
        self.likelihood = gp.likelihoods.FixedNoiseGaussianLikelihood(
            noise=noise, learn_additional_noise=True, noise_prior=HalfNormalPrior(scale=0.03))
        return ExactGPModel(X, y, self.likelihood)

    def fit(self, covariates, target, target_unc=None, iterations=100, optimizer=None, learning_rate=None,
            early_stopping=False, patience=60, scheduler=True, resume=False,
            monotonic_penalty_weight: float = 0.0, grid_size: int = 64, monotonic_penalty_interval: int = 1):
        """``fit`` with the optional monotonicity penalty of the reference (gpytorch.py:81-202)."""
        common = dict(covariates=covariates, target=target, target_unc=target_unc, iterations=iterations,
                      optimizer=optimizer, learning_rate=learning_rate, early_stopping=early_stopping,
                      patience=patience, scheduler=scheduler, resume=resume)
        if monotonic_penalty_weight <= 0:
            return super().fit(**common, penalty_callback=None, penalty_weight=0.0)

        step = {"i": 0}

        def penalty_callback():
            # optionally skip iterations (the skipped ones are compensated by the interval factor below)
            step["i"] += 1
            if monotonic_penalty_interval > 1 and (step["i"] % monotonic_penalty_interval) != 0:
                return torch.zeros((), device=self._train_x.device, dtype=self.dtype)
            # random grid in model space: time uniform, stage log-uniform (denser at low stage)
            time_dim, stage_dim = 0, 1
            x_min, x_max = self.dm.X.min(axis=0), self.dm.X.max(axis=0)
            dev = self._train_x.device
            u_time = torch.rand((grid_size,), dtype=self.dtype, device=dev)
            time_grid = u_time * (x_max[time_dim] - x_min[time_dim]) + x_min[time_dim]
            eps = 1e-6
            log_lo, log_hi = float(np.log(x_min[stage_dim] + eps)), float(np.log(x_max[stage_dim] + eps))
            u_stage = torch.rand((grid_size,), dtype=self.dtype, device=dev)
            stage_grid = torch.exp(u_stage * (log_hi - log_lo) + log_lo)
            x_grid = torch.stack([time_grid, stage_grid], dim=1)
            # finite difference in stage of the posterior mean (same fd_eps as the reference); both sets of
            # points go through one differentiable predictive-mean call
            fd_eps = 1e-3
            x_plus = x_grid.clone()
            x_plus[:, stage_dim] = x_grid[:, stage_dim] + fd_eps
            mean = self._differentiable_mean(torch.cat([x_grid, x_plus], dim=0))
            d_mean_d_stage = (mean[grid_size:] - mean[:grid_size]) / fd_eps
            pen = torch.clamp(-d_mean_d_stage, min=0.0).mean()
            if monotonic_penalty_interval > 1:
                pen = pen * float(monotonic_penalty_interval)
            return pen

        return super().fit(**common, penalty_callback=penalty_callback, penalty_weight=float(monotonic_penalty_weight))


class ExactGPModel(gp.ExactGP):
    def __init__(self, train_x, train_y, likelihood):
        super().__init__(train_x, train_y, likelihood)
        n_d = train_x.shape[1]
        assert n_d == 2, "Only two dimensions supported"
        self.dims = np.arange(n_d)
        self.time_dim = [self.dims[0]]
        self.stage_dim = [self.dims[1]]
        self.powerlaw = PowerLawTransform()
        self.mean_module = NoOpMean()

        stage = np.asarray(train_x[:, self.stage_dim[0]])
        b_min, b_max = np.quantile(stage, 0.10), np.quantile(stage, 0.90)
        sigmoid_lower = SigmoidKernel(active_dims=self.stage_dim, b_constraint=Interval(b_min, b_max))
        sigmoid_upper = InvertedSigmoidKernel(sigmoid_kernel=sigmoid_lower, active_dims=self.stage_dim,
                                              b_constraint=Interval(b_min, b_max))
        kernel = self.cov_base(eta_prior=HalfNormalPrior(scale=1.0)) + self.cov_periodic(eta_prior=HalfNormalPrior(scale=0.2))
        upper_kernel = self.cov_bend(eta_prior=HalfNormalPrior(scale=0.6))
        lower_kernel = self.cov_shift(
            eta_prior=HalfNormalPrior(scale=0.6),
            time_prior=GammaPrior(concentration=3, rate=1),
            stage_prior=GammaPrior(concentration=3, rate=2),
        ) + self.cov_shift(
            eta_prior=HalfNormalPrior(scale=0.3),
            time_prior=GammaPrior(concentration=1, rate=7),
            stage_prior=GammaPrior(concentration=3, rate=1),
        )
        lower_kernel_warped = LogWarpKernel(lower_kernel, self.stage_dim[0])
        upper_kernel_warped = LogWarpKernel(upper_kernel, self.stage_dim[0])
        kernel_warped = LogWarpKernel(kernel, self.stage_dim[0])
        self.covar_module = sigmoid_lower * lower_kernel_warped + sigmoid_upper * upper_kernel_warped + kernel_warped

    def prior_mean(self, x):
        """Mean half of the reference's ``forward`` (gpytorch.py:258-265), in-place clamps included."""
        self.powerlaw.b.data.clamp_(1.2, 2.5)
        return self.mean_module(self.powerlaw(x[:, self.stage_dim[0]]).unsqueeze(-1))

    def prepare_eval(self, train_x, x):
        """Eval mode evaluates forward on [X; X*], so the c-clamp sees the test stages too."""
        s = self.stage_dim[0]
        self.powerlaw.clamp_c(torch.minimum(train_x[:, s].min(), x[:, s].min()))

    def cov_shift(self, eta_prior=None, time_prior=None, stage_prior=None):
        if eta_prior is None:
            eta_prior = HalfNormalPrior(scale=0.3)
        if time_prior is None:
            time_prior = GammaPrior(concentration=1, rate=7)
        if stage_prior is None:
            stage_prior = GammaPrior(concentration=2, rate=1)
        return ScaleKernel(
            MaternKernel(active_dims=self.stage_dim, lengthscale_prior=stage_prior, nu=2.5)
            * MaternKernel(active_dims=self.time_dim, lengthscale_prior=time_prior, nu=1.5),
            outputscale_prior=eta_prior,
        )

    def cov_bend(self, eta_prior=None):
        if eta_prior is None:
            eta_prior = HalfNormalPrior(scale=0.2)
        return ScaleKernel(
            MaternKernel(active_dims=self.stage_dim, lengthscale_prior=GammaPrior(concentration=3, rate=2))
            * MaternKernel(active_dims=self.time_dim, lengthscale_prior=GammaPrior(concentration=4, rate=2)),
            outputscale_prior=eta_prior,
        )

    def cov_periodic(self, ls_prior=None, eta_prior=None):
        if eta_prior is None:
            eta_prior = HalfNormalPrior(scale=0.5)
        if ls_prior is None:
            ls_prior = GammaPrior(concentration=9, rate=10)
        return ScaleKernel(
            PeriodicKernel(active_dims=self.time_dim, period_length_prior=NormalPrior(loc=1.0, scale=0.05),
                           lengthscale_prior=ls_prior)
            * MaternKernel(active_dims=self.time_dim, nu=2.5),
            outputscale_prior=eta_prior,
        )

    def cov_base(self, eta_prior=None):
        if eta_prior is None:
            eta_prior = HalfNormalPrior(scale=1.0)
        ls = GammaPrior(concentration=4.0, rate=4.0)
        return ScaleKernel(MaternKernel(active_dims=self.stage_dim, lengthscale_prior=ls), outputscale_prior=eta_prior)
