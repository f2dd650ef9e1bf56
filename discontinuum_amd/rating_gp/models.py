"""rating-gp (stage -> discharge) on the MI355X engine.

What is modelled is the reference's (``src/rating_gp/models/gpytorch.py:28-372``): a power-law mean
a + b log(stage - c) and the covariance

    g(s) g(s')          * [ shift_1 + shift_2 ](t, log s)     low-flow part, switched on below the gate
  + (1-g(s))(1-g(s'))   * bend(t, log s)                      high-flow part, switched on above it
  + base(log s) + periodic(t)                                 everywhere

with g a steep logistic gate in stage whose switch point b is learned inside the 10-90 % stage quantiles, every term
an output scale times Matern / periodic factors with the reference's priors (table below).  The module tree this
file builds -- and therefore every ``state_dict`` key -- is the one the reference builds, so its checkpoints load;
the construction itself is table-driven instead of one method per term.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import gp
from ..engines.base import DataMixin, ModelConfig
from .. import _lib
from ..engines.hip import MarginalHIP, MeanShortcut
from ..gp import kernels as K
from ..gp.constraints import Interval
from ..gp.means import NoOpMean
from ..gp.priors import GammaPrior, HalfNormalPrior, NormalPrior
from ..pipeline import TimePipeline, UnitPipeline

TIME, STAGE = 0, 1  # columns of the design matrix (RatingDataMixin fixes the order)


def _gamma(concentration, rate):
    return GammaPrior(concentration=concentration, rate=rate)


def _matern(column, nu, lengthscale=None):
    return K.MaternKernel(nu=nu, active_dims=[column], lengthscale_prior=_gamma(*lengthscale) if lengthscale else None)


def _scaled(kernel, eta_scale):
    """sigma^2 * kernel with sigma^2 ~ HalfNormal(eta_scale)."""
    return K.ScaleKernel(kernel, outputscale_prior=HalfNormalPrior(scale=eta_scale))


# (eta scale, stage lengthscale Gamma(k, rate), time lengthscale Gamma(k, rate), Matern nu in time)
#   reference: cov_shift x2 (gpytorch.py:242-250, 293-316) and cov_bend (:318-335)
_SHIFT_A = (0.6, (3, 2), (3, 1), 1.5)
_SHIFT_B = (0.3, (3, 1), (1, 7), 1.5)
_BEND = (0.6, (3, 2), (4, 2), 2.5)


def _stage_time_term(eta_scale, stage_ls, time_ls, time_nu):
    return _scaled(_matern(STAGE, 2.5, stage_ls) * _matern(TIME, time_nu, time_ls), eta_scale)


def rating_covariance(stage):
    """The composite kernel for model-space stages ``stage`` (they only fix the gate's admissible interval)."""
    lo, hi = (float(q) for q in np.quantile(np.asarray(stage), [0.10, 0.90]))
    gate = K.SigmoidKernel(active_dims=[STAGE], b_constraint=Interval(lo, hi))
    anti_gate = K.InvertedSigmoidKernel(sigmoid_kernel=gate, active_dims=[STAGE], b_constraint=Interval(lo, hi))
    low_flow = _stage_time_term(*_SHIFT_A) + _stage_time_term(*_SHIFT_B)
    high_flow = _stage_time_term(*_BEND)
    base = _scaled(_matern(STAGE, 2.5, (4.0, 4.0)), 1.0)                                    # cov_base  (:360-372)
    seasonal = _scaled(K.PeriodicKernel(active_dims=[TIME], lengthscale_prior=_gamma(9, 10),   # cov_periodic (:337-358)
                                        period_length_prior=NormalPrior(loc=1.0, scale=0.05))
                       * _matern(TIME, 2.5), 0.2)

    def in_log_stage(kernel):
        return K.LogWarpKernel(kernel, STAGE)

    return gate * in_log_stage(low_flow) + anti_gate * in_log_stage(high_flow) + in_log_stage(base + seasonal)


class PowerLawTransform(nn.Module):
    """mu(s) = a + b log(s - c).  Initial values a ~ N(0,1), b ~ N(1.3,1), c ~ U(0,1) and the clamp that keeps
    s - c positive are the reference's (gpytorch.py:28-40)."""

    def __init__(self):
        super().__init__()
        draw = {"a": torch.randn(1, dtype=torch.float64), "b": 1.3 + torch.randn(1, dtype=torch.float64),
                "c": torch.rand(1, dtype=torch.float64)}
        for name, value in draw.items():
            setattr(self, name, nn.Parameter(value))

    def clamp_c(self, smallest_stage):
        # on .data, like the reference: the clamp must not count as a modification of a tensor autograd has saved
        self.c.data.clamp_(max=float(smallest_stage) - 1e-6)

    def forward(self, stage):
        self.clamp_c(stage.min())
        a, b, c = (getattr(self, name).to(stage.device, stage.dtype) for name in "abc")
        return a + b * torch.log(stage - c)


class ExactGPModel(gp.ExactGP):
    def __init__(self, train_x, train_y, likelihood):
        super().__init__(train_x, train_y, likelihood)
        if train_x.shape[1] != 2:
            raise AssertionError("Only two dimensions supported")
        self.time_dim, self.stage_dim = [TIME], [STAGE]
        self.powerlaw = PowerLawTransform()
        self.mean_module = NoOpMean()
        self.covar_module = rating_covariance(train_x[:, STAGE])

    def prior_mean(self, x):
        """The mean half of the reference's ``forward`` (gpytorch.py:258-265): the exponent b is held in [1.2, 2.5]
        by clamping the parameter itself on every call."""
        self.powerlaw.b.data.clamp_(1.2, 2.5)
        return self.mean_module(self.powerlaw(x[:, STAGE]).unsqueeze(-1))

    def prepare_eval(self, train_x, x):
        """gpytorch's eval mode runs ``forward`` on [X; X*]: the c-clamp then sees the test stages too."""
        self.powerlaw.clamp_c(torch.minimum(train_x[:, STAGE].min(), x[:, STAGE].min()))


class _PowerLawShortcut(MeanShortcut):
    """Host-side form of rating-gp's mean and noise for the marginal likelihood: mu_i = a + b log(s_i - c) and
    Sigma = diag(y_unc) + sigma_add^2 I, with the reference's in-forward clamps (gpytorch.py:39, 259).  The residual,
    the noise diagonal and the two weight vectors log(s - c), 1 / (s - c) are a few elementwise device kernels per
    iteration with the scalars passed by value; the gradients of (a, b, c, sigma_add^2) come from the result row:
        dNLL/da = -sum dr      dNLL/db = -sum dr log(s - c)      dNLL/dc = b sum dr / (s - c)      dNLL/dsigma^2 = sum dnoise"""

    def __init__(self, engine):
        # built once per training set (the stage column, its minimum and the fixed noise do not change during a fit)
        self.powerlaw, self.likelihood = engine.model.powerlaw, engine.likelihood
        self.stage = engine._train_x[:, STAGE].contiguous()
        self.stage_min = float(self.stage.min())
        self.fixed_noise = self.likelihood.train_noise_fixed(self.stage.device, engine.dtype)
        self.second = None
        self.b_value = 1.0

    @property
    def params(self):
        return (self.powerlaw.a, self.powerlaw.b, self.powerlaw.c, self.second)

    def param_values(self):
        return (self.powerlaw.a, self.powerlaw.b, self.powerlaw.c, self.likelihood.second_noise)

    def residual_and_noise(self, plan, target):
        pw = self.powerlaw
        pw.b.data.clamp_(1.2, 2.5)
        pw.clamp_c(self.stage_min)
        # host tensor with grad: the constraint transform of the raw noise parameter, evaluated once per iteration
        self.second = self.likelihood.second_noise
        a, b, c = (float(t.detach()) for t in (pw.a, pw.b, pw.c))
        self.b_value = b
        shifted = self.stage - c
        weights = torch.stack([torch.log(shifted), torch.reciprocal(shifted)])
        plan.set_dr_weights(weights)
        self.weights = weights
        return target - a - b * weights[0], self.fixed_noise + float(self.second.detach())

    def grads(self, row):
        w0 = _lib.OUT_DR_W0
        return (-row[_lib.OUT_SUM_DR], -row[w0], self.b_value * row[w0 + 1], row[_lib.OUT_SUM_DNOISE])


class _MonotonicPenalty:
    """mean(relu(-d mu / d stage)) of the posterior mean on a fresh random grid each call -- time uniform over the
    record, stage log-uniform (denser at low stage), forward difference 1e-3 in model-space stage
    (gpytorch.py:130-187).  With ``interval`` > 1 only every interval-th call is evaluated and weighted by it."""

    FD = 1e-3
    uniforms = None  # test hook: callable(m) -> (2, m) uniforms in [0, 1) replacing the random draw

    def __init__(self, engine, grid_size, interval):
        self.engine, self.m, self.interval, self.calls = engine, int(grid_size), int(interval), 0

    def __call__(self):
        eng = self.engine
        dev, dt = eng._train_x.device, eng.dtype
        self.calls += 1
        if self.interval > 1 and self.calls % self.interval:
            return torch.zeros((), device=dev, dtype=dt)
        lo, hi = eng.dm.X.min(axis=0), eng.dm.X.max(axis=0)
        u = (self.uniforms(self.m).to(dev, dt) if self.uniforms is not None else torch.rand((2, self.m), dtype=dt, device=dev))
        t = lo[TIME] + u[0] * (hi[TIME] - lo[TIME])
        log_lo, log_hi = float(np.log(lo[STAGE] + 1e-6)), float(np.log(hi[STAGE] + 1e-6))
        s = torch.exp(log_lo + u[1] * (log_hi - log_lo))
        here = torch.stack([t, s], dim=1)
        there = torch.stack([t, s + self.FD], dim=1)
        mu = eng._differentiable_mean(torch.cat([here, there]))  # both point sets through one solve
        slope = (mu[self.m:] - mu[: self.m]) / self.FD
        penalty = torch.relu(-slope).mean()
        return penalty * float(self.interval) if self.interval > 1 else penalty


    def explicit_terms(self, obj, x):
        """The same penalty for the closed-form training loop (``gp/explicit.py::ExplicitObjective.evaluate``): value and
        gradient with respect to the CONSTRAINED values ``x`` (hyperparameters by ``obj.theta_src``; a, b, c and the learned
        noise by ``obj.shortcut_src``), without autograd: one ``dgp_predict_mean``, the slope / relu / mean on the host,
        one ``dgp_mean_vjp`` for d pen / d mu -> (d theta, d r, d noise), and the power law's chain rule in closed form
        (r = y - a - b log(s - c): the reductions ``_PowerLawShortcut.grads`` reads from a result row, here formed from
        this VJP's dr / dnoise; plus the prior mean's own part at the grid points)."""
        eng, sc = self.engine, obj.shortcut
        dev, dt = eng._train_x.device, eng.dtype
        self.calls += 1
        zero = np.zeros_like(x)
        if self.interval > 1 and self.calls % self.interval:
            return 0.0, zero
        lo, hi = eng.dm.X.min(axis=0), eng.dm.X.max(axis=0)
        u = (self.uniforms(self.m).to(torch.float64).cpu().numpy() if self.uniforms is not None
             else torch.rand((2, self.m), dtype=torch.float64).numpy())
        t = lo[TIME] + u[0] * (hi[TIME] - lo[TIME])
        log_lo, log_hi = float(np.log(lo[STAGE] + 1e-6)), float(np.log(hi[STAGE] + 1e-6))
        st = np.exp(log_lo + u[1] * (log_hi - log_lo))
        grid = np.concatenate([np.stack([t, st], axis=1), np.stack([t, st + self.FD], axis=1)])
        pw = sc.powerlaw
        pw.clamp_c(min(sc.stage_min, float(grid[:, STAGE].min())))  # gpytorch's eval-mode forward sees the test stages too
        a, b, c = (float(v.detach()) for v in (pw.a, pw.b, pw.c))
        theta = x[obj.theta_src].tolist()
        xg = torch.tensor(grid, dtype=dt, device=dev)
        mu = eng._plan.predict_mean(theta, xg).to("cpu", torch.float64).numpy() + a + b * np.log(grid[:, STAGE] - c)
        m = self.m
        slope = (mu[m:] - mu[:m]) / self.FD
        scale = float(self.interval) if self.interval > 1 else 1.0
        value = scale * float(np.maximum(-slope, 0.0).mean())
        dslope = np.where(slope < 0.0, -scale / m, 0.0)          # d value / d slope
        w = np.concatenate([-dslope, dslope]) / self.FD           # d value / d mu
        if not w.any():
            return value, zero
        dtheta, dr, dnoise = eng._plan.mean_vjp(theta, xg, torch.tensor(w, dtype=dt, device=dev))
        wts = sc.weights                                          # (2, n): log(s - c), 1 / (s - c) of the training stages
        red = torch.cat([dr.sum().reshape(1), wts @ dr, dnoise.sum().reshape(1), dtheta.reshape(-1)]).to("cpu", torch.float64).numpy()
        gx = zero
        np.add.at(gx, obj.theta_src, red[4:4 + len(obj.theta_src)])
        shifted = grid[:, STAGE] - c
        direct = np.asarray([w.sum(), float(w @ np.log(shifted)), -b * float(w @ (1.0 / shifted)), 0.0])
        through_r = np.asarray([-red[0], -red[1], b * red[2], red[3]])
        np.add.at(gx, obj.shortcut_src, direct + through_r)
        return value, gx


class RatingDataMixin(DataMixin):
    """Design matrix columns (time, stage); stage rescaled to [1, 2] -- ``src/rating_gp/models/base.py:14-17``."""

    def build_datamanager(self, model_config: ModelConfig | None = None):
        self._build_datamanager({"time": TimePipeline, "stage": UnitPipeline}, model_config)


class RatingGPMarginalHIP(RatingDataMixin, MarginalHIP):
    """Stage-discharge rating curve as an exact GP (marginal likelihood), on the MI355X engine."""

    def __init__(self, model_config: ModelConfig | None = None):
        config = model_config or ModelConfig()
        super().__init__(model_config=config)
        self.build_datamanager(config)

    def build_model(self, X, y, y_unc=None):
        """Measurement variances ``y_unc`` as fixed noise plus one learned homoskedastic term with a HalfNormal(0.03)
        prior (gpytorch.py:64-79)."""
        fixed = y_unc if y_unc is not None else torch.full((1, y.shape[0]), 0.1 ** 2, dtype=y.dtype)
        self.likelihood = gp.likelihoods.FixedNoiseGaussianLikelihood(
            noise=fixed, learn_additional_noise=True, noise_prior=HalfNormalPrior(scale=0.03))
        return ExactGPModel(X, y, self.likelihood)

    def _mean_shortcut(self):
        lik = self.likelihood
        if (type(self.model) is ExactGPModel and type(self.model.mean_module) is NoOpMean
                and getattr(lik, "second_noise_covar", None) is not None and hasattr(self._plan, "set_dr_weights")):
            key = (id(self.model), id(lik), self._train_x.data_ptr(), self._train_x.shape[0])
            if getattr(self, "_shortcut_key", None) != key:
                self._shortcut, self._shortcut_key = _PowerLawShortcut(self), key
            return self._shortcut
        return None

    def fit(self, covariates, target, target_unc=None, iterations=100, optimizer=None, learning_rate=None,
            early_stopping=False, patience=60, scheduler=True, resume=False,
            monotonic_penalty_weight: float = 0.0, grid_size: int = 64, monotonic_penalty_interval: int = 1):
        """The engine's ``fit`` plus the optional monotonicity penalty (signature of gpytorch.py:81-128)."""
        callback, weight = None, 0.0
        if monotonic_penalty_weight > 0:
            callback = _MonotonicPenalty(self, grid_size, monotonic_penalty_interval)
            weight = float(monotonic_penalty_weight)
        return super().fit(covariates=covariates, target=target, target_unc=target_unc, iterations=iterations,
                           optimizer=optimizer, learning_rate=learning_rate, early_stopping=early_stopping,
                           patience=patience, scheduler=scheduler, resume=resume, penalty_callback=callback,
                           penalty_weight=weight)
