"""``MarginalHIP`` -- the MI355X-native exact-GP engine behind discontinuum's engine surface.

Drop-in for ``MarginalGPyTorch`` (``src/discontinuum/engines/gpytorch.py:36-626``): same
``fit / predict / predict_grid / sample / save / load / build_model`` signatures, attributes and error
behaviour, same optimisation loop (Adam/AdamW, ReduceLROnPlateau, clip-norm 1.0, NaN policy, early
stopping, resume).  What changes is underneath ``mll(self.model(train_x), train_y)`` and
``likelihood(model(x))``: the Gram build, Cholesky, solves and gradient contraction run as hand-written
HIP kernels on the GPU through the C ABI of ``libdgp_hip.so``.  There is no CPU path: construction of
the device plan raises if the GPU or the shared library is missing.

Differences from the reference, on purpose:
  * the factorisation is exact at every n (stock gpytorch switches to CG/Lanczos above n = 800,
    SURVEY.md Appendix A.7);
  * arithmetic dtype is ``self.dtype`` (float64 by default; the reference casts to float32 at
    ``engines/gpytorch.py:221-222``) -- set ``MarginalHIP.dtype = torch.float32`` to match it.
"""
from __future__ import annotations

import math

from dataclasses import dataclass

import numpy as np
import torch
import tqdm

from ..backend import GPPlan
from ..gp.lowering import lower
from ..gp.mll import ExactMarginalLogLikelihood, NotPSDError, predictive_mean
from ..xr_compat import DataArray
from .base import BaseModel, is_fitted


def _get_optimizer_name(optimizer_obj):
    if isinstance(optimizer_obj, torch.optim.AdamW):
        return "adamw"
    if isinstance(optimizer_obj, torch.optim.Adam):
        return "adam"
    return optimizer_obj.__class__.__name__


@dataclass
class PriorSpec:
    """What ``self.model(train_x)`` hands to the marginal likelihood: the device plan, the constrained
    kernel hyperparameters (host, with grad), the prior mean and the noise diagonal (device, with grad)."""

    plan: object
    theta: torch.Tensor
    mean: torch.Tensor
    noise: torch.Tensor


class MarginalHIP(BaseModel):
    dtype = torch.float64
    device = "cuda"
    _plan_factory = staticmethod(GPPlan)  # tests substitute an oracle-backed double; the product never does

    def __init__(self, model_config: dict | None = None):
        super().__init__(model_config=model_config)
        self._resume_info = None
        self._last_optimizer = None
        self._last_scheduler = None
        self._current_iteration = 0
        self._plan = None
        self._factor_key = None

    # ------------------------------------------------------------------ device plumbing
    def _tensor(self, a):
        return torch.as_tensor(np.asarray(a), dtype=self.dtype).to(self.device).contiguous()

    def _setup_device(self, train_x, train_y):
        """Lower the kernel tree, (re)create the device plan and upload the training inputs."""
        n, d = train_x.shape
        model_name, self._theta_fn = lower(self.model.covar_module, d)
        key = (model_name, n, d, self.dtype, str(self.device))
        if self._plan is None or self._plan_key != key:
            self._plan = self._plan_factory(model_name, n, d, dtype=self.dtype, device=self.device)
            self._plan_key = key
        self._train_x = train_x.to(self.device, self.dtype).contiguous()
        self._train_y = train_y.to(self.device, self.dtype).contiguous()
        self._plan.set_inputs(self._train_x)
        self._factor_key = None

    def _prior(self):
        """``self.model(train_x)`` of the reference loop (engines/gpytorch.py:350)."""
        return PriorSpec(
            plan=self._plan,
            theta=self._theta_fn(),
            mean=self.model.prior_mean(self._train_x),
            noise=self.likelihood.train_noise(self._train_x.device, self.dtype),
        )

    def _differentiable_mean(self, x: torch.Tensor):
        """Posterior mean at model-space points ``x`` WITH gradients to every hyperparameter
        (``likelihood(model(x)).mean`` inside the reference's penalty callback,
        src/rating_gp/models/gpytorch.py:160-176).  Valid inside a training iteration, after the marginal
        likelihood of that iteration has been evaluated (the plan holds its factorisation)."""
        x = x.to(self.device, self.dtype).contiguous()
        if hasattr(self.model, "prepare_eval"):
            self.model.prepare_eval(self._train_x, x)
        spec = self._prior()
        r = (self._train_y - spec.mean).contiguous()
        return predictive_mean(self._plan, spec.theta, r, spec.noise.contiguous(), x) + self.model.prior_mean(x)

    def _param_key(self):
        return tuple(float(v) for p in self.model.parameters() for v in p.detach().reshape(-1).tolist())

    # ------------------------------------------------------------------ checkpointing
    @classmethod
    def load(cls, f, covariates, target, target_unc=None):
        """Load a checkpoint written by ``save()`` and prepare for prediction / resumed fitting
        (engines/gpytorch.py:47-105)."""
        ckpt = torch.load(f, map_location="cpu", weights_only=False)
        model = cls()
        model.dm.fit(target=target, covariates=covariates, target_unc=target_unc)
        model.X, model.y = model.dm.X, model.dm.y
        train_x = torch.tensor(model.X, dtype=model.dtype)
        train_y = torch.tensor(model.y, dtype=model.dtype)
        if target_unc is None:
            model.model = model.build_model(train_x, train_y)
        else:
            model.y_unc = model.dm.y_unc
            model.model = model.build_model(train_x, train_y, torch.tensor(model.y_unc, dtype=model.dtype))
        model.model.load_state_dict(ckpt["model_state_dict"])
        if ckpt.get("likelihood_state_dict") is not None:
            model.likelihood.load_state_dict(ckpt["likelihood_state_dict"])
        model._resume_info = {k: ckpt.get(k) for k in (
            "optimizer_state_dict", "optimizer_name", "optimizer_lr", "scheduler_state_dict", "scheduler_name")}
        model._resume_info["current_iteration"] = ckpt.get("current_iteration", 0)
        model._current_iteration = ckpt.get("current_iteration", 0)
        model._setup_device(train_x, train_y)
        model.is_fitted = True
        return model

    def save(self, f, optimizer_obj=None, scheduler=None, extra=None) -> None:
        """Weights + optimizer/scheduler state, same dictionary keys as engines/gpytorch.py:147-159."""
        if optimizer_obj is None:
            optimizer_obj = getattr(self, "_last_optimizer", None)
        if scheduler is None:
            scheduler = getattr(self, "_last_scheduler", None)
        if not hasattr(self, "model"):
            raise RuntimeError("No model to save. Call fit() first.")
        if not hasattr(self, "likelihood"):
            raise RuntimeError("No likelihood to save. Call fit() first.")
        opt_name = lr_val = None
        if optimizer_obj is not None:
            opt_name = _get_optimizer_name(optimizer_obj)
            try:
                lr_val = optimizer_obj.param_groups[0].get("lr", None)
            except Exception:  # noqa: BLE001
                lr_val = None
        torch.save({
            "model_class": f"{self.__class__.__module__}.{self.__class__.__name__}",
            "model_state_dict": self.model.state_dict(),
            "likelihood_state_dict": self.likelihood.state_dict(),
            "optimizer_state_dict": optimizer_obj.state_dict() if optimizer_obj is not None else None,
            "optimizer_name": opt_name,
            "optimizer_lr": lr_val,
            "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
            "scheduler_name": scheduler.__class__.__name__ if scheduler is not None else None,
            "current_iteration": getattr(self, "_current_iteration", 0),
            "model_config": getattr(self, "model_config", None),
            "extra": extra or {},
        }, f)

    # ------------------------------------------------------------------ fit
    def fit(self, covariates, target, target_unc=None, iterations: int = 100, optimizer: str | None = None,
            learning_rate: float | None = None, early_stopping: bool = False, patience: int = 60,
            scheduler: bool = True, resume: bool = False, penalty_callback=None, penalty_weight: float = 0.0):
        """Fit the model to data; parameters as ``MarginalGPyTorch.fit`` (engines/gpytorch.py:162-212)."""
        has_model = (getattr(self, "model", None) is not None and getattr(self, "likelihood", None) is not None
                     and self.is_fitted)
        from_checkpoint = self._resume_info is not None and has_model
        from_interruption = resume and has_model and not from_checkpoint
        if not from_interruption:
            self.dm.fit(target=target, covariates=covariates, target_unc=target_unc)
        self.X, self.y = self.dm.X, self.dm.y
        train_x = torch.tensor(self.X, dtype=self.dtype)
        train_y = torch.tensor(self.y, dtype=self.dtype)
        can_restore = from_checkpoint or from_interruption

        def fresh_build():
            if target_unc is None:
                self.model = self.build_model(train_x, train_y)  # also sets self.likelihood
            else:
                self.y_unc = self.dm.y_unc
                self.model = self.build_model(train_x, train_y, torch.tensor(self.y_unc, dtype=self.dtype))

        if not can_restore:
            fresh_build()
        elif from_checkpoint:
            try:
                self.model.set_train_data(inputs=train_x, targets=train_y, strict=False)
            except Exception:  # noqa: BLE001
                fresh_build()
                can_restore = False
        self._setup_device(train_x, train_y)
        self.model.train()
        self.likelihood.train()

        resume_info = self._resume_info or {}
        if from_interruption and self._last_optimizer is not None:
            resume_info = {
                "optimizer_name": _get_optimizer_name(self._last_optimizer),
                "optimizer_lr": self._last_optimizer.param_groups[0]["lr"] if self._last_optimizer.param_groups else None,
                "optimizer_state_dict": self._last_optimizer.state_dict(),
                "scheduler_state_dict": self._last_scheduler.state_dict() if self._last_scheduler else None,
            }
        saved_name, saved_lr = resume_info.get("optimizer_name"), resume_info.get("optimizer_lr")
        opt_choice = optimizer if optimizer is not None else (saved_name or "adam")
        lr_choice = learning_rate if learning_rate is not None else (saved_lr or 0.05)
        params = list(self.model.parameters())
        if opt_choice == "adamw" or (saved_name and saved_name.lower() == "adamw"):
            optimizer_obj = torch.optim.AdamW(params, lr=lr_choice, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                                              foreach=True)
        elif opt_choice == "adam" or (saved_name and saved_name.lower() == "adam"):
            optimizer_obj = torch.optim.Adam(params, lr=lr_choice, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4,
                                             foreach=True)  # same update rule, one multi-tensor call per step
        else:
            raise ValueError(f"Unsupported optimizer: {opt_choice!r}. Supported optimizers are 'adam' and 'adamw'.")
        if can_restore and resume_info.get("optimizer_state_dict") is not None:
            try:
                optimizer_obj.load_state_dict(resume_info["optimizer_state_dict"])
            except Exception:  # noqa: BLE001, S110
                pass
        scheduler_obj = None
        if scheduler:
            scheduler_obj = torch.optim.lr_scheduler.ReduceLROnPlateau(
                optimizer_obj, mode="min", factor=0.7, patience=max(20, patience // 2), threshold=1e-4,
                threshold_mode="rel", min_lr=1e-6, cooldown=10)
            if can_restore and resume_info.get("scheduler_state_dict") is not None:
                try:
                    scheduler_obj.load_state_dict(resume_info["scheduler_state_dict"])
                except Exception:  # noqa: BLE001, S110
                    pass

        mll = ExactMarginalLogLikelihood(self.likelihood, self.model)

        start_iteration = self._current_iteration if (resume and hasattr(self, "_current_iteration")) else 0
        if iterations - start_iteration <= 0:
            print(f"Model already trained for {start_iteration} iterations (>= target {iterations}). "
                  "No further training needed.")
            return
        pbar = tqdm.tqdm(range(iterations - start_iteration), ncols=100, desc=f"Training {start_iteration}->{iterations}")
        best_obj, patience_counter, min_improvement = float("inf"), 0, 1e-6
        nan_loss_counter = 0
        i = 0

        def plateau_bookkeeping(obj_item):
            nonlocal best_obj, patience_counter
            if obj_item < best_obj - min_improvement:
                best_obj, patience_counter = obj_item, 0
            else:
                patience_counter += 1
            if early_stopping and patience_counter >= patience:
                print(f"\nEarly stopping triggered after {i + 1} iterations")
                print(f"Best objective: {best_obj:.6f}")
                return True
            return False

        try:
            for i in pbar:
                self._current_iteration = start_iteration + i
                optimizer_obj.zero_grad(set_to_none=True)
                output = self._prior()
                try:
                    nll = -mll(output, self._train_y)
                except Exception:
                    nan_loss_counter += 1
                    if nan_loss_counter > 10:
                        raise
                    continue
                penalty_val = None
                if penalty_callback is not None and penalty_weight > 0.0:
                    try:
                        penalty_val = penalty_callback()
                        if not torch.is_tensor(penalty_val):
                            penalty_val = None
                    except Exception:  # noqa: BLE001
                        penalty_val = None
                objective = nll
                if penalty_val is not None:
                    objective = objective + float(penalty_weight) * penalty_val.to(nll.device, nll.dtype)
                if torch.isnan(objective) or torch.isinf(objective):
                    nan_loss_counter += 1
                    if nan_loss_counter > 10:
                        raise RuntimeError(
                            f"Encountered more than 10 consecutive NaN/Inf objectives at iteration {i + 1}")
                    continue
                nan_loss_counter = 0
                objective.backward()
                total_norm = torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
                # the reference scans every p.grad for NaN after clipping (engines/gpytorch.py:387-392); a clipped
                # gradient holds a NaN exactly when the pre-clip norm is NaN or Inf (Inf * 0 = NaN), so one scalar says it
                has_nan_grad = not math.isfinite(float(total_norm))
                if has_nan_grad:
                    for p in params:
                        if p.grad is not None:
                            p.grad = torch.nan_to_num(p.grad, nan=0.0, posinf=0.0, neginf=0.0)
                optimizer_obj.step()
                obj_item = float(objective.item())
                if scheduler_obj is not None:
                    scheduler_obj.step(obj_item)
                stop = plateau_bookkeeping(obj_item)
                if has_nan_grad:
                    if stop:
                        break
                    continue
                suffix = f"obj={obj_item:.4f}, lr={optimizer_obj.param_groups[0]['lr']:.1e}"
                if penalty_val is not None:
                    try:
                        suffix += f", pen={float(penalty_val.item()):.3e}"
                    except Exception:  # noqa: BLE001, S110
                        pass
                pbar.set_postfix_str(suffix)
                if stop:
                    break
        except KeyboardInterrupt:
            print(f"\nTraining interrupted at iteration {i + 1}")
            print(f"Best objective: {best_obj:.6f}")
        finally:
            self.is_fitted = True
        self._last_optimizer = optimizer_obj
        self._last_scheduler = scheduler_obj
        self._factor_key = None
        return

    # ------------------------------------------------------------------ prediction
    def _ensure_factor(self):
        """Eval-mode cache of the reference's prediction strategy: (L, L^-1, alpha) at the current
        hyperparameters, rebuilt only when a parameter changed."""
        key = self._param_key()
        if self._factor_key != key:
            with torch.no_grad():
                spec = self._prior()
                r = (self._train_y - spec.mean).contiguous()
                out = self._plan.factorize(spec.theta, r, spec.noise.contiguous())
                info = int(out[3].item())
            if info != 0:
                raise NotPSDError(f"Matrix not positive definite: Cholesky pivot {info} is not positive")
            self._factor_key = key
            self._factor_theta = spec.theta.detach()

    def _model_space_predict(self, x: torch.Tensor):
        """(mu, var) in model space -- ``__gpytorch_predict`` of the reference (engines/gpytorch.py:599-626)."""
        self.model.eval()
        self.likelihood.eval()
        x = x.to(self.device, self.dtype).contiguous()
        with torch.no_grad():
            if hasattr(self.model, "prepare_eval"):
                self.model.prepare_eval(self._train_x, x)  # data-dependent clamps see [X; X*]
            self._ensure_factor()
            kmean, kvar = self._plan.predict(self._factor_theta, x)
            mu = kmean + self.model.prior_mean(x)
            var = kvar + self.likelihood.predictive_noise(x.shape[0], x.device, self.dtype)
        return mu, var

    @is_fitted
    def predict(self, covariates, diag=True, pred_noise=False):
        """Predictions in the original data space: (target, standard error) (engines/gpytorch.py:461-501).
        ``diag`` / ``pred_noise`` are accepted and ignored exactly like the reference does."""
        Xnew = torch.tensor(self.dm.Xnew(covariates), dtype=self.dtype)
        mu, var = self._model_space_predict(Xnew)
        target = self.dm.y_t(mu.cpu().numpy())
        target = target.assign_coords(covariates.coords)
        se = self.dm.error_pipeline.inverse_transform(var.cpu().numpy())
        se = se.assign_coords(covariates.coords)
        return target, se

    @is_fitted
    def predict_grid(self, covariate: str, coord: str | None = None, t_step: int = 12):
        """Posterior mean on a (coord x 18) grid (engines/gpytorch.py:503-549)."""
        if coord is None:
            coord = next(iter(self.dm.data.covariates.coords))
        coord_dim, covariate_dim = self.dm.get_dim(coord), self.dm.get_dim(covariate)
        x_max, x_min = self.dm.X.max(axis=0), self.dm.X.min(axis=0)
        n_cov = 18
        n_coord = int(np.round((x_max - x_min)[coord_dim] * t_step))
        x_coord = torch.linspace(float(x_min[coord_dim]), float(x_max[coord_dim]), n_coord, dtype=self.dtype)
        x_cov = torch.linspace(float(x_min[covariate_dim]), float(x_max[covariate_dim]), n_cov, dtype=self.dtype)
        mu, _var = self._model_space_predict(torch.cartesian_prod(x_coord, x_cov))
        target = self.dm.y_t(mu.cpu().numpy())
        index = self.dm.covariate_pipelines[coord].inverse_transform(x_coord.numpy())
        cov_vals = self.dm.covariate_pipelines[covariate].inverse_transform(x_cov.numpy())
        return DataArray(np.asarray(target.data).reshape(n_coord, n_cov), coords=[np.asarray(index), np.asarray(cov_vals)],
                         dims=[coord, covariate], attrs=target.attrs)

    @is_fitted
    def sample(self, covariates, n=1000):
        """Draws from the latent posterior (engines/gpytorch.py:551-593): full m x m covariance
        K** - V^T V with V = L^-1 K(X, X*), its Cholesky factor (our blocked HIP potrf) times N(0, I)."""
        Xnew = torch.tensor(self.dm.Xnew(covariates), dtype=self.dtype).to(self.device).contiguous()
        self.model.eval()
        self.likelihood.eval()
        with torch.no_grad():
            if hasattr(self.model, "prepare_eval"):
                self.model.prepare_eval(self._train_x, Xnew)
            self._ensure_factor()
            mean, cov_factor = self._plan.posterior_factor(self._factor_theta, Xnew)
            mean = mean + self.model.prior_mean(Xnew)
            z = torch.randn(cov_factor.shape[0], n, dtype=self.dtype, device=cov_factor.device)
            sim = (mean[:, None] + cov_factor @ z).T.contiguous()  # (n, m)
        temp = self.dm.y_t(sim.reshape(-1).cpu().numpy())
        data = np.asarray(temp.data).reshape(n, -1)
        return DataArray(data, coords=dict(covariates.coords, draw=np.arange(n)),
                         dims=["draw"] + list(covariates.coords), attrs=temp.attrs)

    def build_model(self, X, y, **kwargs):
        raise NotImplementedError("This method must be implemented in a subclass")
