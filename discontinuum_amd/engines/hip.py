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

import numpy as np
import torch
import tqdm

from .. import _lib
from ..backend import GPPlan
from ..gp.explicit import ExplicitObjective
from ..gp.lowering import lower
from ..gp.mll import ExactMarginalLogLikelihood, NotPSDError, predictive_mean
from ..xr_compat import DataArray
from .base import BaseModel, ModelConfig, is_fitted


class _FlatState:
    """The optimiser state of all parameters in three flat buffers; ``state[p]['exp_avg' | 'exp_avg_sq' | 'step']``
    are views into them, so ``state_dict()`` still holds torch's per-parameter layout.  ``load_state_dict`` and
    torch's own ``step`` (the fallback) invalidate it: it is then rebuilt from whatever the state holds."""

    def __init__(self, params, state):
        self.sizes = [int(p.numel()) for p in params]
        self.shapes = [p.shape for p in params]
        self.exp_avg = torch.cat([state[p]["exp_avg"].detach().reshape(-1) for p in params])
        self.exp_avg_sq = torch.cat([state[p]["exp_avg_sq"].detach().reshape(-1) for p in params])
        self.steps = torch.stack([state[p]["step"].detach().reshape(()) for p in params])
        self.count = int(self.steps[0])  # equal for all parameters (checked by the caller)
        self.param_views = [p.detach().view(-1) for p in params]  # same storage as the parameters, flat shapes
        offset = 0
        for k, (p, n, shape) in enumerate(zip(params, self.sizes, self.shapes)):
            state[p]["exp_avg"] = self.exp_avg[offset:offset + n].view(shape)
            state[p]["exp_avg_sq"] = self.exp_avg_sq[offset:offset + n].view(shape)
            state[p]["step"] = self.steps[k]
            offset += n


def _flat_state(opt, params):
    """The optimiser's ``_FlatState``, (re)built if needed; None if the state is not one the fast path handles."""
    flat = opt.__dict__.get("_flat_state")
    if flat is not None:
        # the flat parameter views must still be the parameters' storage (someone may have reassigned ``p.data``)
        if (flat.param_views[0].data_ptr() == params[0].data_ptr()
                and flat.param_views[-1].data_ptr() == params[-1].data_ptr()):
            return flat
    state, first = opt.state, None
    for p in params:
        st = state.get(p)
        if not st or not torch.is_tensor(st.get("step")) or "exp_avg" not in st or "exp_avg_sq" not in st:
            return None  # before torch's first step has created the state
        if first is None:
            first = float(st["step"])
        elif float(st["step"]) != first:
            return None  # parameters that have taken different numbers of steps
    flat = opt.__dict__["_flat_state"] = _FlatState(params, state)
    return flat


def _lean_step(opt, decoupled, flat_grad=None):
    """One Adam / AdamW step with the arithmetic of ``torch.optim.adam._multi_tensor_adam`` (no amsgrad / maximize /
    capturable), element for element, on FLAT views of the state: a dozen scalar-sized host parameters cost torch's
    ``Optimizer.step`` about 200 us, three quarters of it bookkeeping and per-tensor dispatch; this is a dozen vector
    operations.  ``state_dict()`` / ``load_state_dict()`` and checkpoints keep torch's per-parameter layout
    (``_FlatState``), and parameters and state stay bit-identical to torch's (tests/test_engine_cpu.py).  Returns False
    if the fast path does not apply (first step, a parameter without gradient, several groups, unequal step counts):
    the caller then takes torch's own step."""
    if len(opt.param_groups) != 1:
        return False
    group = opt.param_groups[0]
    params = group["params"]
    if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
        return False
    if flat_grad is None:
        for p in params:
            if p.grad is None or p.grad.is_sparse:
                return False
    with torch.no_grad():
        flat = _flat_state(opt, params)
        if flat is None:
            return False
        beta1, beta2 = group["betas"]
        lr, wd, eps = float(group["lr"]), group["weight_decay"], group["eps"]
        if flat_grad is None:
            grad = torch.cat([p.grad.reshape(-1) for p in params])
        elif flat_grad.numel() != flat.exp_avg.numel():
            return False
        else:
            grad = flat_grad  # the caller's gradient of all parameters in group order (never modified here)
        flat.steps += 1
        flat.count += 1
        step = flat.count
        if wd != 0:
            if decoupled:
                torch._foreach_mul_(flat.param_views, 1 - lr * wd)
            else:
                grad = grad.add(torch.cat(flat.param_views), alpha=wd)
        flat.exp_avg.lerp_(grad, 1 - beta1)
        flat.exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        bias1, bias2 = 1 - beta1 ** step, 1 - beta2 ** step
        denom = flat.exp_avg_sq.sqrt().div_(bias2 ** 0.5).add_(eps)
        update = (flat.exp_avg * (-(lr / bias1))).div_(denom)  # addcdiv's  value * t1 / t2, in its order
        torch._foreach_add_(flat.param_views, list(update.split(flat.sizes)))
    return True


class _LeanMixin:
    """``step`` through ``_lean_step`` where it applies; anything that changes the state behind its back drops the
    flat view of it."""

    _decoupled = False

    def step(self, closure=None):
        if closure is None and _lean_step(self, self._decoupled):
            return None
        self.__dict__["_flat_state"] = None  # torch's own step may leave the parameters' step counts unequal
        return super().step(closure)

    def step_flat(self, flat_grad):
        """A step from the gradient of all parameters as ONE flat tensor (group order).  False if the flat state is
        not available (before torch's first step has created it): the caller then sets ``p.grad`` and calls ``step``."""
        return _lean_step(self, self._decoupled, flat_grad)

    def load_state_dict(self, state_dict):
        self.__dict__["_flat_state"] = None
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        self.__dict__["_flat_state"] = None
        return super().add_param_group(param_group)


class _LeanAdam(_LeanMixin, torch.optim.Adam):
    _decoupled = False


class _LeanAdamW(_LeanMixin, torch.optim.AdamW):
    _decoupled = True


def _clip_grad_norm(params, max_norm):
    """``torch.nn.utils.clip_grad_norm_(params, max_norm)`` (2-norm, in place) for a few dozen scalar-sized host
    gradients: the same arithmetic -- scale by min(1, max_norm / (norm + 1e-6)) -- through one flat reduction instead
    of the foreach machinery (85 -> 30 us per iteration).  Returns the total norm as a float."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0.0
    flat = grads[0].reshape(-1) if len(grads) == 1 else torch.cat([g.reshape(-1) for g in grads])
    total = float(torch.linalg.vector_norm(flat))
    coef = max_norm / (total + 1e-6)
    if not coef >= 1.0:  # a NaN norm scales by NaN, an infinite one by 0, exactly like torch's clamped coefficient
        for g in grads:
            g.mul_(coef)
    return total


def _get_optimizer_name(optimizer_obj):
    if isinstance(optimizer_obj, torch.optim.AdamW):
        return "adamw"
    if isinstance(optimizer_obj, torch.optim.Adam):
        return "adam"
    return optimizer_obj.__class__.__name__


class MeanShortcut:
    """A prior mean / noise model with a handful of parameters that stay on the host.  ``residual_and_noise`` builds
    y - mu and the noise diagonal on the device WITHOUT autograd (scalars travel as kernel arguments, nothing is copied
    to the device); ``grads`` maps the reductions the fit step leaves in its result row (``DGP_OUT_SUM_DR``,
    ``DGP_OUT_DR_W0..``, ``DGP_OUT_SUM_DNOISE``) to d NLL / d param for ``params`` (host tensors with grad)."""

    params: tuple = ()

    def param_values(self):
        """Current values of ``params`` without side effects on the device (``params`` itself may be refreshed only
        by ``residual_and_noise``)."""
        return self.params

    def residual_and_noise(self, plan, target):
        raise NotImplementedError

    def grads(self, row):
        raise NotImplementedError


class ConstantMeanShortcut(MeanShortcut):
    """mu = c (``ConstantMean``), fixed noise: r = y - c, d NLL / d c = -sum_i d NLL / d r_i."""

    def __init__(self, constant, noise_dev):
        self.params, self.noise_dev = (constant,), noise_dev

    def residual_and_noise(self, plan, target):
        return target - float(self.params[0].detach()), self.noise_dev

    def grads(self, row):
        return (-row[_lib.OUT_SUM_DR],)


class PriorSpec:
    """What ``self.model(train_x)`` hands to the marginal likelihood: the device plan, the constrained kernel
    hyperparameters (host, with grad), the prior mean and the noise diagonal (device, with grad; evaluated when
    read), and -- for mean / noise models that have one -- the host-side ``shortcut`` the marginal likelihood uses
    instead of them."""

    def __init__(self, plan, theta, mean, noise, shortcut=None):
        self.plan, self.theta, self._mean, self._noise, self.shortcut = plan, theta, mean, noise, shortcut

    @property
    def mean(self):
        if callable(self._mean):
            self._mean = self._mean()
        return self._mean

    @property
    def noise(self):
        if callable(self._noise):
            self._noise = self._noise()
        return self._noise


class MarginalHIP(BaseModel):
    dtype = torch.float64
    device = "cuda"
    explicit_host_algebra = True  # closed-form host algebra per iteration where the model allows it (gp/explicit.py)
    _plan_factory = staticmethod(GPPlan)  # tests substitute an oracle-backed double; the product never does

    def __init__(self, model_config: dict | None = None):
        super().__init__(model_config=model_config)
        self._resume_info = None
        self._last_optimizer = None
        self._last_scheduler = None
        self._current_iteration = 0
        self._plan = None
        self._factor_key = None
        self._pending_device = None  # (train_x, train_y) whose upload is deferred to the first prediction (fit_many)

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
        self._pending_device = None

    def _prior(self):
        """``self.model(train_x)`` of the reference loop (engines/gpytorch.py:350)."""
        return PriorSpec(
            plan=self._plan,
            theta=self._theta_fn(),
            mean=lambda: self.model.prior_mean(self._train_x),
            noise=lambda: self.likelihood.train_noise(self._train_x.device, self.dtype),
            shortcut=self._mean_shortcut(),
        )

    def _mean_shortcut(self):
        """The host-side form of this model's prior mean and noise (``MeanShortcut``) or None.  Here: a learned
        constant mean with fixed noise (loadest-gp); model packages override it for their own parametric means."""
        from ..gp.means import ConstantMean
        from ..gp.models import ExactGP

        module = getattr(self.model, "mean_module", None)
        if (type(module) is ConstantMean and type(self.model).prior_mean is ExactGP.prior_mean
                and getattr(self.likelihood, "second_noise_covar", None) is None):
            return ConstantMeanShortcut(module.constant, self.likelihood.train_noise(self._train_x.device, self.dtype))
        return None

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

    # ------------------------------------------------------------------ data -> host model -> device
    def _attach(self, covariates, target, target_unc, refit_data=True):
        """Fit the data manager (unless the caller keeps the current one) and return the model-space training tensors
        (x, y, y_unc or None) in ``self.dtype``."""
        if refit_data:
            self.dm.fit(target=target, covariates=covariates, target_unc=target_unc)
        self.X, self.y = self.dm.X, self.dm.y
        x = torch.tensor(self.X, dtype=self.dtype)
        y = torch.tensor(self.y, dtype=self.dtype)
        unc = None
        if target_unc is not None:
            self.y_unc = self.dm.y_unc
            unc = torch.tensor(self.y_unc, dtype=self.dtype)
        return x, y, unc

    def _fresh_model(self, x, y, unc):
        """``build_model`` with the reference's calling convention: the uncertainty is only passed when there is one.
        (The hook also sets ``self.likelihood``.)"""
        self.model = self.build_model(x, y) if unc is None else self.build_model(x, y, unc)

    # ------------------------------------------------------------------ checkpointing
    # Dictionary layout of the reference's checkpoints (engines/gpytorch.py:107-160): state dicts of model / likelihood /
    # optimiser / scheduler plus bookkeeping, the parameters under gpytorch's raw_* names.  Cross-engine loading is NOT
    # claimed: a reference-written file pickles its own ModelConfig class (not importable here, and never unpickled --
    # load() reads with weights_only=True), and the buffer keys of priors / constraints are unpinned (see _load_state).
    _OPTIMIZER_KEYS = ("optimizer_state_dict", "optimizer_name", "optimizer_lr", "scheduler_state_dict", "scheduler_name")

    def save(self, f, optimizer_obj=None, scheduler=None, extra=None) -> None:
        for attr in ("model", "likelihood"):
            if not hasattr(self, attr):
                raise RuntimeError(f"No {attr} to save. Call fit() first.")
        opt = optimizer_obj if optimizer_obj is not None else getattr(self, "_last_optimizer", None)
        sch = scheduler if scheduler is not None else getattr(self, "_last_scheduler", None)
        record = {
            "model_class": f"{type(self).__module__}.{type(self).__name__}",
            "model_config": self._config_record(),
            "current_iteration": getattr(self, "_current_iteration", 0),
            "extra": extra or {},
            "model_state_dict": self.model.state_dict(),
            "likelihood_state_dict": self.likelihood.state_dict(),
            "optimizer_state_dict": None, "optimizer_name": None, "optimizer_lr": None,
            "scheduler_state_dict": None, "scheduler_name": None,
        }
        if opt is not None:
            groups = getattr(opt, "param_groups", None) or [{}]
            record.update(optimizer_state_dict=opt.state_dict(), optimizer_name=_get_optimizer_name(opt),
                          optimizer_lr=groups[0].get("lr"))
        if sch is not None:
            record.update(scheduler_state_dict=sch.state_dict(), scheduler_name=type(sch).__name__)
        torch.save(record, f)

    def _config_record(self):
        """The model configuration as plain data (a checkpoint must load with ``weights_only=True``)."""
        config = getattr(self, "model_config", None)
        if isinstance(config, ModelConfig):
            return {"transform": config.transform}
        return config if isinstance(config, dict) else None

    @staticmethod
    def _load_state(module, state, what):
        """``load_state_dict`` that insists on every PARAMETER (raw_*, the power law's a / b / c) and tolerates
        differences in the buffers that only describe priors and constraints (``*_prior.scale``, ``*_constraint.
        lower_bound`` ...): those are constants of the model definition, rebuilt by ``build_model``, and whether a prior
        registers them as buffers differs between gpytorch versions -- the gpytorch side of this boundary is parity
        unpinned."""
        result = module.load_state_dict(state, strict=False)
        params = {name for name, _ in module.named_parameters()}
        missing = [k for k in result.missing_keys if k in params]
        unexpected = [k for k in result.unexpected_keys
                      if k.rsplit(".", 1)[-1].startswith("raw_") or "powerlaw" in k or "mean_module" in k]
        if missing or unexpected:
            raise RuntimeError(f"{what} state does not match this model: missing parameters {missing}, "
                               f"unexpected parameters {unexpected}")

    @classmethod
    def load(cls, f, covariates, target, target_unc=None):
        """A model restored from ``save()`` output and re-attached to its data: ready to predict, or to continue
        training with ``fit(..., resume=True)`` (engines/gpytorch.py:47-105).  The file is read with
        ``weights_only=True`` (nothing in it is executed); checkpoints of earlier versions that pickled the
        ``ModelConfig`` dataclass load under an allow-list of exactly that class."""
        with torch.serialization.safe_globals([ModelConfig]):
            record = torch.load(f, map_location="cpu", weights_only=True)
        # the reference rebuilds with the default configuration and ignores the one it saved (engines/gpytorch.py:69);
        # a model trained with transform="standard" would come back in the wrong data space, so the saved one is used
        saved_config = record.get("model_config")
        if isinstance(saved_config, dict) and "transform" in saved_config:
            saved_config = ModelConfig(transform=saved_config["transform"])
        self = cls(model_config=saved_config) if isinstance(saved_config, ModelConfig) else cls()
        x, y, unc = self._attach(covariates, target, target_unc)
        self._fresh_model(x, y, unc)
        self._load_state(self.model, record["model_state_dict"], "model")
        if record.get("likelihood_state_dict") is not None:
            self._load_state(self.likelihood, record["likelihood_state_dict"], "likelihood")
        self._current_iteration = record.get("current_iteration", 0)
        self._resume_info = {key: record.get(key) for key in cls._OPTIMIZER_KEYS}
        self._resume_info["current_iteration"] = self._current_iteration
        self._setup_device(x, y)
        self.is_fitted = True
        return self

    # ------------------------------------------------------------------ fit
    @staticmethod
    def _new_optimizer(params, name, lr):
        """The reference's two optimisers with its hyperparameters (engines/gpytorch.py:268-288)."""
        decay = {"adam": (_LeanAdam, 1e-4), "adamw": (_LeanAdamW, 1e-2)}
        if name not in decay:
            raise ValueError(f"Unsupported optimizer: {name!r}. Supported optimizers are 'adam' and 'adamw'.")
        cls, weight_decay = decay[name]
        # foreach: the same update rule as one multi-tensor call per step
        return cls(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay, foreach=True)

    @staticmethod
    def _restore(obj, state):
        """Best-effort ``load_state_dict`` (a checkpoint from another optimiser layout must not stop a fit)."""
        if obj is not None and state is not None:
            try:
                obj.load_state_dict(state)
            except Exception:  # noqa: BLE001, S110
                pass

    def fit(self, covariates, target, target_unc=None, iterations: int = 100, optimizer: str | None = None,
            learning_rate: float | None = None, early_stopping: bool = False, patience: int = 60,
            scheduler: bool = True, resume: bool = False, penalty_callback=None, penalty_weight: float = 0.0):
        """Train the hyperparameters; arguments and behaviour as ``MarginalGPyTorch.fit``
        (engines/gpytorch.py:162-458): Adam (default lr 0.05) or AdamW, gradient clipping to norm 1, optional
        ReduceLROnPlateau, NaN iterations skipped (more than ten in a row raise), optional early stopping, resume from a
        checkpoint (``load``) or from an interrupted call (``resume=True``), optional penalty term."""
        trained = bool(getattr(self, "model", None) is not None and getattr(self, "likelihood", None) is not None
                       and self.is_fitted)
        from_checkpoint = trained and self._resume_info is not None
        from_interruption = trained and resume and not from_checkpoint
        x, y, unc = self._attach(covariates, target, target_unc, refit_data=not from_interruption)
        restore = from_checkpoint or from_interruption
        if from_checkpoint:
            try:
                self.model.set_train_data(inputs=x, targets=y, strict=False)
            except Exception:  # noqa: BLE001
                restore = False
        if not restore:
            self._fresh_model(x, y, unc)
        self._setup_device(x, y)
        self.model.train()
        self.likelihood.train()

        # what a previous run left behind: a checkpoint's record, or the live optimiser of an interrupted call
        previous = dict(self._resume_info or {})
        if from_interruption and self._last_optimizer is not None:
            groups = self._last_optimizer.param_groups
            previous = {"optimizer_name": _get_optimizer_name(self._last_optimizer),
                        "optimizer_lr": groups[0]["lr"] if groups else None,
                        "optimizer_state_dict": self._last_optimizer.state_dict(),
                        "scheduler_state_dict": self._last_scheduler.state_dict() if self._last_scheduler else None}
        # a saved optimiser kind outranks the argument, as in the reference (engines/gpytorch.py:268-288)
        saved = (previous.get("optimizer_name") or "").lower()
        asked = optimizer if optimizer is not None else (saved or "adam")
        name = next((kind for kind in ("adamw", "adam") if kind in (asked, saved)), asked)
        lr = learning_rate if learning_rate is not None else (previous.get("optimizer_lr") or 0.05)
        params = list(self.model.parameters())
        optimizer_obj = self._new_optimizer(params, name, lr)
        scheduler_obj = None
        if scheduler:
            scheduler_obj = torch.optim.lr_scheduler.ReduceLROnPlateau(
                optimizer_obj, mode="min", factor=0.7, patience=max(20, patience // 2), threshold=1e-4,
                threshold_mode="rel", min_lr=1e-6, cooldown=10)
        if restore:
            self._restore(optimizer_obj, previous.get("optimizer_state_dict"))
            self._restore(scheduler_obj, previous.get("scheduler_state_dict"))

        first = self._current_iteration if (resume and hasattr(self, "_current_iteration")) else 0
        if iterations <= first:
            print(f"Model already trained for {first} iterations (>= target {iterations}). No further training needed.")
            return
        mll = ExactMarginalLogLikelihood(self.likelihood, self.model)
        bar = tqdm.tqdm(range(iterations - first), ncols=100, desc=f"Training {first}->{iterations}")
        best, stale, bad_in_a_row, i = float("inf"), 0, 0, 0
        use_penalty = penalty_callback is not None and penalty_weight > 0.0
        # the iteration's host algebra in closed form where the model allows it (gp/explicit.py); a penalty joins that
        # path if it brings its own closed form (``explicit_terms``: rating-gp's monotonicity penalty does), any other
        # penalty callback is an autograd expression and keeps the autograd path
        closed_penalty = use_penalty and hasattr(penalty_callback, "explicit_terms")
        explicit = (ExplicitObjective.build(self, mll._priors)
                    if self.explicit_host_algebra and (not use_penalty or closed_penalty) else None)
        try:
            for i in bar:
                self._current_iteration = first + i
                optimizer_obj.zero_grad(set_to_none=True)
                try:
                    if explicit is not None:
                        objective = torch.tensor([explicit.evaluate(
                            set_grads=False, penalty=penalty_callback if use_penalty else None,
                            penalty_weight=float(penalty_weight) if use_penalty else 0.0)], dtype=torch.float64)
                    else:
                        objective = -mll(self._prior(), self._train_y)
                except Exception:
                    bad_in_a_row += 1
                    if bad_in_a_row > 10:
                        raise
                    continue
                penalty = None
                if use_penalty and explicit is not None:
                    penalty = None if explicit.last_penalty is None else torch.tensor(explicit.last_penalty, dtype=torch.float64)
                elif use_penalty:
                    try:
                        value = penalty_callback()
                        penalty = value if torch.is_tensor(value) else None
                    except Exception:  # noqa: BLE001
                        penalty = None
                    if penalty is not None:
                        objective = objective + float(penalty_weight) * penalty.to(objective.device, objective.dtype)
                if not bool(torch.isfinite(objective).all()):
                    bad_in_a_row += 1
                    if bad_in_a_row > 10:
                        raise RuntimeError(
                            f"Encountered more than 10 consecutive NaN/Inf objectives at iteration {i + 1}")
                    continue
                bad_in_a_row = 0
                stepped = False
                if explicit is None:
                    objective.backward()
                    total_norm = _clip_grad_norm(params, 1.0)
                else:  # the gradient of all parameters is one flat array: clip it and step from it directly
                    total_norm = explicit.clip_flat_grad(1.0)
                    if math.isfinite(total_norm) and hasattr(optimizer_obj, "step_flat"):
                        stepped = optimizer_obj.step_flat(torch.from_numpy(explicit.flat_grad))
                    if not stepped:
                        explicit.assign_param_grads()
                # the reference scans every p.grad for NaN after clipping (engines/gpytorch.py:387-392); a clipped
                # gradient holds a NaN exactly when the pre-clip norm is NaN or Inf (Inf * 0 = NaN), so one scalar says it
                grads_broken = not math.isfinite(float(total_norm))
                if grads_broken:
                    for p in params:
                        if p.grad is not None:
                            p.grad = torch.nan_to_num(p.grad, nan=0.0, posinf=0.0, neginf=0.0)
                if not stepped:
                    optimizer_obj.step()
                value = float(objective.item())
                if scheduler_obj is not None:
                    scheduler_obj.step(value)
                if value < best - 1e-6:
                    best, stale = value, 0
                else:
                    stale += 1
                stop = early_stopping and stale >= patience
                if stop:
                    print(f"\nEarly stopping triggered after {i + 1} iterations")
                    print(f"Best objective: {best:.6f}")
                if not grads_broken:
                    note = f"obj={value:.4f}, lr={optimizer_obj.param_groups[0]['lr']:.1e}"
                    if penalty is not None and bool(torch.isfinite(penalty).all()):
                        note += f", pen={float(penalty.detach()):.3e}"
                    bar.set_postfix_str(note, refresh=False)  # shown at the bar's own refresh interval, not forced every iteration
                if stop:
                    break
        except KeyboardInterrupt:
            print(f"\nTraining interrupted at iteration {i + 1}")
            print(f"Best objective: {best:.6f}")
        finally:
            self.is_fitted = True
        self._last_optimizer, self._last_scheduler = optimizer_obj, scheduler_obj
        self._factor_key = None

    # ------------------------------------------------------------------ prediction
    def _device_ready(self):
        if getattr(self, "_pending_device", None) is not None:
            pending, self._pending_device = self._pending_device, None
            self._setup_device(*pending)

    def _ensure_factor(self):
        """Eval-mode cache of the reference's prediction strategy: (L, L^-1, alpha) at the current
        hyperparameters, rebuilt only when a parameter changed."""
        self._device_ready()
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
        self._device_ready()
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
        K** - V^T V with V = L^-1 K(X, X*), its Cholesky factor (our blocked HIP potrf, psd_safe_cholesky's jitter
        policy) times N(0, I) (``dgp_sample_draws``)."""
        Xnew = torch.tensor(self.dm.Xnew(covariates), dtype=self.dtype).to(self.device).contiguous()
        self._device_ready()
        self.model.eval()
        self.likelihood.eval()
        with torch.no_grad():
            if hasattr(self.model, "prepare_eval"):
                self.model.prepare_eval(self._train_x, Xnew)
            self._ensure_factor()
            mean, factor, _jitter = self._plan.posterior_factor(self._factor_theta, Xnew)
            mean = mean + self.model.prior_mean(Xnew)
            sim = self._plan.sample_draws(factor, Xnew.shape[0], mean, n)  # (n, m) = mean + (L z)^T, one HIP launch
        temp = self.dm.y_t_device(sim.reshape(-1))
        data = np.asarray(temp.data).reshape(n, -1)
        return DataArray(data, coords=dict(covariates.coords, draw=np.arange(n)),
                         dims=["draw"] + list(covariates.coords), attrs=temp.attrs)

    def build_model(self, X, y, **kwargs):
        raise NotImplementedError("This method must be implemented in a subclass")
