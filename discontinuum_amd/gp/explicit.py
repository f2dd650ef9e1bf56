"""The host algebra of one training iteration in closed form.

What ``ExactMarginalLogLikelihood`` + autograd do per iteration on the host -- constraint transforms, the
hyperparameter vector, the log-prior and the chain rule back to the raw parameters -- is about seventy scalar-sized
autograd nodes (0.35 ms forward + backward at n = 300, more than the device step they wait for).  For the models the
engine ships every piece has a closed form:

    x_k = T_k(raw_k)          softplus (+ lower bound), interval, or identity
    theta = x[theta_src]      the hyperparameter vector is a gather of constrained values
    log p = sum over Gamma / HalfNormal / Normal priors, each on ONE constrained value
    objective = (nll(theta, r, noise) - log p) / n

so d objective / d raw = (scatter(d nll / d theta) - d log p / d x + mean / noise terms) / n * T_k'(raw_k), a few numpy
vector operations.  ``ExplicitObjective.build`` DISCOVERS this structure from the live model (it perturbs the raw
parameters to distinct values and matches the hyperparameter vector, the priors' inputs and the mean shortcut's
parameters against the transformed values); anything it cannot match makes it return None and the engine keeps the
autograd path.  The two paths are compared in tests/test_engine_cpu.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib
from .constraints import GreaterThan, Interval, Positive
from .mll import NotPSDError
from .priors import GammaPrior, HalfNormalPrior, NormalPrior

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _softplus(v):
    return np.logaddexp(0.0, v)


def _sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


class ExplicitObjective:
    def __init__(self):
        raise TypeError("use ExplicitObjective.build")

    # ------------------------------------------------------------------ discovery
    @classmethod
    def build(cls, engine, priors):
        """``priors`` = [(prior, closure, module)] as ``ExactMarginalLogLikelihood`` holds them.  Returns None if the
        model's host algebra is not of the closed form above."""
        try:
            return cls._build(engine, priors)
        except _Unsupported:
            return None

    @classmethod
    def _build(cls, engine, priors):
        shortcut = engine._mean_shortcut()
        if shortcut is None:
            raise _Unsupported("no host-side mean / noise shortcut")
        params = list(engine.model.parameters())
        if any(p.dtype != torch.float64 or p.device.type != "cpu" for p in params):
            raise _Unsupported("host parameters must be float64 CPU tensors")
        if not all(p.requires_grad for p in params):
            raise _Unsupported("frozen parameters: autograd knows which gradients to skip")
        # constraint of every parameter (None: used as it is)
        constraint = {}
        for mod in engine.model.modules():
            for name, p in mod._parameters.items():
                if p is not None and name.startswith("raw_") and hasattr(mod, name + "_constraint"):
                    constraint[id(p)] = getattr(mod, name + "_constraint")
        self = object.__new__(cls)
        self.engine, self.shortcut, self.params = engine, shortcut, params
        self.sizes = [int(p.numel()) for p in params]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        total = int(self.offsets[-1])
        kind = np.zeros(total, dtype=np.int64)  # 0 identity, 1 softplus (+ lower bound), 2 interval
        self.lower, self.span = np.zeros(total), np.ones(total)
        for k, p in enumerate(params):
            sl = slice(int(self.offsets[k]), int(self.offsets[k + 1]))
            c = constraint.get(id(p))
            if c is None:
                continue
            if type(c) in (Positive, GreaterThan):
                kind[sl], self.lower[sl] = 1, float(c.lower_bound)
            elif type(c) is Interval:
                kind[sl], self.lower[sl] = 2, float(c.lower_bound)
                self.span[sl] = float(c.upper_bound) - float(c.lower_bound)
            else:
                raise _Unsupported(f"constraint {type(c).__name__}")
        self.is_softplus, self.is_interval = kind == 1, kind == 2

        # perturb the raw values so that every constrained value is distinct, then match by value
        saved = [p.detach().clone() for p in params]
        gen = torch.Generator().manual_seed(20240229)
        try:
            with torch.no_grad():
                for p in params:
                    p.copy_(0.3 + torch.rand(p.shape, generator=gen, dtype=p.dtype))
                # the model's OWN transforms give the values to match against (bit-equal to what the hyperparameter
                # vector and the priors see); the numpy forms above are only used at run time
                where, i = {}, 0
                for p in params:
                    c = constraint.get(id(p))
                    for v in (p if c is None else c.transform(p)).detach().reshape(-1).tolist():
                        if v in where:
                            raise _Unsupported("constrained values collide")
                        where[v] = i
                        i += 1

                def locate(t, what):
                    out = []
                    for v in torch.as_tensor(t).detach().reshape(-1).tolist():
                        if v not in where:
                            raise _Unsupported(f"{what} is not a constrained parameter value")
                        out.append(where[v])
                    return np.asarray(out, dtype=np.int64)

                self.theta_src = locate(engine._theta_fn(), "a hyperparameter")
                groups = {"gamma": [[], [], []], "half_normal": [[], []], "normal": [[], [], []]}
                for prior, closure, mod in priors:
                    idx = locate(closure(mod), "a prior's argument")
                    rep = np.ones(len(idx))
                    if type(prior) is GammaPrior:
                        g = groups["gamma"]
                        g[0].append(idx), g[1].append(float(prior.concentration) * rep), g[2].append(float(prior.rate) * rep)
                    elif type(prior) is HalfNormalPrior:
                        g = groups["half_normal"]
                        g[0].append(idx), g[1].append(float(prior.scale) * rep)
                    elif type(prior) is NormalPrior:
                        g = groups["normal"]
                        g[0].append(idx), g[1].append(float(prior.loc) * rep), g[2].append(float(prior.scale) * rep)
                    else:
                        raise _Unsupported(f"prior {type(prior).__name__}")
                self.shortcut_src = locate(torch.stack([torch.as_tensor(v).reshape(()) for v in shortcut.param_values()]),
                                           "a mean / noise parameter")
        finally:
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
        cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)  # noqa: E731
        g = groups["gamma"]
        self.gamma = (cat(g[0]).astype(np.int64), cat(g[1]), cat(g[2]))
        h = groups["half_normal"]
        self.half_normal = (cat(h[0]).astype(np.int64), cat(h[1]))
        m = groups["normal"]
        self.normal = (cat(m[0]).astype(np.int64), cat(m[1]), cat(m[2]))
        a, b = self.gamma[1], self.gamma[2]
        self.lp_const = float(np.sum(a * np.log(b) - np.vectorize(math.lgamma)(a))) if len(a) else 0.0
        self.lp_const += float(np.sum(math.log(2.0) - np.log(self.half_normal[1]) - _HALF_LOG_2PI))
        self.lp_const += float(np.sum(-np.log(self.normal[2]) - _HALF_LOG_2PI))
        self.views = [p.detach().numpy().reshape(-1) for p in params]  # the parameters' own memory
        self.ntheta = len(self.theta_src)
        return self

    # ------------------------------------------------------------------ evaluation
    def _raw(self):
        if self.views[0].__array_interface__["data"][0] != self.params[0].data_ptr():  # ``p.data`` was reassigned
            self.views = [p.detach().numpy().reshape(-1) for p in self.params]
        return np.concatenate(self.views)

    def _transform(self, raw):
        """Constrained values and d x / d raw."""
        x, slope = raw.copy(), np.ones_like(raw)
        sp, iv = self.is_softplus, self.is_interval
        if sp.any():
            x[sp] = _softplus(raw[sp]) + self.lower[sp]
            slope[sp] = _sigmoid(raw[sp])
        if iv.any():
            s = _sigmoid(raw[iv])
            x[iv] = self.lower[iv] + self.span[iv] * s
            slope[iv] = self.span[iv] * s * (1.0 - s)
        return x, slope

    def _log_prior(self, x):
        """log p and d log p / d x."""
        lp, g = self.lp_const, np.zeros_like(x)
        idx, a, b = self.gamma
        if len(idx):
            v = x[idx]
            lp += float(np.sum((a - 1.0) * np.log(v) - b * v))
            np.add.at(g, idx, (a - 1.0) / v - b)
        idx, s = self.half_normal
        if len(idx):
            v = x[idx] / s
            lp -= 0.5 * float(np.sum(v * v))
            np.add.at(g, idx, -v / s)
        idx, loc, s = self.normal
        if len(idx):
            v = (x[idx] - loc) / s
            lp -= 0.5 * float(np.sum(v * v))
            np.add.at(g, idx, -v / s)
        return lp, g

    def _row(self, theta, r, noise):
        """The fit step's result row on the host, with the reference's jitter policy for a matrix that is not p.d."""
        plan = self.engine._plan
        jitter0 = 1e-8 if plan.dtype == torch.float64 else 1e-6
        for attempt in range(4):
            out = plan.fit_step(theta, r, noise if attempt == 0 else noise + jitter0 * 10 ** (attempt - 1))[0]
            if attempt == 0:
                yield None  # the device runs: the caller does its own host work now, then asks for the row
            row = out.to("cpu", torch.float64).numpy()
            if int(row[_lib.OUT_INFO]) == 0:
                if attempt:
                    import warnings

                    warnings.warn(f"A not p.d., added jitter of {jitter0 * 10 ** (attempt - 1):.1e} to the diagonal",
                                  RuntimeWarning, stacklevel=3)
                yield row
                return
        raise NotPSDError(f"Matrix not positive definite: Cholesky pivot {int(row[_lib.OUT_INFO])} is not positive")

    def evaluate(self, set_grads=True, penalty=None, penalty_weight=0.0):
        """One objective evaluation: launches the device step, leaves d objective / d raw of all parameters in
        ``flat_grad`` (and as ``p.grad`` of every parameter unless ``set_grads`` is False), returns the value (a
        float).  Same arithmetic as ``-mll(model(train_x), train_y)`` + ``backward()``.

        ``penalty``: an object with ``explicit_terms(objective, x) -> (value, d value / d x)`` in the space of the
        constrained values ``x`` (rating-gp's monotonicity penalty): ``penalty_weight * value`` joins the objective and
        its gradient the flat gradient -- evaluated right after the fit step, on the factorisation it left in the plan.
        A penalty that raises is skipped for this evaluation (the autograd loop does the same); ``last_penalty`` keeps
        the value (None if skipped)."""
        eng = self.engine
        with torch.no_grad():
            r, noise = self.shortcut.residual_and_noise(eng._plan, eng._train_y)
            x, slope = self._transform(self._raw())
            steps = self._row(x[self.theta_src].tolist(), r, noise)
            next(steps)                     # launched
            lp, dlp = self._log_prior(x)    # ... and the host's share runs under it
            row = next(steps)
        n = float(eng._train_y.shape[0])
        gx = -dlp
        np.add.at(gx, self.theta_src, row[_lib.OUT_DTHETA:_lib.OUT_DTHETA + self.ntheta])
        np.add.at(gx, self.shortcut_src, np.asarray([float(v) for v in self.shortcut.grads(row)]))
        self.flat_grad = gx * slope / n  # d objective / d raw, all parameters, in ``params`` order
        value = (float(row[_lib.OUT_NLL]) - lp) / n
        self.last_penalty = None
        if penalty is not None and penalty_weight:
            try:
                pv, pgx = penalty.explicit_terms(self, x)
            except Exception:  # noqa: BLE001
                pv = None
            if pv is not None:
                self.last_penalty = float(pv)
                value += float(penalty_weight) * float(pv)
                self.flat_grad = self.flat_grad + float(penalty_weight) * pgx * slope
        if set_grads:
            self.assign_param_grads()
        return value

    def assign_param_grads(self):
        """``p.grad`` of every parameter as views of the flat gradient of the last ``evaluate``."""
        for k, p in enumerate(self.params):
            p.grad = torch.from_numpy(self.flat_grad[self.offsets[k]:self.offsets[k + 1]].reshape(tuple(p.shape)))

    def clip_flat_grad(self, max_norm):
        """``clip_grad_norm_(params, max_norm)`` on the flat gradient; returns the norm before clipping."""
        total = float(np.sqrt(np.dot(self.flat_grad, self.flat_grad)))
        coef = max_norm / (total + 1e-6)
        if not coef >= 1.0:
            self.flat_grad *= coef
        return total


class _Unsupported(Exception):
    pass
