"""Train MANY sites at once on one GPU: ``fit_many``.

The reference trains one site per process (``model.fit`` inside a cloud map,
``examples/nwqn-loadest-example/nwqn-loadest-example.py:120-159``).  Here the sites share one batched device plan
(``GPPlan(batch=B)``, ragged sizes allowed): every training iteration is ONE batched fit step on the GPU and ONE
vectorised evaluation of the host-side algebra (constraints, priors, the constant mean) over all sites
(``torch.func.vmap`` over the stacked parameters of the per-site models), followed by the reference's optimiser
step applied per site:

* Adam(lr, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-4)            (``engines/gpytorch.py:268-286``)
* gradient clipping to norm 1.0 PER SITE                                 (``:387``)
* ReduceLROnPlateau(min, factor 0.7, patience max(20, patience // 2), threshold 1e-4 rel, cooldown 10, min_lr 1e-6)
  PER SITE                                                               (``:296-305``)
* an iteration whose objective is NaN / Inf (or whose matrix is not positive definite) is skipped for that site,
  more than 10 in a row raise                                            (``:352-379``)

so each site follows the trajectory ``model.fit`` would give it (tests/test_gpu_engine.py compares them).  Afterwards
every model holds its fitted parameters and is ready for ``predict``.  Both model families are covered: loadest-gp
(constant mean, fixed noise) and rating-gp (power-law mean whose parameters the reference clamps in place on every
forward -- done here once per iteration on the stacked parameters -- and a learned homoskedastic noise term).
Early stopping is per site; the rating-gp monotonicity penalty (``monotonic_penalty_weight``) differentiates every
site's posterior mean at its own random grid through ONE batched ``dgp_predict_mean`` / ``dgp_mean_vjp`` per iteration;
``return_state=True`` / ``resume=state`` continue a run where it stopped (optimiser moments, schedules, counters);
``penalty_callback`` / ``penalty_weight`` add the reference's generic penalty term per site (``:362-373``).

Across the GPUs of a node: ``fit_many_distributed`` -- site i trains on rank i mod G (``sites.site_partition``), every rank
runs ``fit_many`` on its share on its own GPU, and ONE collective at the end of the fit (an ``all_gather`` of the fitted raw
parameters, final objectives, iteration counts and stop reasons; RCCL over xGMI, gloo in the CPU tests) hands every rank
the whole table, so any rank can ``predict`` any site.  No per-iteration traffic ("or only at the end
of the fit"); the reference's fan-out is ``examples/nwqn-loadest-example/nwqn-loadest-example.py:38-125, 156-159``.
"""
from __future__ import annotations

import time

import torch
from torch import nn
from torch.func import functional_call, vmap

from . import _lib
from .backend import GPPlan
from .gp.kernels import prior_closures
from .gp.lowering import lower


LAST_TIMING: dict = {}  # diagnostics of the most recent fit_many call in this process (set-up / loop seconds, iterations)


class _HostSide(nn.Module):
    """theta (P,), log-prior, constant mean of ONE site as a function of its module parameters."""

    def __init__(self, model, likelihood, d):
        super().__init__()
        self.model = model
        self.name, self._theta = lower(model.covar_module, d)
        self._priors = prior_closures(model)
        if likelihood is not None and not any(likelihood is m for m in model.modules()):
            self.likelihood = likelihood
            self._priors += prior_closures(likelihood)

    def forward(self):
        lp = torch.zeros((), dtype=torch.float64)
        for prior, closure, mod in self._priors:
            lp = lp + prior.log_prob(closure(mod)).sum()
        if self.name == "rating":  # power law a, b, c and the learned noise term
            pw = self.model.powerlaw
            extras = torch.cat([pw.a.reshape(1), pw.b.reshape(1), pw.c.reshape(1),
                                self.model.likelihood.second_noise.reshape(1)])
        else:
            extras = self.model.mean_module.constant.reshape(1)
        return self._theta(), lp, extras


class _BatchedNLL(torch.autograd.Function):
    """Data terms of B sites from one batched ``dgp_fit_step``; sites that fail come back as NaN with zero gradient.
    The residual and noise vectors arrive without autograd; the gradients of the mean / noise parameters
    (``extras``, (B, E) on the host) are read off the reductions in each site's result row -- like the single-site
    engine (``engines/hip.py::MeanShortcut``): no device-side autograd, one device->host copy per iteration."""

    @staticmethod
    def forward(ctx, plan, theta, r, noise, extras, family):
        if plan.batch == 1:
            out = plan.fit_step(theta[0], r[0].contiguous(), noise[0].contiguous())[0].reshape(1, -1)
        else:
            out = plan.fit_step(theta, r, noise)[0]
        host = out.to("cpu", torch.float64)
        _ok, nll, dtheta, dextras = _row_terms(host, plan.ntheta, family, extras[:, 1].detach() if family == "rating" else None)
        ctx.save_for_backward(dtheta, dextras)
        ctx.dtypes = (theta.dtype, extras.dtype)
        return nll

    @staticmethod
    def backward(ctx, g):
        dtheta, dextras = ctx.saved_tensors
        g = torch.nan_to_num(g, nan=0.0)[:, None]
        return None, (dtheta * g).to(ctx.dtypes[0]), None, None, (dextras * g).to(ctx.dtypes[1]), None


class _BatchedGPMean(torch.autograd.Function):
    """K(x*_b, X_b) alpha_b of every site b at its own points (B, m, d) from the factorisation the batched plan holds
    (the fit step of the same iteration), differentiable: backward runs ONE batched ``dgp_mean_vjp`` and turns its
    dr / dnoise vectors into the gradients of the mean / noise parameters (``extras``) like ``_BatchedNLL`` does with the
    result row's reductions.  Used by the monotonicity penalty (src/rating_gp/models/gpytorch.py:130-187)."""

    @staticmethod
    def forward(ctx, plan, theta, xgrid, extras, family, dr_weights, valid):
        one = plan.batch == 1
        th = theta.detach()
        mu = plan.predict_mean(th[0] if one else th, xgrid[0].contiguous() if one else xgrid)
        ctx.plan, ctx.family, ctx.one = plan, family, one
        ctx.save_for_backward(th, xgrid, extras.detach(), dr_weights if dr_weights is not None else torch.empty(0), valid)
        ctx.dtypes = (theta.dtype, extras.dtype)
        return mu.reshape(xgrid.shape[0], -1).to("cpu", torch.float64)

    @staticmethod
    def backward(ctx, g):
        th, xgrid, extras, dr_w, valid = ctx.saved_tensors
        plan, one = ctx.plan, ctx.one
        w = torch.nan_to_num(g, nan=0.0).to(xgrid.device, xgrid.dtype).contiguous()
        dtheta, dr, dnoise = plan.mean_vjp(th[0] if one else th, xgrid[0].contiguous() if one else xgrid, w[0].contiguous() if one else w)
        B = xgrid.shape[0]
        dtheta, dr, dnoise = dtheta.reshape(B, -1), dr.reshape(B, -1) * valid, dnoise.reshape(B, -1) * valid
        sum_dr = dr.sum(dim=1)
        if ctx.family == "rating":  # r = y - a - b log(s - c); Sigma = fixed + second_noise
            w3 = dr_w.reshape(B, 2, -1)
            dextras = torch.stack([-sum_dr, -(dr * w3[:, 0]).sum(dim=1), extras[:, 1].to(dr) * (dr * w3[:, 1]).sum(dim=1),
                                   dnoise.sum(dim=1)], dim=1)
        else:
            dextras = -sum_dr[:, None]
        return (None, dtheta.to("cpu", ctx.dtypes[0]), None, dextras.to("cpu", ctx.dtypes[1]), None, None, None)


def _row_terms(host, ntheta, family, extras_b):
    """What a batch of result rows (B, OUT_LEN; float64, host) says: ok mask, NLL (NaN where a site failed), dNLL/dtheta
    (B, P) and dNLL/d(mean / noise parameters) (B, E) -- zeros for failed sites.  ``extras_b`` = the current power-law b
    per site (rating-gp: mu = a + b log(s - c), so d mu / d c carries a factor b); shared by both host-algebra paths."""
    ok = (host[:, _lib.OUT_INFO] == 0) & torch.isfinite(host[:, _lib.OUT_NLL])
    clean = torch.nan_to_num(host, nan=0.0, posinf=0.0, neginf=0.0) * ok[:, None]
    dtheta = clean[:, _lib.OUT_DTHETA:_lib.OUT_DTHETA + ntheta]
    sum_dr = clean[:, _lib.OUT_SUM_DR]
    if family == "rating":  # (a, b, c, second_noise): mu = a + b log(s - c), Sigma = fixed + second_noise
        w0 = _lib.OUT_DR_W0
        dextras = torch.stack([-sum_dr, -clean[:, w0], extras_b * clean[:, w0 + 1], clean[:, _lib.OUT_SUM_DNOISE]], dim=1)
    else:                   # (c,): mu = c
        dextras = -sum_dr[:, None]
    nll = torch.where(ok, host[:, _lib.OUT_NLL], torch.full_like(host[:, 0], float("nan")))
    return ok, nll, dtheta, dextras


class _Unsupported(Exception):
    pass


class _ClosedForm:
    """The host algebra of one ``fit_many`` iteration in CLOSED FORM for all sites at once -- ``gp/explicit.py`` (the
    single-site engine's fast path) vectorised over the sites.  With every raw parameter of every site in ONE (B, T) array:

        x = T(raw)                  softplus (+ lower bound), interval (bounds per site: the rating gate), or identity
        theta = x[:, theta_src]     the hyperparameter vectors are a gather of constrained values
        extras = x[:, extras_src]   the mean / noise parameters likewise
        log p = sum over Gamma / HalfNormal / Normal priors, each on ONE constrained value
        d objective / d raw = (scatter(dNLL/dtheta) + scatter(dNLL/dextras) - d log p / d x) T'(raw) / n

    -- a dozen numpy operations on (B, T) arrays instead of ``torch.func.vmap`` over the module trees plus an autograd
    backward (1.3 ms of a 4 ms iteration for 256 sites of n = 300), and the log-prior runs UNDER the device step.  ``build``
    DISCOVERS the structure from site 0's live modules (perturb the raw parameters to distinct values, match the
    hyperparameter vector / prior arguments / mean-noise parameters against the transformed values) and checks that every
    other site has the same tree, constraint types and prior constants; anything it cannot match returns None and
    ``fit_many`` keeps the autograd path -- which is also what it is tested against (tests/test_engine_cpu.py, to 1e-12)."""

    @classmethod
    def build(cls, hosts, own_params, family):
        try:
            return cls(hosts, own_params, family)
        except _Unsupported:
            return None

    def __init__(self, hosts, own_params, family):
        import math

        import numpy as np

        from .gp.constraints import GreaterThan, Interval, Positive
        from .gp.priors import GammaPrior, HalfNormalPrior, NormalPrior

        self.np = np
        host0, B = hosts[0], len(hosts)
        self.names = list(own_params[0])
        shapes = [tuple(own_params[0][k].shape) for k in self.names]
        if any(own[k].dtype != torch.float64 or own[k].device.type != "cpu" or not own[k].requires_grad for own in own_params for k in self.names):
            raise _Unsupported("host parameters must be trainable float64 CPU tensors")
        if any(list(own) != self.names or [tuple(own[k].shape) for k in self.names] != shapes for own in own_params):
            raise _Unsupported("the sites' parameter trees differ")
        self.shapes = shapes
        self.sizes = [int(np.prod(sh)) if sh else 1 for sh in shapes]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        T = int(self.offsets[-1])

        def constraints_by_name(h):
            out = {}
            for mname, mod in h.named_modules():
                for pname, par in mod._parameters.items():
                    if par is not None and pname.startswith("raw_") and hasattr(mod, pname + "_constraint"):
                        out.setdefault((mname + "." if mname else "") + pname, getattr(mod, pname + "_constraint"))
            return out

        kind = np.zeros(T, dtype=np.int64)  # 0 identity, 1 softplus (+ lower bound), 2 interval
        self.lower, self.span = np.zeros((B, T)), np.ones((B, T))
        for b, h in enumerate(hosts):
            cons = constraints_by_name(h)
            for k, name in enumerate(self.names):
                sl = slice(int(self.offsets[k]), int(self.offsets[k + 1]))
                c = cons.get(name)
                code = 0 if c is None else (1 if type(c) in (Positive, GreaterThan) else (2 if type(c) is Interval else -1))
                if code < 0:
                    raise _Unsupported(f"constraint {type(c).__name__}")
                if b == 0:
                    kind[sl] = code
                elif int(kind[sl][0]) != code:
                    raise _Unsupported("the sites' constraints differ")
                if code:
                    self.lower[b, sl] = float(c.lower_bound)
                if code == 2:
                    self.span[b, sl] = float(c.upper_bound) - float(c.lower_bound)
        self.is_softplus, self.is_interval = kind == 1, kind == 2

        # ---- discovery on site 0: distinct raw values, match by value
        cons0 = constraints_by_name(host0)
        par0 = [own_params[0][k] for k in self.names]
        saved = [q.detach().clone() for q in par0]
        gen = torch.Generator().manual_seed(20240229)
        try:
            with torch.no_grad():
                for q in par0:
                    q.copy_(0.3 + torch.rand(q.shape, generator=gen, dtype=q.dtype))
                where, i = {}, 0
                for name, q in zip(self.names, par0):
                    c = cons0.get(name)
                    for v in (q if c is None else c.transform(q)).detach().reshape(-1).tolist():
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

                theta, _lp, extras = host0()
                self.theta_src = locate(theta, "a hyperparameter")
                self.extras_src = locate(extras, "a mean / noise parameter")
                groups = {"gamma": [[], [], []], "half_normal": [[], []], "normal": [[], [], []]}
                signature = []
                for prior, closure, mod in host0._priors:
                    idx = locate(closure(mod), "a prior's argument")
                    rep = np.ones(len(idx))
                    if type(prior) is GammaPrior:
                        g = groups["gamma"]
                        g[0].append(idx), g[1].append(float(prior.concentration) * rep), g[2].append(float(prior.rate) * rep)
                        signature.append(("g", float(prior.concentration), float(prior.rate)))
                    elif type(prior) is HalfNormalPrior:
                        g = groups["half_normal"]
                        g[0].append(idx), g[1].append(float(prior.scale) * rep)
                        signature.append(("h", float(prior.scale)))
                    elif type(prior) is NormalPrior:
                        g = groups["normal"]
                        g[0].append(idx), g[1].append(float(prior.loc) * rep), g[2].append(float(prior.scale) * rep)
                        signature.append(("n", float(prior.loc), float(prior.scale)))
                    else:
                        raise _Unsupported(f"prior {type(prior).__name__}")
        finally:
            with torch.no_grad():
                for q, v in zip(par0, saved):
                    q.copy_(v)
        for h in hosts[1:]:  # the same priors with the same constants on every site (their ARGUMENTS follow from the same tree)
            sig = []
            for prior, _closure, _mod in h._priors:
                if type(prior) is GammaPrior:
                    sig.append(("g", float(prior.concentration), float(prior.rate)))
                elif type(prior) is HalfNormalPrior:
                    sig.append(("h", float(prior.scale)))
                elif type(prior) is NormalPrior:
                    sig.append(("n", float(prior.loc), float(prior.scale)))
                else:
                    raise _Unsupported(f"prior {type(prior).__name__}")
            if sig != signature:
                raise _Unsupported("the sites' priors differ")
        cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)  # noqa: E731
        g = groups["gamma"]
        self.gamma = (cat(g[0]).astype(np.int64), cat(g[1]), cat(g[2]))
        hn = groups["half_normal"]
        self.half_normal = (cat(hn[0]).astype(np.int64), cat(hn[1]))
        nm = groups["normal"]
        self.normal = (cat(nm[0]).astype(np.int64), cat(nm[1]), cat(nm[2]))
        a, bb = self.gamma[1], self.gamma[2]
        half_log_2pi = 0.5 * math.log(2.0 * math.pi)
        self.lp_const = float(np.sum(a * np.log(bb) - np.vectorize(math.lgamma)(a))) if len(a) else 0.0
        self.lp_const += float(np.sum(math.log(2.0) - np.log(self.half_normal[1]) - half_log_2pi))
        self.lp_const += float(np.sum(-np.log(self.normal[2]) - half_log_2pi))
        self.T = T

    # ---- stacked dict of named (B, ...) tensors  <->  one (B, T) tensor
    def flatten(self, named):
        return torch.cat([named[k].detach().reshape(named[k].shape[0], -1).to(torch.float64) for k in self.names], dim=1).contiguous()

    def unflatten(self, flat):
        B = flat.shape[0]
        return {k: flat[:, int(self.offsets[i]):int(self.offsets[i + 1])].reshape((B,) + self.shapes[i]).clone()
                for i, k in enumerate(self.names)}

    def column(self, name):
        i = self.names.index(name)
        if self.sizes[i] != 1:
            raise _Unsupported(f"{name} is not a scalar parameter")
        return int(self.offsets[i])

    def transform(self, raw):
        """(x, d x / d raw), both (B, T), from the raw values (numpy (B, T))."""
        np = self.np
        x, slope = raw.copy(), np.ones_like(raw)
        sp, iv = self.is_softplus, self.is_interval
        if sp.any():
            r = raw[:, sp]
            x[:, sp] = np.logaddexp(0.0, r) + self.lower[:, sp]
            slope[:, sp] = 1.0 / (1.0 + np.exp(-r))
        if iv.any():
            sg = 1.0 / (1.0 + np.exp(-raw[:, iv]))
            x[:, iv] = self.lower[:, iv] + self.span[:, iv] * sg
            slope[:, iv] = self.span[:, iv] * sg * (1.0 - sg)
        return x, slope

    def log_prior(self, x):
        """(log p (B,), d log p / d x (B, T))."""
        np = self.np
        lp, g = np.full(x.shape[0], self.lp_const), np.zeros_like(x)
        idx, a, b = self.gamma
        for j in range(len(idx)):  # (a handful of priors: the column loop keeps repeated indices correct)
            v = x[:, idx[j]]
            lp += (a[j] - 1.0) * np.log(v) - b[j] * v
            g[:, idx[j]] += (a[j] - 1.0) / v - b[j]
        idx, sc = self.half_normal
        for j in range(len(idx)):
            v = x[:, idx[j]] / sc[j]
            lp -= 0.5 * v * v
            g[:, idx[j]] -= v / sc[j]
        idx, loc, sc = self.normal
        for j in range(len(idx)):
            v = (x[:, idx[j]] - loc[j]) / sc[j]
            lp -= 0.5 * v * v
            g[:, idx[j]] -= v / sc[j]
        return lp, g

    def raw_gradient(self, dlp, slope, dtheta, dextras, nvec, ok):
        """d objective / d raw (B, T) for the sites in ``ok`` (zeros elsewhere): objective = (nll - log p) / n."""
        np = self.np
        gx = -dlp
        for j, src in enumerate(self.theta_src):
            gx[:, src] += dtheta[:, j]
        for j, src in enumerate(self.extras_src):
            gx[:, src] += dextras[:, j]
        return np.where(ok[:, None], gx * slope / nvec[:, None], 0.0)


class FitManyState:
    """Everything ``fit_many`` needs to continue a run: stacked raw parameters, Adam moments and step counts, the
    per-site learning rates and plateau / early-stopping counters (plain tensors: ``torch.save`` / ``torch.load`` with
    ``weights_only=True`` round-trips ``state.as_dict()``)."""

    FIELDS = ("params", "m1", "m2", "step", "lr", "best", "num_bad", "cooldown", "es_best", "stale", "live", "last_obj",
              "last_iteration", "iterations_done", "nan_run")

    def __init__(self, **kw):
        for k in self.FIELDS:
            if k == "nan_run" and k not in kw:  # a state saved before the counter travelled: start it at zero
                kw[k] = torch.zeros_like(kw["step"], dtype=torch.float64)
            setattr(self, k, kw[k])

    def as_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS}

    @classmethod
    def from_dict(cls, d):
        return cls(**{k: d[k] for k in cls.FIELDS if k in d})


def _fresh_seeded(m, tx, ty, tu, seed):
    """``m._fresh_model`` -- under its own seed when one is given (the global RNG is left where it was)."""
    if seed is None:
        return m._fresh_model(tx, ty, tu)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(int(seed))
        return m._fresh_model(tx, ty, tu)


def _per_site_clip(raw_grads: dict, B: int, max_norm: float = 1.0):
    """``clip_grad_norm_(max_norm)`` followed by the reference's NaN scan (engines/gpytorch.py:387-400), per site, on
    stacked gradients (leading dimension B).  The norm is taken of the RAW gradient, like the reference and
    ``MarginalHIP.fit`` do: one NaN / Inf component makes a site's norm non-finite, the clip then turns every component
    of that site's gradient into NaN and the scan zeroes them all -- the site steps on weight decay only.
    -> (coef (B,), sanitised gradients); the clipped gradient of a parameter is ``grads[k] * coef`` per site."""
    sq = sum((g.reshape(B, -1) ** 2).sum(dim=1) for g in raw_grads.values())
    broken = ~torch.isfinite(sq)
    coef = torch.clamp(max_norm / (torch.sqrt(torch.where(broken, torch.ones_like(sq), sq)) + 1e-6), max=1.0)
    coef = torch.where(broken, torch.zeros_like(coef), coef)
    grads = {k: torch.nan_to_num(g, nan=0.0, posinf=0.0, neginf=0.0) for k, g in raw_grads.items()}
    return coef, grads


def fit_many(models, datasets, iterations: int = 100, learning_rate: float = 0.05, patience: int = 60,
             scheduler: bool = True, progress: bool = False, early_stopping: bool = False,
             monotonic_penalty_weight: float = 0.0, grid_size: int = 64, monotonic_penalty_interval: int = 1,
             resume: FitManyState | None = None, return_state: bool = False, generator: torch.Generator | None = None,
             optimizer: str = "adam", penalty_callback=None, penalty_weight: float = 0.0, site_seeds=None,
             closed_form: bool = True, _penalty_uniforms=None):
    """Fit ``models[i]`` to ``datasets[i] = (covariates, target[, target_unc])`` for all i at once.  Returns the
    per-site final objectives (a float64 tensor) -- with ``return_state=True`` the pair (objectives, ``FitManyState``);
    the models are updated in place (``is_fitted``, parameters, device state).

    ``monotonic_penalty_weight`` > 0 (rating-gp only) adds ``weight * mean(relu(-d mu / d stage))`` of each site's
    posterior mean on a fresh random grid of ``grid_size`` points per iteration (every ``monotonic_penalty_interval``-th
    iteration, weighted by the interval), exactly the term of ``RatingGP.fit`` (src/rating_gp/models/gpytorch.py:130-187),
    for all sites through one batched launch sequence.  ``resume``: the state a previous call returned (same sites, same
    order) -- ``iterations`` more iterations from there; ``generator`` seeds the penalty grids.  ``optimizer``: "adam"
    (L2 weight decay 1e-4, the reference's default) or "adamw" (decoupled decay 1e-2), as ``MarginalHIP.fit``
    (engines/gpytorch.py:268-286); the default learning rate argument applies to both.

    ``penalty_callback`` / ``penalty_weight``: the reference's optional penalty term (engines/gpytorch.py:362-373:
    ``objective = nll + penalty_weight * penalty_callback()``) for many sites.  The reference's callback is a closure over
    ONE engine; here the sites' parameters live stacked in this loop, so the callback receives them:
    ``penalty_callback(site_index, params)`` with ``params`` = {name: that site's raw parameter tensor}, names as in
    ``models[i].model.named_parameters()`` prefixed with ``model.`` (likelihood parameters: ``likelihood.``), each a
    differentiable view -- it returns a scalar tensor (anything torch can differentiate w.r.t. those tensors).  Same
    tolerance as the reference: an exception or a non-tensor result drops the term for that site and iteration.

    ``closed_form`` (default True): the host algebra of an iteration -- constraints, hyperparameter vectors, log-priors and
    the chain rule back to the raw parameters -- in closed form for all sites at once (``_ClosedForm``, the single-site
    engine's ``gp/explicit.py`` vectorised) whenever the models allow it and no penalty term needs autograd; False (or a
    model it cannot match) keeps ``torch.func.vmap`` over the module trees + autograd, which gives the same trajectory
    (tests/test_engine_cpu.py: parameters to 1e-12).

    ``site_seeds`` (one int per site): site b's model is built under ``torch.manual_seed(site_seeds[b])`` (inside a forked
    RNG scope) -- the rating-gp power law and gate start from random draws like the reference's
    (src/rating_gp/models/gpytorch.py:30-33, kernels.py:276), so without seeds a site's trajectory depends on how many
    sites were built before it; with them it depends on the site alone (``fit_many_distributed`` relies on that)."""
    if site_seeds is not None and len(site_seeds) != len(models):
        raise ValueError("site_seeds needs one seed per model")
    t_enter = time.perf_counter()
    if optimizer not in ("adam", "adamw"):
        raise ValueError(f"Unsupported optimizer: {optimizer!r}. Supported optimizers are 'adam' and 'adamw'.")
    if len(models) != len(datasets) or not models:
        raise ValueError("fit_many needs one (covariates, target) pair per model")
    B = len(models)
    dtype, device = models[0].dtype, torch.device(models[0].device)
    xs, ys, hosts = [], [], []
    for m, record in zip(models, datasets):
        cov, tgt, unc = (tuple(record) + (None,))[:3]
        tx, ty, tu = m._attach(cov, tgt, unc)
        _fresh_seeded(m, tx, ty, tu, None if site_seeds is None else site_seeds[len(xs)])
        _set_training(m.model, True)
        _set_training(m.likelihood, True)
        xs.append(tx)
        ys.append(ty)
        hosts.append(_HostSide(m.model, m.likelihood, tx.shape[1]))
    d = xs[0].shape[1]
    if any(x.shape[1] != d for x in xs) or any(h.name != hosts[0].name for h in hosts):
        raise ValueError("fit_many needs sites of one model family and one input dimension")
    # the mean / noise models this loop knows: loadest-gp (learned constant, fixed noise) and rating-gp (power law,
    # fixed + one learned noise term); anything else would silently lose the gradients of its extra parameters
    for m in models:
        learned_noise = getattr(m.likelihood, "second_noise_covar", None) is not None
        if (hosts[0].name == "rating") != learned_noise or (hosts[0].name == "rating") != hasattr(m.model, "powerlaw"):
            raise NotImplementedError("fit_many supports the loadest-gp and rating-gp mean / noise models; "
                                      "train this model with its own fit()")
    sizes = [x.shape[0] for x in xs]
    n = max(sizes)
    plan = GPPlan(hosts[0].name, n, d, dtype=dtype, device=device, lookahead=1 if B > 1 else 2, batch=B)
    if B > 1:
        plan.set_site_sizes(sizes)

    def slots(ts):
        out = torch.zeros((B, n) + tuple(ts[0].shape[1:]), dtype=dtype)
        for b, t in enumerate(ts):
            out[b, : t.shape[0]] = t
        return out.to(device).contiguous()

    X, Y = slots(xs), slots(ys)
    fixed_noise = slots([m.likelihood.noise.reshape(-1).to(dtype) for m in models])
    nvec = torch.tensor(sizes, dtype=torch.float64)
    family = hosts[0].name
    valid = (torch.arange(n)[None, :] < torch.tensor(sizes)[:, None]).to(device)  # (B, n): real observations
    if family == "rating":
        stage = torch.where(valid, X[:, :, 1], torch.full_like(X[:, :, 1], 2.0))  # finite filler in the unused slots
        stage_floor = torch.stack([x[:, 1].min() for x in xs]).to(torch.float64) - 1e-6

    use_penalty = monotonic_penalty_weight > 0.0
    if use_penalty and family != "rating":
        raise NotImplementedError("the monotonicity penalty is rating-gp's (stage is its second column)")
    if use_penalty:
        lo = torch.stack([x.min(dim=0).values for x in xs]).to(torch.float64)  # (B, 2) model-space ranges of every site
        hi = torch.stack([x.max(dim=0).values for x in xs]).to(torch.float64)
        FD, interval = 1e-3, max(1, int(monotonic_penalty_interval))

        def penalty_grid():
            """(B, 2 m, 2): m random points per site -- time uniform over the record, stage log-uniform -- followed by
            the same points with the stage moved up by the forward-difference step (rating_gp/models.py::_MonotonicPenalty)."""
            u = (_penalty_uniforms(grid_size).to(torch.float64) if _penalty_uniforms is not None  # test hook: (B, 2, m)
                 else torch.rand((B, 2, grid_size), dtype=torch.float64, generator=generator))
            t = lo[:, 0:1] + u[:, 0] * (hi[:, 0:1] - lo[:, 0:1])
            llo, lhi = torch.log(lo[:, 1:2] + 1e-6), torch.log(hi[:, 1:2] + 1e-6)
            st = torch.exp(llo + u[:, 1] * (lhi - llo))
            here, there = torch.stack([t, st], dim=2), torch.stack([t, st + FD], dim=2)
            return torch.cat([here, there], dim=1)

    def mean_and_noise(extras):
        """Prior mean and noise diagonal of every site in its slots, built on the device WITHOUT autograd from the
        current values of ``extras``; for rating-gp also the two weight vectors whose reductions give the power
        law's gradients (``dgp_plan_set_dr_weights``)."""
        e = extras.detach().to(device, dtype)
        if family == "rating":
            shifted = stage - e[:, 2:3]
            log_s = torch.log(shifted)
            weights = torch.stack([log_s, torch.reciprocal(shifted)], dim=1).contiguous()  # (B, 2, n)
            plan.set_dr_weights(weights if B > 1 else weights[0].contiguous())
            mean_and_noise.weights = weights
            return e[:, 0:1] + e[:, 1:2] * log_s, fixed_noise + e[:, 3:4]
        return e[:, 0:1].expand(B, n), fixed_noise
    mean_and_noise.weights = None
    plan.set_inputs(X if B > 1 else X[0].contiguous())

    # stacked state of the per-site host modules: every parameter, and the buffers that have one shape across the
    # sites (constraint bounds, prior hyperparameters); site-shaped buffers (the fixed noise) stay the module's own
    own_params = [dict(h.named_parameters()) for h in hosts]  # one tree walk per site, not one per (site, name)
    own_buffers = [dict(h.named_buffers()) for h in hosts]
    params = {k: torch.stack([own[k].detach() for own in own_params]).clone().requires_grad_(True) for k in own_params[0]}
    buffers = {}
    for k, b0 in own_buffers[0].items():
        bs = [own[k] for own in own_buffers]
        if all(b.shape == b0.shape for b in bs):
            buffers[k] = torch.stack(bs)
    host0 = hosts[0]

    def one_site(p, b):
        # tie_weights=False: the rating gate module sits at two places in the tree; swapping each registered name once
        # (what named_parameters lists) is what restores cleanly
        return functional_call(host0, (p, b), (), tie_weights=False)

    host_all = vmap(one_site)

    # ---- per-site optimiser state (torch.optim.Adam + ReduceLROnPlateau semantics, vectorised over the sites)
    lr = torch.full((B,), float(learning_rate), dtype=torch.float64)
    step = torch.zeros(B, dtype=torch.float64)
    m1 = {k: torch.zeros_like(v) for k, v in params.items()}
    m2 = {k: torch.zeros_like(v) for k, v in params.items()}
    beta1, beta2, eps, wd = 0.9, 0.999, 1e-8, (1e-4 if optimizer == "adam" else 1e-2)
    decoupled = optimizer == "adamw"
    best = torch.full((B,), float("inf"), dtype=torch.float64)
    num_bad = torch.zeros(B, dtype=torch.float64)
    cooldown = torch.zeros(B, dtype=torch.float64)
    sched_patience, factor, threshold, cool, min_lr = max(20, patience // 2), 0.7, 1e-4, 10, 1e-6
    nan_run = torch.zeros(B, dtype=torch.float64)
    last_obj = torch.full((B,), float("nan"), dtype=torch.float64)
    it0 = 0
    # early stopping per site (engines/gpytorch.py:407-428): a site that has not improved its best objective by 1e-6
    # for `patience` of its own steps is frozen -- it keeps its slot in the batched step but is no longer updated
    live = torch.ones(B, dtype=torch.bool)
    es_best = torch.full((B,), float("inf"), dtype=torch.float64)
    stale = torch.zeros(B, dtype=torch.float64)
    last_iteration = torch.full((B,), iterations - 1, dtype=torch.int64)
    if resume is not None:
        if set(resume.params) != set(params) or any(resume.params[k].shape != params[k].shape for k in params):
            raise ValueError("resume: the state belongs to a different set of sites / models")
        with torch.no_grad():
            for k in params:
                params[k].copy_(resume.params[k])
        m1 = {k: v.clone() for k, v in resume.m1.items()}
        m2 = {k: v.clone() for k, v in resume.m2.items()}
        step, lr, best, num_bad, cooldown = (getattr(resume, k).clone() for k in ("step", "lr", "best", "num_bad", "cooldown"))
        es_best, stale, live, last_obj = (getattr(resume, k).clone() for k in ("es_best", "stale", "live", "last_obj"))
        nan_run = resume.nan_run.clone()  # consecutive non-finite objectives: a site 2 bad steps from aborting stays there
        it0 = int(resume.iterations_done)
        last_iteration = torch.where(live, torch.full_like(resume.last_iteration, it0 + iterations - 1), resume.last_iteration)

    def per_site(t, v):  # broadcast a (B,) vector against a stacked parameter
        return v.reshape((B,) + (1,) * (t.dim() - 1))

    # ---- closed-form host algebra: every raw parameter of every site in ONE (B, T) tensor; the optimiser below is written over
    # a dict of stacked tensors and does not care that the dict then has a single entry
    cf = None
    if closed_form and not use_penalty and not (penalty_callback is not None and penalty_weight > 0.0):
        cf = _ClosedForm.build(hosts, own_params, family)
    if cf is not None:
        try:
            if family == "rating":
                col_b, col_c = cf.column("model.powerlaw.b"), cf.column("model.powerlaw.c")
            params = {"flat": cf.flatten(params)}
            m1, m2 = {"flat": cf.flatten(m1)}, {"flat": cf.flatten(m2)}
            nvec_np = nvec.numpy()
        except (_Unsupported, KeyError, ValueError):
            cf = None

    # The host share of an iteration is a few dozen operations on (B, T)-sized arrays.  With torch's default intra-op pool (one
    # thread per core: 128 on the GPU boxes) the few of them that cross a parallel threshold wake the whole pool, whose workers
    # then spin while every following small operation waits for a core: measured 19.9 ms per iteration for 256 sites of n = 300
    # in the closed-form path against 0.66 ms with ONE thread (8 threads: 0.94; the autograd path: 1.94 -> 1.44).  One thread
    # for the loop; the caller's setting is restored behind it.
    host_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    t_loop = time.perf_counter()
    it = it0 - 1  # (iterations = 0: nothing runs)
    try:
        for it in range(it0, it0 + iterations):
            if cf is not None:
                flat = params["flat"]
                if family == "rating":  # the reference's in-forward clamps (rating_gp/models/gpytorch.py:39, 259)
                    flat[:, col_b].clamp_(1.2, 2.5)
                    flat[:, col_c] = torch.minimum(flat[:, col_c], stage_floor)
                x, slope = cf.transform(flat.numpy())
                theta, extras = torch.from_numpy(x[:, cf.theta_src]), torch.from_numpy(x[:, cf.extras_src])
                mean, noise = mean_and_noise(extras)
                r = (Y - mean).contiguous()
                noise = noise.contiguous()
                if plan.batch == 1:
                    out = plan.fit_step(theta[0], r[0].contiguous(), noise[0].contiguous())[0].reshape(1, -1)
                else:
                    out = plan.fit_step(theta, r, noise)[0]
                lp_np, dlp = cf.log_prior(x)  # the host's share runs under the device step
                _ok_row, nll, dtheta, dextras = _row_terms(out.to("cpu", torch.float64), plan.ntheta, family,
                                                          extras[:, 1] if family == "rating" else None)
                obj = (nll - torch.from_numpy(lp_np)) / nvec
            else:
                for v in params.values():
                    v.grad = None
                if family == "rating":  # the reference's in-forward clamps (rating_gp/models/gpytorch.py:39, 259)
                    with torch.no_grad():
                        params["model.powerlaw.b"].clamp_(1.2, 2.5)
                        pc = params["model.powerlaw.c"]
                        pc.copy_(torch.minimum(pc, stage_floor.reshape(pc.shape)))
                theta, lp, extras = host_all(params, buffers)
                mean, noise = mean_and_noise(extras)
                r = (Y - mean).contiguous()
                noise = noise.contiguous()
                nll = _BatchedNLL.apply(plan, theta, r, noise, extras, family)
                obj = (nll - lp) / nvec
            if use_penalty and (it + 1) % interval == 0:
                # posterior mean of every site on its grid = GP part (device, batched, differentiable through
                # dgp_mean_vjp) + power-law prior mean at the grid's stages (host autograd)
                grid = penalty_grid()
                gp_part = _BatchedGPMean.apply(plan, theta, grid.to(device, dtype).contiguous(), extras, family,
                                               mean_and_noise.weights, valid.to(dtype))
                prior = extras[:, 0:1] + extras[:, 1:2] * torch.log(grid[:, :, 1] - extras[:, 2:3])
                mu = gp_part + prior
                slope = (mu[:, grid_size:] - mu[:, :grid_size]) / FD
                obj = obj + float(monotonic_penalty_weight) * float(interval) * torch.relu(-slope).mean(dim=1)
            if penalty_callback is not None and penalty_weight > 0.0:
                pens = []
                for b in range(B):
                    try:
                        val = penalty_callback(b, {k: v[b] for k, v in params.items()})
                        if not torch.is_tensor(val):
                            val = None
                    except Exception:  # noqa: BLE001 -- the reference swallows callback failures (engines/gpytorch.py:366-370)
                        val = None
                    pens.append(torch.zeros((), dtype=torch.float64) if val is None else val.reshape(()).to(torch.float64))
                obj = obj + float(penalty_weight) * torch.stack(pens)
            finite = torch.isfinite(obj.detach())
            ok = finite & live
            nan_run = torch.where(finite | ~live, torch.zeros_like(nan_run), nan_run + 1)
            if bool((nan_run > 10).any()):
                raise RuntimeError(f"site {int(torch.argmax(nan_run))}: more than 10 consecutive NaN/Inf objectives "
                                   f"at iteration {it + 1}")
            if cf is not None:
                raw_grads = {"flat": torch.from_numpy(cf.raw_gradient(dlp, slope, dtheta.numpy(), dextras.numpy(), nvec_np, ok.numpy()))}
            else:
                torch.where(ok, obj, torch.zeros_like(obj)).sum().backward()
                raw_grads = {k: v.grad for k, v in params.items()}
            coef, grads = _per_site_clip(raw_grads, B)
            okf = ok.to(torch.float64)
            step = step + okf
            bc1 = 1.0 - beta1 ** step
            bc2 = 1.0 - beta2 ** step
            with torch.no_grad():
                for k, p in params.items():
                    mk = per_site(p, okf)
                    if decoupled:  # torch.optim.AdamW: p *= 1 - lr wd first, the moments see the bare gradient
                        g = grads[k] * per_site(p, coef)
                        p.mul_(torch.where(mk > 0, 1.0 - per_site(p, lr) * wd, torch.ones_like(mk)))
                    else:
                        g = grads[k] * per_site(p, coef) + wd * p
                    m1[k] = torch.where(mk > 0, beta1 * m1[k] + (1 - beta1) * g, m1[k])
                    m2[k] = torch.where(mk > 0, beta2 * m2[k] + (1 - beta2) * g * g, m2[k])
                    denom = torch.sqrt(m2[k]) / per_site(p, torch.sqrt(torch.clamp(bc2, min=1e-300))) + eps
                    upd = per_site(p, lr / torch.clamp(bc1, min=1e-300)) * m1[k] / denom
                    p -= torch.where(mk > 0, upd, torch.zeros_like(upd))
            last_obj = torch.where(ok, obj.detach(), last_obj)
            if scheduler:  # ReduceLROnPlateau.step(obj) for the sites that stepped
                cur = obj.detach()
                better = ok & (cur < best * (1.0 - threshold))
                best = torch.where(better, cur, best)
                num_bad = torch.where(ok, torch.where(better, torch.zeros_like(num_bad), num_bad + 1), num_bad)
                in_cool = ok & (cooldown > 0)
                cooldown = torch.where(in_cool, cooldown - 1, cooldown)
                num_bad = torch.where(in_cool, torch.zeros_like(num_bad), num_bad)
                reduce = ok & (num_bad > sched_patience)
                new_lr = torch.clamp(lr * factor, min=min_lr)
                lr = torch.where(reduce & (lr - new_lr > 1e-8), new_lr, lr)
                cooldown = torch.where(reduce, torch.full_like(cooldown, float(cool)), cooldown)
                num_bad = torch.where(reduce, torch.zeros_like(num_bad), num_bad)
            cur = obj.detach()
            improved = ok & (cur < es_best - 1e-6)
            es_best = torch.where(improved, cur, es_best)
            stale = torch.where(ok, torch.where(improved, torch.zeros_like(stale), stale + 1), stale)
            if early_stopping:
                stop = live & (stale >= patience)
                last_iteration = torch.where(stop, torch.full_like(last_iteration, it), last_iteration)
                live = live & ~stop
            if progress and (it + 1) % 10 == 0:
                print(f"iteration {it + 1}: mean objective {float(last_obj.nanmean()):.4f}, {int(live.sum())} sites training",
                      flush=True)
            if not bool(live.any()):
                break
    finally:
        torch.set_num_threads(host_threads)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    # diagnostics of the last call (bench.py reads them): seconds from entry to the first iteration, seconds in the loop
    LAST_TIMING.update(setup_s=t_loop - t_enter, loop_s=time.perf_counter() - t_loop, iterations=it - it0 + 1, sites=B,
                       closed_form=cf is not None)

    if cf is not None:  # back to the named, stacked form (hand-back, FitManyState)
        params, m1, m2 = cf.unflatten(params["flat"]), cf.unflatten(m1["flat"]), cf.unflatten(m2["flat"])
    # ---- hand the fitted parameters back to the per-site models
    for b, (m, own) in enumerate(zip(models, own_params)):
        _hand_back(m, own, {k: v[b] for k, v in params.items()}, int(last_iteration[b]), xs[b], ys[b])
    if return_state:
        state = FitManyState(params={k: v.detach().clone() for k, v in params.items()}, m1=m1, m2=m2, step=step, lr=lr, best=best,
                             num_bad=num_bad, cooldown=cooldown, es_best=es_best, stale=stale, live=live, last_obj=last_obj,
                             last_iteration=last_iteration, iterations_done=torch.tensor(it0 + iterations), nan_run=nan_run)
        return last_obj, state
    return last_obj


def _set_training(module, flag: bool):
    """``module.train(flag)`` for a whole tree without ``nn.Module.__setattr__``'s type checks per submodule: ``training`` is a
    plain bool in every module's ``__dict__``, and with hundreds of sites of ~60 modules each the recursive ``train()`` /
    ``eval()`` calls were a third of ``fit_many``'s set-up (0.15 s + 0.12 s for 256 sites).  Freshly built trees are already
    in training mode: then nothing is touched."""
    if module.training == flag and all(mod.training == flag for mod in module.children()):
        return
    for mod in module.modules():
        mod.__dict__["training"] = flag


def _hand_back(m, own, values, last_iteration, tx, ty):
    """Fitted raw parameters ``values`` ({name: tensor}) -> the model object ``m`` (``own`` = its ``_HostSide``'s
    ``named_parameters()``); afterwards ``m`` is what ``m.fit`` leaves behind: ready for ``predict``."""
    with torch.no_grad():
        for k, v in values.items():
            own[k].copy_(v.reshape(own[k].shape))
    m._current_iteration = int(last_iteration)
    m._pending_device = (tx, ty)  # the site's own plan is created when it first predicts
    m._plan, m._factor_key = None, None
    _set_training(m.model, False)
    _set_training(m.likelihood, False)
    m.is_fitted = True


# stop reasons in the gathered table of fit_many_distributed
STOP_BUDGET, STOP_EARLY, STOP_FAILED = 0.0, 1.0, 2.0


def fit_many_distributed(models, datasets, group=None, load: str = "all", seed: int | None = 0, **kw):
    """``fit_many`` over the ranks of a ``torch.distributed`` process group (one rank per GPU).

    Every rank passes the SAME ``models`` / ``datasets`` lists (like the reference's ``iterdata`` list,
    ``examples/nwqn-loadest-example/nwqn-loadest-example.py:149-159``).  Site i belongs to rank i mod world
    (``sites.site_partition``); each rank trains its share with ``fit_many(share, **kw)`` on its own device; then ONE
    collective -- an ``all_gather`` of a zero-padded float64 table with one row per site,

        [ raw parameters flattened in ``named_parameters()`` order (P_raw) | final objective | iterations run | stop reason ]

    (stop reason: 0 iteration budget used up, 1 early stopping, 2 the rank's ``fit_many`` raised) -- gives every rank the
    result of every site.  ``load``: "all" (default) -- every rank loads every site's parameters into its model objects, so
    ``predict`` / ``predict_many`` work anywhere; "rank0" -- only rank 0 loads the sites it does not own; "own" -- nobody
    does (the table is still returned everywhere).  There is no per-iteration traffic and no data-path collective.

    A rank whose ``fit_many`` raises (e.g. more than 10 consecutive NaN objectives at one of its sites,
    ``engines/gpytorch.py:356-357``) still takes part in the collective -- its rows carry stop reason 2 and NaN -- and the
    error is then raised on EVERY rank, so no rank is left waiting in the gather.

    ``seed`` (default 0; None: unseeded): site i's model is built under ``torch.manual_seed(seed + i)`` (``fit_many``'s
    ``site_seeds``), so the result of a site does not depend on the number of ranks or on which rank trained it -- a
    single-process ``fit_many(..., site_seeds=[seed + i ...])`` gives the same bits.

    Without an initialised process group (or world 1) this is ``fit_many`` plus the table.  ``resume`` / ``return_state``
    are per-rank notions and not supported here.  Returns ``(objectives (n_sites,), table (n_sites, P_raw + 3))``, float64,
    identical on every rank."""
    import torch.distributed as dist

    from .sites import gather_site_results, site_partition

    if load not in ("all", "rank0", "own"):
        raise ValueError("load must be 'all', 'rank0' or 'own'")
    if "resume" in kw or kw.get("return_state") or "site_seeds" in kw:
        raise ValueError("fit_many_distributed: resume / return_state are per-rank; use fit_many on each rank's share")
    if len(models) != len(datasets) or not models:
        raise ValueError("fit_many_distributed needs one (covariates, target) pair per model")
    n_sites = len(models)
    active = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if active else 1
    rank = dist.get_rank(group) if active else 0
    mine = site_partition(n_sites, world, rank)

    error = None
    objs = state = None
    if mine:
        try:
            objs, state = fit_many([models[i] for i in mine], [datasets[i] for i in mine], return_state=True,
                                   site_seeds=None if seed is None else [seed + i for i in mine], **kw)
        except Exception as e:  # noqa: BLE001 -- reported after the collective, on every rank
            error = e

    # the row layout is a property of the model family: take it from a site this rank has walked, else walk site 0
    def tree(m, record):
        cov, tgt, unc = (tuple(record) + (None,))[:3]
        tx, ty, tu = m._attach(cov, tgt, unc)
        _fresh_seeded(m, tx, ty, tu, 0)  # values are overwritten below; seeded so that the global RNG is not consumed
        return _HostSide(m.model, m.likelihood, tx.shape[1]), tx, ty

    if state is not None:
        names = list(state.params)
        widths = [int(state.params[k][0].numel()) for k in names]
    else:
        host, _, _ = tree(models[0], datasets[0])  # (a rank without sites, or one whose fit raised)
        shapes = dict(host.named_parameters())
        names = list(shapes)
        widths = [int(shapes[k].numel()) for k in names]
    P = sum(widths)
    local = torch.full((len(mine), P + 3), float("nan"), dtype=torch.float64)
    if state is not None:
        flat = torch.cat([state.params[k].detach().reshape(len(mine), -1).to(torch.float64) for k in names], dim=1)
        local[:, :P] = flat
        local[:, P] = objs
        local[:, P + 1] = (state.last_iteration + 1).to(torch.float64)
        local[:, P + 2] = torch.where(state.live, torch.full((len(mine),), STOP_BUDGET, dtype=torch.float64),
                                      torch.full((len(mine),), STOP_EARLY, dtype=torch.float64))
    else:
        local[:, P + 2] = STOP_FAILED
    if active and world > 1:
        on_device = dist.get_backend(group) == "nccl"  # RCCL moves device memory; gloo host memory
        dev = torch.device(models[0].device) if on_device else torch.device("cpu")
        table = gather_site_results(local.to(dev), n_sites, group).cpu()
    else:
        table = local
    failed = torch.nonzero(table[:, P + 2] == STOP_FAILED).reshape(-1).tolist()
    if failed:
        who = sorted({i % world for i in failed})
        raise RuntimeError(f"fit_many_distributed: fit_many failed on rank(s) {who} (sites {failed[:8]}"
                           f"{' ...' if len(failed) > 8 else ''})" + (f": {error}" if error is not None else "")) from error

    # ---- every site's result into the model objects of the ranks that want it
    if load == "all" or (load == "rank0" and rank == 0):
        owned = set(mine)
        for i in range(n_sites):
            if i in owned:
                continue  # fit_many has already handed these back, bitwise the gathered values
            host, tx, ty = tree(models[i], datasets[i])
            own = dict(host.named_parameters())
            if list(own) != names:
                raise ValueError(f"site {i}: parameter tree differs from the gathered table's (one model family per call)")
            values, o = {}, 0
            for k, w in zip(names, widths):
                values[k] = table[i, o:o + w].to(own[k].dtype)
                o += w
            _hand_back(models[i], own, values, int(table[i, P + 1]) - 1, tx, ty)
    return table[:, P].clone(), table


def predict_many(models, covariates_list):
    """``models[i].predict(covariates_list[i])`` for many fitted sites through ONE batched device plan: one batched
    factorisation (``dgp_factorize`` with gridDim.z = sites) and one ``dgp_predict`` over all sites, instead of a plan, a
    factorisation and a prediction per site (the reference fans predictions out per site / per date:
    ``examples/nwqn-loadest-example/nwqn-loadest-example.py:38-125``, ``src/rating_gp/plot.py:259-285``).
    Returns ``[(target, se), ...]`` in the original data space, like ``MarginalHIP.predict``.  Sites may differ in the
    number of observations and of prediction points (ragged batch; shorter sites are padded)."""
    if len(models) != len(covariates_list) or not models:
        raise ValueError("predict_many needs one covariates object per model")
    for m in models:
        if not m.is_fitted:
            raise RuntimeError("The model hasn't been fitted yet, call .fit().")
    B = len(models)
    dtype, device = models[0].dtype, torch.device(models[0].device)
    xs, ys, xnew, thetas, means, noises, names = [], [], [], [], [], [], set()
    with torch.no_grad():
        for m, cov in zip(models, covariates_list):
            m.model.eval()
            m.likelihood.eval()
            x = torch.tensor(m.dm.X, dtype=dtype)
            y = torch.tensor(m.dm.y, dtype=dtype)
            xn = torch.tensor(m.dm.Xnew(cov), dtype=dtype)
            if hasattr(m.model, "prepare_eval"):
                m.model.prepare_eval(x, xn)  # data-dependent clamps see [X; X*]
            name, theta_fn = lower(m.model.covar_module, x.shape[1])
            names.add((name, x.shape[1]))
            xs.append(x)
            ys.append(y)
            xnew.append(xn)
            thetas.append(theta_fn().detach().to(torch.float64))
            means.append(m.model.prior_mean(x).detach().to(dtype))
            noises.append(m.likelihood.train_noise(torch.device("cpu"), dtype).detach().reshape(-1))
    if len(names) != 1:
        raise ValueError("predict_many needs sites of one model family and one input dimension")
    (name, d), = names
    sizes, msizes = [x.shape[0] for x in xs], [x.shape[0] for x in xnew]
    n, mmax = max(sizes), max(msizes)
    plan = GPPlan(name, n, d, dtype=dtype, device=device, lookahead=1 if B > 1 else 2, batch=B)
    if B > 1:
        plan.set_site_sizes(sizes)

    def slots(ts, width, fill=0.0):
        out = torch.full((B, width) + tuple(ts[0].shape[1:]), fill, dtype=dtype)
        for b, t in enumerate(ts):
            out[b, : t.shape[0]] = t
        return out

    X = slots(xs, n).to(device).contiguous()
    R = slots([y - mu for y, mu in zip(ys, means)], n).to(device).contiguous()
    Nz = slots(noises, n, 1.0).to(device).contiguous()
    Xs = torch.stack([torch.cat([xn, xn[-1:].expand(mmax - xn.shape[0], -1)]) for xn in xnew]).to(device).contiguous()
    theta = torch.stack(thetas)
    single = B == 1
    plan.set_inputs(X[0].contiguous() if single else X)
    out = plan.factorize(theta[0] if single else theta, R[0].contiguous() if single else R, Nz[0].contiguous() if single else Nz)
    info = out.reshape(B, -1)[:, _lib.OUT_INFO].cpu()
    if bool((info != 0).any()):
        bad = int(torch.nonzero(info)[0])
        raise RuntimeError(f"site {bad}: matrix not positive definite (Cholesky pivot {int(info[bad])})")
    kmean, kvar = plan.predict(theta[0] if single else theta, Xs[0].contiguous() if single else Xs)
    kmean, kvar = kmean.reshape(B, mmax).cpu(), kvar.reshape(B, mmax).cpu()
    results = []
    with torch.no_grad():
        for b, (m, cov) in enumerate(zip(models, covariates_list)):
            mb = msizes[b]
            mu = kmean[b, :mb] + m.model.prior_mean(xnew[b]).to(dtype)
            var = kvar[b, :mb] + m.likelihood.predictive_noise(mb, torch.device("cpu"), dtype)
            target = m.dm.y_t(mu.numpy()).assign_coords(cov.coords)
            se = m.dm.error_pipeline.inverse_transform(var.numpy()).assign_coords(cov.coords)
            results.append((target, se))
    return results


__all__ = ["fit_many", "fit_many_distributed", "predict_many", "FitManyState"]
