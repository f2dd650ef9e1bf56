"""Host logic of the engine on CPU: lowering, constraints, priors, objective assembly, fit loop,
prediction wrappers and checkpointing -- with the device plan replaced by an oracle-backed double."""
import io

import numpy as np
import pytest
import torch

from discontinuum_amd.engines.hip import MarginalHIP
from discontinuum_amd.gp import kernels as K
from discontinuum_amd.gp.lowering import UnsupportedKernelError, lower
from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.rating_gp import RatingGP
from oracle import gp_oracle as orc
from tests.helpers import OraclePlan, loadest_dataset, rating_dataset


@pytest.fixture(autouse=True)
def cpu_engine(monkeypatch):
    monkeypatch.setattr(MarginalHIP, "_plan_factory", staticmethod(OraclePlan))
    monkeypatch.setattr(MarginalHIP, "device", "cpu")
    torch.manual_seed(0)


def _loadest_raw_from_model(m):
    """Flatten the model's raw parameters in oracle order."""
    cm = m.model.covar_module
    s0, s1, s2 = cm.kernels
    per, m52 = s0.base_kernel.kernels
    return torch.cat([v.detach().reshape(-1) for v in (
        m.model.mean_module.raw_constant, s0.raw_outputscale, per.raw_lengthscale, per.raw_period_length,
        m52.raw_lengthscale, s1.raw_outputscale, s1.base_kernel.raw_lengthscale, s2.raw_outputscale,
        s2.base_kernel.raw_lengthscale)])


def test_loadest_objective_and_gradient_match_oracle():
    cov, tgt = loadest_dataset(60)
    m = LoadestGP()
    m.fit(cov, tgt, iterations=1)
    # perturb so that no parameter sits at its initial value
    with torch.no_grad():
        for p in m.model.parameters():
            p.add_(0.3 * torch.randn_like(p))
    m.model.zero_grad(set_to_none=True)  # the training iteration left its gradients behind
    mll = ExactMarginalLogLikelihood(m.likelihood, m.model)
    obj = -mll(m._prior(), m._train_y)
    assert obj.shape == (1,)
    obj.sum().backward()
    raw = _loadest_raw_from_model(m).clone().requires_grad_(True)
    o = orc.LoadestOracle(2)
    ref = o.objective(raw, torch.tensor(m.X), torch.tensor(m.y))
    ref.backward()
    assert abs(obj.item() - ref.item()) < 1e-12 * max(1, abs(ref.item()))
    got = torch.cat([p.grad.reshape(-1) for p in (
        m.model.mean_module.raw_constant, *[q for q in m.model.covar_module.parameters()])])
    assert torch.allclose(got, raw.grad, rtol=1e-9, atol=1e-12)


def test_rating_objective_matches_oracle():
    cov, tgt, unc = rating_dataset(50)
    m = RatingGP()
    m.fit(cov, tgt, target_unc=unc, iterations=1)
    mll = ExactMarginalLogLikelihood(m.likelihood, m.model)
    obj = -mll(m._prior(), m._train_y)
    mod = m.model
    X, y, yu = torch.tensor(m.X), torch.tensor(m.y), torch.tensor(m.y_unc)
    o = orc.RatingOracle.from_stage(X[:, 1])
    name, theta_fn = lower(mod.covar_module, 2)
    assert name == "rating"
    theta = theta_fn().detach()
    raw = torch.zeros(20, dtype=torch.float64)
    raw[0], raw[1], raw[2] = mod.powerlaw.a.item(), mod.powerlaw.b.item(), mod.powerlaw.c.item()
    raw[3] = m.likelihood.second_noise_covar.raw_noise.item()
    raw[4] = orc.inv_interval(theta[0], o.b_lo, o.b_hi)
    raw[5:] = orc.inv_softplus(theta[1:])
    raw.requires_grad_(True)
    ref = o.objective(raw, X, y, yu)
    assert abs(obj.item() - ref.item()) < 1e-10 * max(1, abs(ref.item()))
    # power-law mean and learned-noise gradients (the engine takes them from the result row's reductions)
    for p in m.model.parameters():
        p.grad = None
    obj.backward()
    ref.backward()
    got = torch.stack([mod.powerlaw.a.grad.reshape(()), mod.powerlaw.b.grad.reshape(()), mod.powerlaw.c.grad.reshape(()),
                       m.likelihood.second_noise_covar.raw_noise.grad.reshape(())])
    assert torch.allclose(got, raw.grad[:4], rtol=1e-8, atol=1e-12), (got, raw.grad[:4])


def test_fit_reduces_objective_and_predicts():
    cov, tgt = loadest_dataset(50)
    m = LoadestGP()
    assert not m.is_fitted
    with pytest.raises(RuntimeError, match="hasn't been fitted"):
        m.predict(cov)
    m.fit(cov, tgt, iterations=15)
    assert m.is_fitted and m._current_iteration == 14
    mll = ExactMarginalLogLikelihood(m.likelihood, m.model)
    after = -mll(m._prior(), m._train_y).item()
    fresh = LoadestGP()
    fresh.fit(cov, tgt, iterations=1)
    with torch.no_grad():
        for p in fresh.model.parameters():
            p.zero_()
    before = -ExactMarginalLogLikelihood(fresh.likelihood, fresh.model)(fresh._prior(), fresh._train_y).item()
    assert after < before
    target, se = m.predict(cov)
    assert target.values.shape == (50,) and se.values.shape == (50,)
    assert np.all(np.isfinite(target.values)) and np.all(se.values >= 1.0)  # GSE >= 1
    # model-space parity of the prediction wrapper with the oracle
    raw = _loadest_raw_from_model(m)
    mu_ref, var_ref = orc.LoadestOracle(2).predict(raw, torch.tensor(m.X), torch.tensor(m.y), torch.tensor(m.X))
    mu, var = m._model_space_predict(torch.tensor(m.X))
    assert torch.allclose(mu, mu_ref, atol=1e-9) and torch.allclose(var, var_ref, atol=1e-9)
    grid = m.predict_grid("flow")
    assert grid.values.shape[1] == 18
    draws = m.sample(cov, n=7)
    assert draws.values.shape == (7, 50)


def test_unsupported_optimizer_and_resume():
    cov, tgt = loadest_dataset(30)
    m = LoadestGP()
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        m.fit(cov, tgt, iterations=2, optimizer="sgd")
    m.fit(cov, tgt, iterations=4, optimizer="adamw")
    calls = m._plan.calls
    m.fit(cov, tgt, iterations=4, resume=True)  # nothing left to do
    m.fit(cov, tgt, iterations=6, resume=True)
    assert m._current_iteration == 5 and m._plan.calls > calls


def test_save_load_roundtrip():
    cov, tgt = loadest_dataset(30)
    m = LoadestGP()
    with pytest.raises(RuntimeError, match="No model to save"):
        m.save(io.BytesIO())
    m.fit(cov, tgt, iterations=3)
    buf = io.BytesIO()
    m.save(buf, extra={"site": "x"})
    buf.seek(0)
    m2 = LoadestGP.load(buf, cov, tgt)
    assert m2.is_fitted and m2._current_iteration == 2
    for a, b in zip(m.model.parameters(), m2.model.parameters()):
        assert torch.equal(a, b)
    t1, _ = m.predict(cov)
    t2, _ = m2.predict(cov)
    assert np.allclose(t1.values, t2.values)
    m2.fit(cov, tgt, iterations=5, resume=True)  # continues from the checkpointed iteration


def test_checkpoints_are_plain_data_and_loading_executes_nothing():
    """``save`` writes plain data (``torch.load(weights_only=True)`` reads it), the transform of the saved configuration
    comes back, an earlier-format file that pickled the ``ModelConfig`` dataclass still loads (allow-listed class), a
    file that needs any other global is refused instead of unpickled, and a state dict that lacks a parameter is an
    error while differing prior / constraint buffers are not."""
    from discontinuum_amd.engines.base import ModelConfig

    cov, tgt = loadest_dataset(30)
    m = LoadestGP(ModelConfig(transform="standard"))
    m.fit(cov, tgt, iterations=2)
    buf = io.BytesIO()
    m.save(buf)
    buf.seek(0)
    record = torch.load(buf, map_location="cpu", weights_only=True)
    assert record["model_config"] == {"transform": "standard"}
    buf.seek(0)
    assert LoadestGP.load(buf, cov, tgt).model_config.transform == "standard"
    # earlier format: the dataclass itself in the record
    old = dict(record, model_config=ModelConfig(transform="standard"))
    buf = io.BytesIO()
    torch.save(old, buf)
    buf.seek(0)
    assert LoadestGP.load(buf, cov, tgt).model_config.transform == "standard"
    # anything else that would need unpickling is refused
    evil = dict(record, extra={"payload": io.BytesIO})
    buf = io.BytesIO()
    torch.save(evil, buf)
    buf.seek(0)
    with pytest.raises(Exception, match="(?i)weights_only|unsupported|allowlist|global"):
        LoadestGP.load(buf, cov, tgt)
    # buffers of priors / constraints may differ; parameters may not
    lean = dict(record, model_state_dict={k: v for k, v in record["model_state_dict"].items() if "prior" not in k})
    assert len(lean["model_state_dict"]) < len(record["model_state_dict"])
    buf = io.BytesIO()
    torch.save(lean, buf)
    buf.seek(0)
    m3 = LoadestGP.load(buf, cov, tgt)
    for a, b in zip(m.model.parameters(), m3.model.parameters()):
        assert torch.equal(a, b)
    broken = dict(record, model_state_dict={k: v for k, v in record["model_state_dict"].items() if "raw_period" not in k})
    buf = io.BytesIO()
    torch.save(broken, buf)
    buf.seek(0)
    with pytest.raises(RuntimeError, match="missing parameters"):
        LoadestGP.load(buf, cov, tgt)


def test_rating_fit_and_predict():
    cov, tgt, unc = rating_dataset(40)
    m = RatingGP()
    m.fit(cov, tgt, target_unc=unc, iterations=5)
    assert m.is_fitted
    target, se = m.predict(cov)
    assert np.all(np.isfinite(target.values))
    b = m.model.powerlaw.b.item()
    assert 1.2 <= b <= 2.5
    # monotonicity penalty (reference: tests/test_rating_gp.py:52-65): differentiable predictive mean
    m2 = RatingGP()
    m2.fit(cov, tgt, target_unc=unc, iterations=5, monotonic_penalty_weight=0.5, grid_size=16, scheduler=False)
    assert m2.is_fitted
    # the penalty's gradient reaches kernel, mean and noise parameters
    m2.model.zero_grad(set_to_none=True)
    x = torch.tensor(m2.X[:6]).clone()
    mu = m2._differentiable_mean(x)
    mu.sum().backward()
    grads = [p.grad for p in m2.model.parameters()]
    assert all(g is not None for g in grads) and any(float(g.abs().sum()) > 0 for g in grads)


def test_lowering_of_other_structures():
    """Trees other than the two fused models: sums of (scaled) products of RBF / Matern / Periodic factors lower onto
    the generic composite evaluator (the description and the parameter order are checked against the fused loadest
    model through the oracle); gates, warps and nested sums are rejected."""
    from discontinuum_amd.gp.lowering import composite_spec
    from discontinuum_amd.loadest_gp.models import loadest_covariance

    k = K.ScaleKernel(K.RBFKernel(active_dims=[0])) + K.ScaleKernel(K.RBFKernel(active_dims=[1]))
    name, theta = lower(k, 2)
    assert name.startswith("composite:") and theta().numel() == 4
    assert lower(K.ScaleKernel(K.RBFKernel()), 2)[0].startswith("composite:")
    assert lower(k, 2)[0] == name  # an identical description is registered once
    # the generic description of the loadest covariance reproduces the fused model's Gram with the same theta order
    cov = loadest_covariance(3)
    spec, parts = composite_spec(cov, 3)
    th = torch.cat([getattr(m, a).reshape(-1) for m, a in parts]).detach() * torch.linspace(0.7, 1.4, 11, dtype=torch.float64)
    X = torch.randn(30, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    assert (orc.composite_gram(spec)(X, X, th) - orc.loadest_gram(X, X, th)).abs().max() < 1e-14
    assert lower(cov, 3)[0] == "loadest"  # the fused evaluator still takes what it recognises
    with pytest.raises(UnsupportedKernelError):
        lower(K.LogWarpKernel(K.ScaleKernel(K.RBFKernel(active_dims=[1])), dim=1), 2)
    with pytest.raises(UnsupportedKernelError):  # a sum inside a product
        lower(K.ScaleKernel(K.RBFKernel(active_dims=[0]) * (K.RBFKernel(active_dims=[1]) + K.MaternKernel(active_dims=[1]))), 2)
    with pytest.raises(UnsupportedKernelError):  # more columns than the evaluator carries
        lower(K.ScaleKernel(K.RBFKernel(active_dims=list(range(7)))), 7)


def test_nan_objective_guard():
    """More than 10 consecutive failing objective evaluations re-raise (engines/gpytorch.py:352-358)."""
    cov, tgt = loadest_dataset(20)
    m = LoadestGP()
    m.fit(cov, tgt, iterations=1)

    def boom(*a, **k):
        raise RuntimeError("not psd")

    m._plan.fit_step = boom
    with pytest.raises(RuntimeError, match="not psd"):
        m.fit(cov, tgt, iterations=30, resume=True)
    assert m.is_fitted


def test_gradient_clipping_matches_torch():
    """The loop's own clip (one flat reduction) against ``torch.nn.utils.clip_grad_norm_``, broken gradients included."""
    from discontinuum_amd.engines.hip import _clip_grad_norm

    for case in range(8):
        torch.manual_seed(case)
        ours = [torch.nn.Parameter(torch.randn(s, dtype=torch.float64)) for s in [(), (1,), (1, 3), (2,), (1, 1)]]
        ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
        for p, q in zip(ours, ref):
            g = torch.randn_like(p) * (10 if case % 2 else 0.01)
            if case == 6:
                g.reshape(-1)[0] = float("nan")
            if case == 7:
                g.reshape(-1)[0] = float("inf")
            p.grad, q.grad = g.clone(), g.clone()
        a = _clip_grad_norm(ours, 1.0)
        b = float(torch.nn.utils.clip_grad_norm_(ref, 1.0))
        assert (a == b) or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-15 * abs(b)
        for p, q in zip(ours, ref):
            assert torch.allclose(p.grad, q.grad, rtol=1e-15, atol=0, equal_nan=True)


def test_lean_optimizer_steps_are_torchs_bit_for_bit():
    """The loop's Adam / AdamW (torch's multi-tensor arithmetic without ``Optimizer.step``'s bookkeeping) against
    ``torch.optim``: parameters and optimiser state identical after 60 steps with a learning-rate change, a parameter
    without gradient falls back to torch's own step."""
    from discontinuum_amd.engines.hip import _LeanAdam, _LeanAdamW

    for lean, ref, wd in ((_LeanAdam, torch.optim.Adam, 1e-4), (_LeanAdamW, torch.optim.AdamW, 1e-2)):
        torch.manual_seed(0)
        ours = [torch.nn.Parameter(torch.randn(s, dtype=torch.float64)) for s in [(), (1,), (1, 3), (2,), (1, 1)]]
        theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
        a = lean(ours, lr=0.05, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=True)
        b = ref(theirs, lr=0.05, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=True)
        for it in range(60):
            for p, q in zip(ours, theirs):
                g = torch.randn_like(p)
                p.grad, q.grad = g.clone(), g.clone()
            if it == 30:
                a.param_groups[0]["lr"] = b.param_groups[0]["lr"] = 0.035
            if it == 45:  # a parameter without gradient: torch skips it, and so must we
                ours[2].grad = theirs[2].grad = None
            if it == 20:  # a checkpoint round trip: the flat buffers are rebuilt from what load_state_dict put there
                import copy
                import io

                blobs = []
                for opt in (a, b):
                    buf = io.BytesIO()
                    torch.save(opt.state_dict(), buf)
                    buf.seek(0)
                    blobs.append(torch.load(buf, weights_only=False))
                a = lean(ours, lr=0.05, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=True)
                b = ref(theirs, lr=0.05, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=True)
                a.load_state_dict(copy.deepcopy(blobs[1]))  # crosswise: each loads the other's checkpoint
                b.load_state_dict(copy.deepcopy(blobs[0]))
            a.step()
            b.step()
        for p, q in zip(ours, theirs):
            assert torch.equal(p.detach(), q.detach())
        sa, sb = a.state_dict(), b.state_dict()
        assert sa["param_groups"] == sb["param_groups"]
        for k in sb["state"]:
            for name in ("step", "exp_avg", "exp_avg_sq"):
                assert torch.equal(sa["state"][k][name], sb["state"][k][name]), (k, name)


@pytest.mark.parametrize("family", ["loadest", "rating"])
def test_explicit_host_algebra_matches_autograd(family):
    """gp/explicit.py (closed-form constraints, priors and chain rule) against the autograd path: same objective, same
    gradient for every parameter, at the initial point and after training has moved the parameters."""
    from discontinuum_amd.gp.explicit import ExplicitObjective

    if family == "loadest":
        cov, tgt = loadest_dataset(60)
        m = LoadestGP()
        m.fit(cov, tgt, iterations=1)
    else:
        cov, tgt, unc = rating_dataset(50)
        m = RatingGP()
        m.fit(cov, tgt, target_unc=unc, iterations=1)
    m.model.train()
    m.likelihood.train()
    for trial in range(3):
        if trial:  # move every parameter somewhere else
            with torch.no_grad():
                for p in m.model.parameters():
                    p.add_(0.3 * torch.randn(p.shape, dtype=p.dtype))
        mll = ExactMarginalLogLikelihood(m.likelihood, m.model)
        explicit = ExplicitObjective.build(m, mll._priors)
        assert explicit is not None, "the shipped models must take the closed-form path"
        params = list(m.model.parameters())
        for p in params:
            p.grad = None
        value = explicit.evaluate()
        got = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        objective = -mll(m._prior(), m._train_y)
        objective.backward()
        assert abs(value - objective.item()) <= 1e-12 * max(1.0, abs(objective.item()))
        for p, g in zip(params, got):
            assert p.grad is not None and g.shape == p.grad.shape
            assert torch.allclose(g, p.grad, rtol=1e-10, atol=1e-13), (family, trial, g, p.grad)


def test_explicit_host_algebra_declines_what_it_cannot_match():
    """A prior on something that is not one constrained parameter value keeps the autograd path."""
    from discontinuum_amd.gp.explicit import ExplicitObjective
    from discontinuum_amd.gp.priors import NormalPrior

    cov, tgt = loadest_dataset(40)
    m = LoadestGP()
    m.fit(cov, tgt, iterations=1)
    scale = m.model.covar_module.kernels[0]
    scale.register_prior("odd_prior", NormalPrior(0.0, 1.0), lambda mod: mod.outputscale * 2.0)
    mll = ExactMarginalLogLikelihood(m.likelihood, m.model)
    assert ExplicitObjective.build(m, mll._priors) is None
    m.fit(cov, tgt, iterations=3)  # and training still works, through autograd


@pytest.mark.parametrize("family", ["loadest", "rating"])
def test_closed_form_training_follows_the_autograd_trajectory(family, monkeypatch):
    """A whole fit through the closed-form host algebra (flat gradient, flat clip, flat optimiser step) against the
    same fit through autograd: the parameters stay together to rounding."""
    def run(explicit):
        monkeypatch.setattr(MarginalHIP, "explicit_host_algebra", explicit)
        torch.manual_seed(123)
        if family == "loadest":
            cov, tgt = loadest_dataset(45, seed=4)
            m = LoadestGP()
            m.fit(cov, tgt, iterations=25, learning_rate=0.1)
        else:
            cov, tgt, unc = rating_dataset(40, seed=4)
            m = RatingGP()
            m.fit(cov, tgt, target_unc=unc, iterations=25, learning_rate=0.1)
        return torch.cat([p.detach().reshape(-1) for p in m.model.parameters()])

    a, b = run(True), run(False)
    assert torch.allclose(a, b, rtol=1e-8, atol=1e-10), (a - b).abs().max()


def test_fit_many_clip_matches_the_single_site_loop_on_a_broken_gradient():
    """``fit_many``'s per-site clipping must do what the reference loop (engines/gpytorch.py:387-400) and
    ``MarginalHIP.fit`` do with a gradient that holds one NaN / Inf: the whole site's gradient ends up zero (the step is
    weight decay only); healthy sites are clipped by torch's rule."""
    from discontinuum_amd.engines.hip import _clip_grad_norm
    from discontinuum_amd.multisite_fit import _per_site_clip

    torch.manual_seed(0)
    B = 4
    raw = {"a": torch.randn(B, 3, dtype=torch.float64) * 3, "b": torch.randn(B, 1, 2, dtype=torch.float64) * 3}
    raw["a"][1, 2] = float("nan")
    raw["b"][3, 0, 1] = float("inf")
    raw["a"][2] *= 1e-3  # a site below the clip threshold
    raw["b"][2] *= 1e-3
    coef, grads = _per_site_clip({k: v.clone() for k, v in raw.items()}, B)
    for site in range(B):
        ps = [torch.nn.Parameter(torch.zeros_like(raw[k][site])) for k in ("a", "b")]
        for p, k in zip(ps, ("a", "b")):
            p.grad = raw[k][site].clone()
        total = _clip_grad_norm(ps, 1.0)  # the single-site loop: clip, then zero everything if the norm is not finite
        if not np.isfinite(total):
            for p in ps:
                p.grad = torch.nan_to_num(p.grad, nan=0.0, posinf=0.0, neginf=0.0)
        for p, k in zip(ps, ("a", "b")):
            got = grads[k][site] * coef[site]
            assert torch.equal(got, p.grad) or (got - p.grad).abs().max() < 1e-15, (site, k, got, p.grad)
    assert float(coef[1]) == 0.0 and float(coef[3]) == 0.0 and float(coef[2]) == 1.0 and 0 < float(coef[0]) < 1


def test_parameter_names_follow_the_reference_module_trees():
    """The parameter keys of ``state_dict()`` are those gpytorch gives the reference's module trees: attribute names from
    the reference's model definitions (``mean_module``, ``covar_module``, ``powerlaw.a/b/c``:
    src/loadest_gp/models/gpytorch.py:61-128, src/rating_gp/models/gpytorch.py:28-40, 205-256), gpytorch's ``raw_*`` names for
    constrained parameters, ``kernels.<i>`` for the flattened sums / products in the order the reference composes them,
    ``base_kernel`` under ScaleKernel / LogWarpKernel.  (Buffers of priors and constraints are NOT pinned: whether gpytorch
    registers them differs between versions -- ``load`` tolerates them)"""
    cov, tgt = loadest_dataset(20)
    m = LoadestGP()
    m.fit(cov, tgt, iterations=1)
    assert [k for k, _ in m.model.named_parameters()] == [
        "mean_module.raw_constant",
        "covar_module.kernels.0.raw_outputscale",                          # cov_seasonal: Scale(Periodic * Matern52)
        "covar_module.kernels.0.base_kernel.kernels.0.raw_lengthscale",
        "covar_module.kernels.0.base_kernel.kernels.0.raw_period_length",
        "covar_module.kernels.0.base_kernel.kernels.1.raw_lengthscale",
        "covar_module.kernels.1.raw_outputscale",                          # cov_covariates: Scale(RBF)
        "covar_module.kernels.1.base_kernel.raw_lengthscale",
        "covar_module.kernels.2.raw_outputscale",                          # cov_residual: Scale(Matern32)
        "covar_module.kernels.2.base_kernel.raw_lengthscale",
    ]
    assert list(m.likelihood.state_dict()) == [] or all("noise" in k for k in m.likelihood.state_dict())
    cov, tgt, unc = rating_dataset(20)
    r = RatingGP()
    r.fit(cov, tgt, target_unc=unc, iterations=1)
    names = [k for k, _ in r.model.named_parameters()]
    lower = "covar_module.kernels.0.kernels.1.base_kernel"   # sigmoid_lower * LogWarp(cov_shift + cov_shift)
    upper = "covar_module.kernels.1.kernels.1.base_kernel"   # sigmoid_upper * LogWarp(cov_bend)
    rest = "covar_module.kernels.2.base_kernel"              # LogWarp(cov_base + cov_periodic)
    assert names == [
        "likelihood.second_noise_covar.raw_noise", "powerlaw.a", "powerlaw.b", "powerlaw.c",
        "covar_module.kernels.0.kernels.0.raw_b",
        f"{lower}.kernels.0.raw_outputscale", f"{lower}.kernels.0.base_kernel.kernels.0.raw_lengthscale",
        f"{lower}.kernels.0.base_kernel.kernels.1.raw_lengthscale",
        f"{lower}.kernels.1.raw_outputscale", f"{lower}.kernels.1.base_kernel.kernels.0.raw_lengthscale",
        f"{lower}.kernels.1.base_kernel.kernels.1.raw_lengthscale",
        f"{upper}.raw_outputscale", f"{upper}.base_kernel.kernels.0.raw_lengthscale", f"{upper}.base_kernel.kernels.1.raw_lengthscale",
        f"{rest}.kernels.0.raw_outputscale", f"{rest}.kernels.0.base_kernel.raw_lengthscale",
        f"{rest}.kernels.1.raw_outputscale", f"{rest}.kernels.1.base_kernel.kernels.0.raw_lengthscale",
        f"{rest}.kernels.1.base_kernel.kernels.0.raw_period_length", f"{rest}.kernels.1.base_kernel.kernels.1.raw_lengthscale",
    ]
    # the inverted gate shares the switch point: its state_dict entry is the same tensor under a second name
    sd = r.model.state_dict()
    assert "covar_module.kernels.1.kernels.0.sigmoid_kernel.raw_b" in sd
    assert [k for k, _ in r.likelihood.named_parameters()] == ["second_noise_covar.raw_noise"]


def test_closed_form_monotonic_penalty_matches_the_autograd_penalty(monkeypatch):
    """rating-gp's monotonicity penalty inside the closed-form training loop (``_MonotonicPenalty.explicit_terms``: one
    predict_mean + one mean_vjp, power-law chain rule by hand) against the autograd expression of the same penalty
    (``_differentiable_mean``), on identical penalty grids and a rating that bends down (so that the penalty is ACTIVE):
    same trajectory (parameters 1e-9 after 25 iterations), and not the trajectory of a fit without penalty."""
    from discontinuum_amd.engines.hip import MarginalHIP
    from discontinuum_amd.rating_gp import models as rmod

    cov, tgt, unc = rating_dataset(70, seed=3)
    stage = cov["stage"].values
    bent = tgt.values * np.exp(-1.5 * np.maximum(stage - np.median(stage), 0.0) ** 2)
    tgt = type(tgt)(bent, dims=tgt.dims, coords={"time": tgt.coords["time"]}, name=tgt.name, attrs=dict(tgt.attrs))
    iters, m = 25, 12
    table = torch.rand((iters + 2, 2, m), dtype=torch.float64, generator=torch.Generator().manual_seed(5))

    def run(explicit, weight):
        calls = {"k": 0}

        def draw(mm):
            calls["k"] += 1
            return table[calls["k"] - 1]

        monkeypatch.setattr(rmod._MonotonicPenalty, "uniforms", staticmethod(draw))
        monkeypatch.setattr(MarginalHIP, "explicit_host_algebra", explicit)
        torch.manual_seed(7)
        mod = RatingGP()
        mod.fit(cov, tgt, target_unc=unc, iterations=iters, monotonic_penalty_weight=weight, grid_size=m, scheduler=False)
        return torch.cat([p.detach().reshape(-1) for p in mod.model.parameters()]), calls["k"]

    p_auto, k_auto = run(False, 4.0)
    p_closed, k_closed = run(True, 4.0)
    p_plain, _ = run(True, 0.0)
    assert k_auto == iters and k_closed == iters
    assert (p_auto - p_closed).abs().max() < 1e-9, (p_auto - p_closed).abs().max()
    assert (p_closed - p_plain).abs().max() > 1e-4


def test_bench_gpu_state_sampler_degrades_to_none():
    """bench.GpuStateSampler without a working rocm-smi (this container has no GPU) yields None, never an exception."""
    import time

    import bench

    smp = bench.GpuStateSampler()
    t0 = time.time()
    time.sleep(0.3)
    out = smp.window(t0, time.time())
    assert out is None or out["samples"] >= 1


def test_fit_many_state_round_trips_through_weights_only_and_carries_the_nan_counter():
    """``FitManyState`` is plain tensors: ``torch.save(state.as_dict())`` -> ``torch.load(weights_only=True)`` ->
    ``from_dict`` gives every field back, including ``nan_run`` (consecutive non-finite objectives per site -- the counter
    behind the abort of engines/gpytorch.py:356-357, 378-379: a resumed run must not hand a site a fresh budget); a
    dictionary written before the counter travelled loads with zeros."""
    import io

    from discontinuum_amd.multisite_fit import FitManyState

    B = 3
    vec = lambda v: torch.full((B,), float(v), dtype=torch.float64)  # noqa: E731
    st = FitManyState(params={"a.raw": torch.randn(B, 2, dtype=torch.float64)}, m1={"a.raw": torch.zeros(B, 2, dtype=torch.float64)},
                      m2={"a.raw": torch.ones(B, 2, dtype=torch.float64)}, step=vec(7), lr=vec(0.1), best=vec(1.5), num_bad=vec(2),
                      cooldown=vec(0), es_best=vec(1.4), stale=vec(3), live=torch.tensor([True, False, True]), last_obj=vec(1.6),
                      last_iteration=torch.tensor([6, 4, 6]), iterations_done=torch.tensor(7),
                      nan_run=torch.tensor([0.0, 2.0, 9.0], dtype=torch.float64))
    buf = io.BytesIO()
    torch.save(st.as_dict(), buf)
    buf.seek(0)
    back = FitManyState.from_dict(torch.load(buf, weights_only=True))
    assert set(FitManyState.FIELDS) == set(back.as_dict()) and "nan_run" in FitManyState.FIELDS
    for k in FitManyState.FIELDS:
        a, b = getattr(st, k), getattr(back, k)
        if isinstance(a, dict):
            assert all(torch.equal(a[q], b[q]) for q in a)
        else:
            assert torch.equal(a, b), k
    old = {k: v for k, v in st.as_dict().items() if k != "nan_run"}
    assert torch.equal(FitManyState.from_dict(old).nan_run, torch.zeros(B, dtype=torch.float64))


@pytest.mark.parametrize("family", ["loadest", "rating"])
def test_fit_many_closed_form_follows_the_autograd_trajectory(family, monkeypatch):
    """``fit_many(closed_form=True)`` (the default: constraints, hyperparameter vectors, log-priors and the chain rule for all
    sites as a dozen numpy operations on one (B, T) array, ``multisite_fit._ClosedForm``) against ``closed_form=False``
    (``torch.func.vmap`` over the module trees + autograd) on the oracle-backed batched plan double: the same arithmetic
    in a different order, so after 8 Adam iterations the raw parameters agree to 1e-12 and the objectives to 1e-12 relative;
    a run split by ``return_state`` / ``resume`` (5 + 3) is bitwise the uninterrupted closed-form run, and the state it returns
    has the named layout the autograd path's has.  Reference loop: /root/reference/src/discontinuum/engines/gpytorch.py:346-451."""
    from discontinuum_amd import multisite_fit

    monkeypatch.setattr(MarginalHIP, "_plan_factory", staticmethod(OraclePlan))
    monkeypatch.setattr(MarginalHIP, "device", "cpu")
    monkeypatch.setattr(multisite_fit, "GPPlan", OraclePlan)
    sizes = (31, 44, 27, 38) if family == "loadest" else (33, 29, 41)

    def sites():
        if family == "loadest":
            return [LoadestGP() for _ in sizes], [loadest_dataset(n, seed=10 + i) for i, n in enumerate(sizes)]
        return [RatingGP() for _ in sizes], [rating_dataset(n, seed=20 + i) for i, n in enumerate(sizes)]

    def flat(ms):
        return torch.cat([p.detach().reshape(-1) for m in ms for _, p in sorted(m.model.named_parameters())]
                         + [p.detach().reshape(-1) for m in ms for _, p in sorted(m.likelihood.named_parameters())])

    seeds = list(range(len(sizes)))
    used = []
    real_build = multisite_fit._ClosedForm.build
    monkeypatch.setattr(multisite_fit._ClosedForm, "build", classmethod(lambda cls, *a: used.append(real_build(*a)) or used[-1]))
    ma, da = sites()
    oa, sa = multisite_fit.fit_many(ma, da, iterations=8, site_seeds=seeds, return_state=True)
    assert used and used[-1] is not None, "the shipped models must take the closed-form path"
    mb, db = sites()
    ob, sb = multisite_fit.fit_many(mb, db, iterations=8, site_seeds=seeds, closed_form=False, return_state=True)
    assert len(used) == 1  # closed_form=False never builds it
    assert (flat(ma) - flat(mb)).abs().max() <= 1e-12, (flat(ma) - flat(mb)).abs().max()
    assert ((oa - ob).abs() / ob.abs()).max() <= 1e-12
    assert set(sa.params) == set(sb.params) and all(sa.params[k].shape == sb.params[k].shape for k in sb.params)
    assert all((sa.m1[k] - sb.m1[k]).abs().max() <= 1e-12 and (sa.m2[k] - sb.m2[k]).abs().max() <= 1e-12 for k in sb.m1)
    # 5 + 3 iterations through a saved state = 8
    mc, dc = sites()
    _, st = multisite_fit.fit_many(mc, dc, iterations=5, site_seeds=seeds, return_state=True)
    oc = multisite_fit.fit_many(mc, dc, iterations=3, site_seeds=seeds, resume=st)
    assert torch.equal(flat(mc), flat(ma)) and torch.equal(oc, oa)
    # a penalty term needs autograd: the closed form steps aside
    md, dd = sites()
    n_before = len(used)
    multisite_fit.fit_many(md, dd, iterations=1, site_seeds=seeds, penalty_callback=lambda i, p: sum((v ** 2).sum() for v in p.values()),
                           penalty_weight=0.1)
    assert len(used) == n_before
