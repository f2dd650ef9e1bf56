"""Engine-level parity on the GPU: MarginalHIP with the real device plan vs the CPU oracle."""
import io

import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc
from tests.helpers import loadest_dataset, rating_dataset
from tests.test_engine_cpu import _loadest_raw_from_model

pytestmark = pytest.mark.gpu


def test_loadest_fit_objective_predict_sample(gpu_device):
    from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood
    from discontinuum_amd.loadest_gp import LoadestGP

    torch.manual_seed(0)
    cov, tgt = loadest_dataset(300)
    m = LoadestGP()
    m.fit(cov, tgt, iterations=10)
    assert m.is_fitted
    m.model.zero_grad(set_to_none=True)
    obj = -ExactMarginalLogLikelihood(m.likelihood, m.model)(m._prior(), m._train_y)
    obj.sum().backward()
    raw = _loadest_raw_from_model(m).clone().requires_grad_(True)
    ref = orc.LoadestOracle(2).objective(raw, torch.tensor(m.X), torch.tensor(m.y))
    ref.backward()
    assert abs(obj.item() - ref.item()) < 1e-10 * max(1.0, abs(ref.item()))
    got = torch.cat([p.grad.reshape(-1) for p in (
        m.model.mean_module.raw_constant, *[q for q in m.model.covar_module.parameters()])])
    assert (got - raw.grad).abs().max() / raw.grad.abs().max() < 1e-8
    # prediction in model space vs the oracle, then the data-space wrappers
    mu_ref, var_ref = orc.LoadestOracle(2).predict(raw.detach(), torch.tensor(m.X), torch.tensor(m.y), torch.tensor(m.X))
    mu, var = m._model_space_predict(torch.tensor(m.X))
    assert (mu.cpu() - mu_ref).abs().max() < 1e-9
    assert ((var.cpu() - var_ref).abs() / (var_ref.abs() + 1e-4)).max() < 1e-7
    target, se = m.predict(cov)
    assert np.all(np.isfinite(target.values)) and np.all(se.values >= 1.0)
    grid = m.predict_grid("flow")
    assert grid.values.shape[1] == 18 and np.all(np.isfinite(grid.values))
    draws = m.sample(cov, n=64)
    assert draws.values.shape == (64, 300) and np.all(np.isfinite(draws.values))
    # the draws scatter around the posterior mean with the posterior spread (loose statistical check)
    z = (np.log(draws.values).mean(axis=0) - np.log(target.values)) / (np.log(draws.values).std(axis=0) / 8 + 1e-9)
    assert np.abs(z).mean() < 2.0
    # checkpoint round trip
    buf = io.BytesIO()
    m.save(buf)
    buf.seek(0)
    m2 = LoadestGP.load(buf, cov, tgt)
    t2, _ = m2.predict(cov)
    assert np.allclose(t2.values, target.values, rtol=1e-9)


def test_rating_fit_objective_predict(gpu_device):
    from discontinuum_amd.gp.lowering import lower
    from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood
    from discontinuum_amd.rating_gp import RatingGP

    torch.manual_seed(1)
    cov, tgt, unc = rating_dataset(200)
    m = RatingGP()
    m.fit(cov, tgt, target_unc=unc, iterations=8)
    obj = -ExactMarginalLogLikelihood(m.likelihood, m.model)(m._prior(), m._train_y)
    X, y, yu = torch.tensor(m.X), torch.tensor(m.y), torch.tensor(m.y_unc)
    o = orc.RatingOracle.from_stage(X[:, 1])
    theta = lower(m.model.covar_module, 2)[1]().detach()
    raw = torch.zeros(20, dtype=torch.float64)
    raw[0], raw[1], raw[2] = m.model.powerlaw.a.item(), m.model.powerlaw.b.item(), m.model.powerlaw.c.item()
    raw[3] = m.likelihood.second_noise_covar.raw_noise.item()
    raw[4] = orc.inv_interval(theta[0], o.b_lo, o.b_hi)
    raw[5:] = orc.inv_softplus(theta[1:])
    ref = o.objective(raw, X, y, yu)
    assert abs(obj.item() - ref.item()) < 1e-9 * max(1.0, abs(ref.item()))
    mu_ref, var_ref = o.predict(raw, X, y, X[:50].clone(), yu)
    mu, var = m._model_space_predict(X[:50].clone())
    assert (mu.cpu() - mu_ref).abs().max() < 1e-8
    assert ((var.cpu() - var_ref).abs() / (var_ref.abs() + 1e-4)).max() < 1e-6


# float32 at the ENGINE surface -- the reference's only dtype (engines/gpytorch.py:221-222).  Bounds are SURVEY.md section
# 8d's fp32 row: objective rel 1e-4 max(1, n / 1024), gradients rel 1e-2, posterior mean / variance abs 1e-3 -- against the
# fp64 oracle evaluated at the model's own (float32-valued) raw parameters and the model's own (float64) data.
def test_fp32_loadest_engine_matches_the_oracle(gpu_device):
    from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood
    from discontinuum_amd.loadest_gp import LoadestGP

    class LoadestGP32(LoadestGP):
        dtype = torch.float32

    torch.manual_seed(0)
    n = 1200
    cov, tgt = loadest_dataset(n)
    m = LoadestGP32()
    m.fit(cov, tgt, iterations=10)
    assert m.is_fitted and m._train_y.dtype == torch.float32
    m.model.zero_grad(set_to_none=True)
    obj = -ExactMarginalLogLikelihood(m.likelihood, m.model)(m._prior(), m._train_y)
    obj.sum().backward()
    raw = _loadest_raw_from_model(m).double().clone().requires_grad_(True)
    X, y = torch.tensor(m.X, dtype=torch.float64), torch.tensor(m.y, dtype=torch.float64)
    ref = orc.LoadestOracle(2).objective(raw, X, y)
    ref.backward()
    assert abs(obj.item() - ref.item()) <= 1e-4 * max(1.0, n / 1024) * abs(ref.item()), (obj.item(), ref.item())
    got = torch.cat([p.grad.reshape(-1).double() for p in (
        m.model.mean_module.raw_constant, *[q for q in m.model.covar_module.parameters()])])
    assert (got - raw.grad).abs().max() / raw.grad.abs().max() <= 1e-2
    mu_ref, var_ref = orc.LoadestOracle(2).predict(raw.detach(), X, y, X[:400].clone())
    mu, var = m._model_space_predict(torch.tensor(m.X[:400], dtype=torch.float32))
    assert mu.dtype == torch.float32
    assert (mu.cpu().double() - mu_ref).abs().max() <= 1e-3
    assert (var.cpu().double() - var_ref).abs().max() <= 1e-3
    target, se = m.predict(cov)
    assert np.all(np.isfinite(target.values)) and np.all(se.values >= 1.0)


def test_fp32_rating_engine_matches_the_oracle(gpu_device):
    from discontinuum_amd.gp.lowering import lower
    from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood
    from discontinuum_amd.rating_gp import RatingGP

    class RatingGP32(RatingGP):
        dtype = torch.float32

    torch.manual_seed(1)
    n = 900
    cov, tgt, unc = rating_dataset(n)
    m = RatingGP32()
    m.fit(cov, tgt, target_unc=unc, iterations=8)
    m.model.zero_grad(set_to_none=True)
    obj = -ExactMarginalLogLikelihood(m.likelihood, m.model)(m._prior(), m._train_y)
    obj.sum().backward()
    X, y, yu = (torch.tensor(a, dtype=torch.float64) for a in (m.X, m.y, m.y_unc))
    o = orc.RatingOracle.from_stage(X[:, 1])
    theta = lower(m.model.covar_module, 2)[1]().detach().double()
    raw = torch.zeros(20, dtype=torch.float64)
    raw[0], raw[1], raw[2] = m.model.powerlaw.a.item(), m.model.powerlaw.b.item(), m.model.powerlaw.c.item()
    raw[3] = m.likelihood.second_noise_covar.raw_noise.item()
    raw[4] = orc.inv_interval(theta[0], o.b_lo, o.b_hi)
    raw[5:] = orc.inv_softplus(theta[1:])
    raw.requires_grad_(True)
    ref = o.objective(raw, X, y, yu)
    ref.backward()
    assert abs(obj.item() - ref.item()) <= 1e-4 * abs(ref.item()) + 1e-5, (obj.item(), ref.item())
    # the mean's and the learned noise's gradients (host-side parameters fed by the device's dr / dnoise reductions)
    got = torch.tensor([m.model.powerlaw.a.grad.item(), m.model.powerlaw.b.grad.item(), m.model.powerlaw.c.grad.item(),
                        m.likelihood.second_noise_covar.raw_noise.grad.item()], dtype=torch.float64)
    assert (got - raw.grad[:4]).abs().max() <= 1e-2 * raw.grad[:4].abs().max()
    mu_ref, var_ref = o.predict(raw.detach(), X, y, X[:200].clone(), yu)
    mu, var = m._model_space_predict(torch.tensor(m.X[:200], dtype=torch.float32))
    assert (mu.cpu().double() - mu_ref).abs().max() <= 1e-3
    assert (var.cpu().double() - var_ref).abs().max() <= 1e-3


def test_rating_monotonic_penalty_fit(gpu_device):
    """The reference's third smoke test (tests/test_rating_gp.py:52-65): fit with the monotonicity penalty."""
    from discontinuum_amd.rating_gp import RatingGP

    torch.manual_seed(2)
    cov, tgt, unc = rating_dataset(120)
    m = RatingGP()
    m.fit(cov, tgt, target_unc=unc, iterations=5, monotonic_penalty_weight=0.5, grid_size=16, scheduler=False)
    assert m.is_fitted
    m.model.zero_grad(set_to_none=True)
    # penalty gradient check against the oracle at the fitted parameters (one extra evaluation)
    from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood

    _ = -ExactMarginalLogLikelihood(m.likelihood, m.model)(m._prior(), m._train_y)  # refresh the plan state
    x = torch.tensor(m.X[:10]).clone()
    mu = m._differentiable_mean(x)
    mu.sum().backward()
    grads = [p.grad for p in m.model.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    target, _ = m.predict(cov)
    assert np.all(np.isfinite(target.values))


def test_fit_sites_two_plans_match_sequential(gpu_device):
    """Config-4 style batches: sites dealt over two plans/streams, or carried by one batched plan, give the same rows
    as one plan."""
    from discontinuum_amd.backend import GPPlan
    from discontinuum_amd.sites import fit_sites

    n, d, dev = 600, 3, gpu_device
    data = []
    for i in range(5):
        X, y = orc.synth_loadest(n, d, seed=40 + i)
        data.append((torch.tensor(X, device=dev), torch.tensor(y, device=dev)))
    noise = torch.full((n,), 0.01, dtype=torch.float64, device=dev)
    theta = [0.6931471805599453] * 11
    one = fit_sites(GPPlan("loadest", n, d, device=dev), [x for x, _ in data], [y for _, y in data], [noise] * 5, theta)
    two = fit_sites([GPPlan("loadest", n, d, device=dev) for _ in range(2)], [x for x, _ in data], [y for _, y in data],
                    [noise] * 5, theta)
    bat = fit_sites(GPPlan("loadest", n, d, device=dev, lookahead=1, batch=4), [x for x, _ in data], [y for _, y in data],
                    [noise] * 5, theta)  # 5 sites through a batch of 4: one full chunk and one padded chunk
    # several BATCHED plans, a contiguous share of the sites each on its own stream (3 + 2 sites over two plans of 3; and a
    # third plan that gets no site at all)
    bat2 = fit_sites([GPPlan("loadest", n, d, device=dev, lookahead=1, batch=3) for _ in range(2)], [x for x, _ in data],
                     [y for _, y in data], [noise] * 5, theta)
    bat3 = fit_sites([GPPlan("loadest", n, d, device=dev, lookahead=1, batch=3) for _ in range(3)], [x for x, _ in data][:4],
                     [y for _, y in data][:4], [noise] * 4, theta)
    torch.cuda.synchronize()
    assert torch.equal(one, two)
    assert bat.shape == one.shape and (bat - one).abs().max() <= 1e-10 * one.abs().max()
    assert bat2.shape == one.shape and (bat2 - one).abs().max() <= 1e-10 * one.abs().max()
    assert bat3.shape == one[:4].shape and (bat3 - one[:4]).abs().max() <= 1e-10 * one.abs().max()
    # ragged sites through the same batched plan: each row must match a plan of that site's own size
    sizes = [600, 433, 600, 128, 57]
    rag = fit_sites(GPPlan("loadest", n, d, device=dev, lookahead=1, batch=4), [x[:k] for (x, _), k in zip(data, sizes)],
                    [y[:k] for (_, y), k in zip(data, sizes)], [noise[:k] for k in sizes], theta)
    for i in (1, 3, 4):
        solo = fit_sites(GPPlan("loadest", sizes[i], d, device=dev), [data[i][0][:sizes[i]]], [data[i][1][:sizes[i]]],
                         [noise[:sizes[i]]], theta)
        assert (rag[i] - solo[0]).abs().max() <= 1e-9 * solo[0].abs().max()
    val, g, _, _ = orc.nll_data_and_grads("loadest", data[3][0].cpu(), data[3][1].cpu(), noise.cpu(), torch.tensor(theta, dtype=torch.float64))
    assert abs(two[3, 0].item() - val.item()) / abs(val.item()) < 1e-10


def test_fit_many_follows_the_single_site_trajectories(gpu_device):
    """``fit_many`` (one batched plan, vectorised host algebra, per-site Adam / clipping / plateau schedule) must land
    where ``model.fit`` lands for each site on its own -- sites of different length, scheduler on."""
    from discontinuum_amd.loadest_gp import LoadestGP
    from discontinuum_amd.multisite_fit import fit_many

    sizes, iters = [90, 64, 130, 75], 40
    data = [loadest_dataset(k, seed=300 + i) for i, k in enumerate(sizes)]
    solo = []
    for cov, tgt in data:
        m = LoadestGP()
        m.fit(cov, tgt, iterations=iters)
        solo.append(m)
    many = [LoadestGP() for _ in sizes]
    final = fit_many(many, data, iterations=iters)
    assert final.shape == (len(sizes),) and bool(torch.isfinite(final).all())
    for a, b, (cov, _tgt) in zip(solo, many, data):
        assert b.is_fitted
        pa = torch.cat([p.detach().reshape(-1) for p in a.model.parameters()])
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        assert (pa - pb).abs().max() < 1e-6, (pa - pb).abs().max()
        ta, sa = a.predict(cov)
        tb, sb = b.predict(cov)
        assert np.allclose(ta.values, tb.values, rtol=1e-6) and np.allclose(sa.values, sb.values, rtol=1e-6)


def test_fit_many_rating_follows_the_single_site_trajectories(gpu_device):
    """rating-gp through ``fit_many``: power-law mean with its per-iteration clamps, learned extra noise, per-site gate
    interval and measurement variances -- against ``model.fit`` per site (identical initial draws per site)."""
    from discontinuum_amd.multisite_fit import fit_many
    from discontinuum_amd.rating_gp import RatingGP

    class Seeded(RatingGP):
        seed = 0

        def build_model(self, *args):
            torch.manual_seed(self.seed)
            return super().build_model(*args)

    sizes, iters = [70, 48, 96], 30
    data = [rating_dataset(k, seed=500 + i) for i, k in enumerate(sizes)]

    def new(i):
        m = Seeded()
        m.seed = 900 + i
        return m

    solo = []
    for i, (cov, tgt, unc) in enumerate(data):
        m = new(i)
        m.fit(cov, tgt, target_unc=unc, iterations=iters)
        solo.append(m)
    many = [new(i) for i in range(len(sizes))]
    final = fit_many(many, data, iterations=iters)
    assert bool(torch.isfinite(final).all())
    for a, b, (cov, _tgt, _unc) in zip(solo, many, data):
        pa = torch.cat([p.detach().reshape(-1) for p in a.model.parameters()])
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        assert (pa - pb).abs().max() < 1e-6, (pa - pb).abs().max()
        ta, _ = a.predict(cov)
        tb, _ = b.predict(cov)
        assert np.allclose(ta.values, tb.values, rtol=1e-6)


def test_fit_many_early_stopping_freezes_each_site_where_its_own_fit_stops(gpu_device):
    """Per-site early stopping: every site must stop at the iteration its own ``model.fit(early_stopping=True)`` stops
    at (a large learning rate makes the objectives oscillate, so the sites do stop, and at different iterations)."""
    from discontinuum_amd.loadest_gp import LoadestGP
    from discontinuum_amd.multisite_fit import fit_many

    sizes, iters, lr, patience = [60, 90, 45, 75], 120, 0.4, 4
    data = [loadest_dataset(k, seed=700 + i) for i, k in enumerate(sizes)]
    solo = []
    for cov, tgt in data:
        m = LoadestGP()
        m.fit(cov, tgt, iterations=iters, learning_rate=lr, early_stopping=True, patience=patience)
        solo.append(m)
    many = [LoadestGP() for _ in sizes]
    fit_many(many, data, iterations=iters, learning_rate=lr, early_stopping=True, patience=patience)
    stops = [m._current_iteration for m in solo]
    assert [m._current_iteration for m in many] == stops, (stops, [m._current_iteration for m in many])
    assert min(stops) < iters - 1, stops  # the scenario does exercise the stop
    for a, b in zip(solo, many):
        pa = torch.cat([p.detach().reshape(-1) for p in a.model.parameters()])
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        assert (pa - pb).abs().max() < 1e-6, (pa - pb).abs().max()


def test_fit_many_with_a_single_site_uses_the_unbatched_plan(gpu_device):
    """One site through ``fit_many`` (unbatched plan, (2, n) weight vectors) ends where a batch of two puts it."""
    from discontinuum_amd.multisite_fit import fit_many
    from discontinuum_amd.rating_gp import RatingGP

    class Seeded(RatingGP):
        seed = 0

        def build_model(self, *args):
            torch.manual_seed(self.seed)
            return super().build_model(*args)

    data = [rating_dataset(64, seed=41), rating_dataset(50, seed=42)]

    def new(i):
        m = Seeded()
        m.seed = 77 + i
        return m

    pair = [new(0), new(1)]
    fit_many(pair, data, iterations=25)
    alone = [new(0)]
    fit_many(alone, data[:1], iterations=25)
    pa = torch.cat([p.detach().reshape(-1) for p in pair[0].model.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in alone[0].model.parameters()])
    assert (pa - pb).abs().max() < 1e-6, (pa - pb).abs().max()


@pytest.mark.parametrize("family", ["loadest", "rating"])
def test_predict_many_matches_per_site_predict(family, gpu_device):
    """``predict_many``: one batched factorisation + one batched ``dgp_predict`` for all sites (ragged in n and in m)
    gives what each site's own ``predict`` gives (its own plan, its own factorisation): rel 1e-9."""
    from discontinuum_amd.multisite_fit import fit_many, predict_many

    if family == "loadest":
        from discontinuum_amd.loadest_gp import LoadestGP as Model

        data = [loadest_dataset(k, seed=800 + i) for i, k in enumerate([90, 64, 130])]
        new = [loadest_dataset(k, seed=850 + i)[0] for i, k in enumerate([40, 75, 33])]
    else:
        from discontinuum_amd.rating_gp import RatingGP as Model

        data = [rating_dataset(k, seed=800 + i) for i, k in enumerate([70, 48, 96])]
        new = [rating_dataset(k, seed=850 + i)[0] for i, k in enumerate([30, 51, 18])]
    torch.manual_seed(4)
    models = [Model() for _ in data]
    fit_many(models, data, iterations=15)
    both = predict_many(models, new)
    alone = predict_many(models[:1], new[:1])
    for m, cov, (t_b, se_b) in zip(models, new, both):
        t_s, se_s = m.predict(cov)
        assert np.allclose(t_b.values, t_s.values, rtol=1e-9) and np.allclose(se_b.values, se_s.values, rtol=1e-9)
        assert list(t_b.coords) == list(t_s.coords)
    assert np.allclose(alone[0][0].values, both[0][0].values, rtol=1e-9)


def test_fit_many_monotonic_penalty_follows_the_single_site_trajectories(gpu_device):
    """The rating-gp monotonicity penalty inside ``fit_many`` -- every site's differentiable posterior mean through ONE
    batched ``dgp_predict_mean`` / ``dgp_mean_vjp`` per iteration -- against ``RatingGP.fit(monotonic_penalty_weight=...)``
    per site, on the SAME penalty grids (both draw their uniforms from a per-site, per-call table here)."""
    from discontinuum_amd.multisite_fit import fit_many
    from discontinuum_amd.rating_gp import RatingGP
    from discontinuum_amd.rating_gp import models as rmod

    class Seeded(RatingGP):
        seed = 0

        def build_model(self, *args):
            torch.manual_seed(self.seed)
            return super().build_model(*args)

    sizes, iters, m, weight = [70, 48, 96], 25, 16, 5.0
    data = []
    for i, k in enumerate(sizes):  # a rating that BENDS DOWN over the upper half of the stages: the penalty is active
        cov, tgt, unc = rating_dataset(k, seed=500 + i)
        stage = cov["stage"].values
        bent = tgt.values * np.exp(-1.5 * np.maximum(stage - np.median(stage), 0.0) ** 2)
        data.append((cov, type(tgt)(bent, dims=tgt.dims, coords={"time": tgt.coords["time"]}, name=tgt.name, attrs=dict(tgt.attrs)), unc))
    table = torch.rand((len(sizes), iters, 2, m), dtype=torch.float64, generator=torch.Generator().manual_seed(11))

    def new(i):
        mod = Seeded()
        mod.seed = 900 + i
        return mod

    solo, penalties = [], []
    try:
        for i, (cov, tgt, unc) in enumerate(data):
            calls = {"k": 0}

            def draw(mm, i=i, calls=calls):
                calls["k"] += 1
                return table[i, calls["k"] - 1]

            rmod._MonotonicPenalty.uniforms = staticmethod(draw)
            mod = new(i)
            mod.fit(cov, tgt, target_unc=unc, iterations=iters, monotonic_penalty_weight=weight, grid_size=m)
            solo.append(mod)
    finally:
        rmod._MonotonicPenalty.uniforms = None
    calls = {"k": 0}

    def draw_all(mm):
        calls["k"] += 1
        return table[:, calls["k"] - 1]

    many = [new(i) for i in range(len(sizes))]
    final = fit_many(many, data, iterations=iters, monotonic_penalty_weight=weight, grid_size=m, _penalty_uniforms=draw_all)
    assert bool(torch.isfinite(final).all()) and calls["k"] == iters
    plain = [new(i) for i in range(len(sizes))]
    fit_many(plain, data, iterations=iters)
    for a, b, c in zip(solo, many, plain):
        pa = torch.cat([p.detach().reshape(-1) for p in a.model.parameters()])
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        pc = torch.cat([p.detach().reshape(-1) for p in c.model.parameters()])
        assert (pa - pb).abs().max() < 1e-5, (pa - pb).abs().max()
        assert (pb - pc).abs().max() > 1e-4  # the penalty does steer the fit


def test_fit_many_resume_continues_the_run(gpu_device):
    """40 iterations in one call = 25 + 15 through ``return_state`` / ``resume`` (optimiser moments, learning rates, plateau
    and early-stopping counters travel in ``FitManyState``; its dictionary form survives ``torch.save`` / ``weights_only``)."""
    import io

    from discontinuum_amd.loadest_gp import LoadestGP
    from discontinuum_amd.multisite_fit import FitManyState, fit_many

    sizes = [80, 64, 120]
    data = [loadest_dataset(k, seed=40 + i) for i, k in enumerate(sizes)]
    a = [LoadestGP() for _ in sizes]
    fa = fit_many(a, data, iterations=40)
    b = [LoadestGP() for _ in sizes]
    _, st = fit_many(b, data, iterations=25, return_state=True)
    buf = io.BytesIO()
    torch.save(st.as_dict(), buf)
    buf.seek(0)
    st2 = FitManyState.from_dict(torch.load(buf, weights_only=True))
    assert torch.equal(st2.nan_run, torch.zeros(len(sizes), dtype=torch.float64))  # the NaN counter travels with the state
    fb, st3 = fit_many(b, data, iterations=15, resume=st2, return_state=True)
    assert torch.equal(fa, fb) and int(st3.iterations_done) == 40 and torch.equal(st3.nan_run, st2.nan_run)
    for x, y in zip(a, b):
        px = torch.cat([p.detach().reshape(-1) for p in x.model.parameters()])
        py = torch.cat([p.detach().reshape(-1) for p in y.model.parameters()])
        assert torch.equal(px, py)
    with pytest.raises(ValueError):
        fit_many([LoadestGP()], data[:1], iterations=1, resume=st)


def test_fit_many_with_an_arbitrary_penalty_callback_follows_model_fit(gpu_device):
    """The reference's optional penalty term (engines/gpytorch.py:362-373) for many sites at once: a user callback --
    here a pull of the residual kernel's lengthscales towards 0.5 plus an ordering penalty on two outputscales -- through
    ``fit_many(penalty_callback=f, penalty_weight=w)`` against ``model.fit(penalty_callback=closure, penalty_weight=w)`` per
    site (parameters to 1e-6); a callback that raises, or returns a non-tensor, is ignored like the reference ignores it."""
    import torch.nn.functional as F

    from discontinuum_amd.loadest_gp import LoadestGP
    from discontinuum_amd.multisite_fit import fit_many

    sizes, iters, w = [70, 96, 64], 30, 0.7
    data = [loadest_dataset(k, seed=500 + i) for i, k in enumerate(sizes)]

    def penalty(named):  # named: {parameter name (relative to the ExactGP model): raw tensor}
        ls = F.softplus(named["covar_module.kernels.2.base_kernel.raw_lengthscale"])
        os0 = F.softplus(named["covar_module.kernels.0.raw_outputscale"])
        os1 = F.softplus(named["covar_module.kernels.1.raw_outputscale"])
        return ((ls - 0.5) ** 2).sum() + torch.relu(os1 - os0).sum()

    probe = LoadestGP()
    probe.fit(*data[0], iterations=1)
    names = dict(probe.model.named_parameters())
    assert "covar_module.kernels.2.base_kernel.raw_lengthscale" in names and "covar_module.kernels.0.raw_outputscale" in names
    solo, plain = [], []
    for cov, tgt in data:
        m = LoadestGP()
        m.fit(cov, tgt, iterations=iters, penalty_callback=lambda m=m: penalty(dict(m.model.named_parameters())), penalty_weight=w)
        solo.append(m)
        q = LoadestGP()
        q.fit(cov, tgt, iterations=iters)
        plain.append(q)
    many = [LoadestGP() for _ in sizes]
    fit_many(many, data, iterations=iters, penalty_weight=w,
             penalty_callback=lambda b, params: penalty({k[len("model."):]: v for k, v in params.items() if k.startswith("model.")}))
    for a, b, c in zip(solo, many, plain):
        pa = torch.cat([p.detach().reshape(-1) for p in a.model.parameters()])
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        pc = torch.cat([p.detach().reshape(-1) for p in c.model.parameters()])
        assert (pa - pb).abs().max() < 1e-6, (pa - pb).abs().max()
        assert (pb - pc).abs().max() > 1e-3  # the penalty does steer the fit
    # failures of the callback are swallowed, like the reference does (the fit is then the plain one)

    def broken(b, params):
        if b == 1:
            raise RuntimeError("boom")
        return 3.0  # not a tensor

    again = [LoadestGP() for _ in sizes]
    fit_many(again, data, iterations=iters, penalty_callback=broken, penalty_weight=w)
    for b, c in zip(again, plain):
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        pc = torch.cat([p.detach().reshape(-1) for p in c.model.parameters()])
        assert (pb - pc).abs().max() < 1e-6


def test_fit_many_adamw_follows_the_single_site_trajectories(gpu_device):
    """``fit_many(optimizer="adamw")`` (decoupled weight decay 1e-2) against ``model.fit(optimizer="adamw")`` per site."""
    from discontinuum_amd.loadest_gp import LoadestGP
    from discontinuum_amd.multisite_fit import fit_many

    sizes, iters = [90, 64, 130], 30
    data = [loadest_dataset(k, seed=320 + i) for i, k in enumerate(sizes)]
    solo = []
    for cov, tgt in data:
        m = LoadestGP()
        m.fit(cov, tgt, iterations=iters, optimizer="adamw")
        solo.append(m)
    many = [LoadestGP() for _ in sizes]
    fit_many(many, data, iterations=iters, optimizer="adamw")
    ref = [LoadestGP() for _ in sizes]
    fit_many(ref, data, iterations=iters)
    for a, b, c in zip(solo, many, ref):
        pa = torch.cat([p.detach().reshape(-1) for p in a.model.parameters()])
        pb = torch.cat([p.detach().reshape(-1) for p in b.model.parameters()])
        pc = torch.cat([p.detach().reshape(-1) for p in c.model.parameters()])
        assert (pa - pb).abs().max() < 1e-6, (pa - pb).abs().max()
        assert (pb - pc).abs().max() > 1e-5  # and it is not the Adam trajectory
    with pytest.raises(ValueError):
        fit_many(many, data, iterations=1, optimizer="sgd")
