"""GPU parity of the ``sample()`` path and of batched inference, through the C ABI.

What the reference does (``src/discontinuum/engines/gpytorch.py:551-593``): ``f_preds = self.model(Xnew)`` -- the
latent posterior N(K*^T alpha, K** - K*^T K^^-1 K*) -- and ``f_preds.sample(torch.Size([n]))``, i.e. the covariance's
Cholesky factor (linear_operator's ``psd_safe_cholesky``: the matrix itself first, then jitter 1e-8 / 1e-7 / 1e-6 in
fp64) times standard normals.  Here: ``dgp_posterior_cov`` -> blocked HIP potrf with the same jitter policy ->
``dgp_sample_draws``.  Tolerances (fp64, vs ``oracle.posterior(..., full_cov=True)``): mean abs 1e-9, covariance
1e-8 relative to its largest entry, factor reconstruction L L^T = cov + jitter I to 1e-10 relative, draws against
the dense product of the same factor and the same normals 1e-12 relative.
"""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc
from tests.test_gpu_stages import make_case, plan_for

pytestmark = pytest.mark.gpu


def _lower(t, m):
    return torch.tril(t[:m, :m]).cpu().double()


@pytest.mark.parametrize("model,d,n,m", [
    ("loadest", 3, 300, 1), ("loadest", 3, 300, 130), ("loadest", 2, 200, 300), ("loadest", 3, 500, 1000),
    ("loadest", 3, 256, 256),  # m == n (the likelihood's noise quirk lives above this level: latent f has no noise)
    ("rating", 2, 300, 1), ("rating", 2, 300, 130), ("rating", 2, 200, 300), ("rating", 2, 400, 1000),
    ("rating", 2, 128, 128),
])
def test_posterior_cov_fp64(model, d, n, m, gpu_device):
    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=21, perturb=0.2)
    Xs, *_ = make_case(model, d, m, seed=22)
    mu_ref, cov_ref = orc.posterior(model, X, r, noise, theta, Xs, full_cov=True)
    p = plan_for(model, d, n, X, torch.float64, dev)
    with pytest.raises(Exception):  # no factorisation yet
        p.posterior_cov(theta, Xs.to(dev))
    p.factorize(theta, r.to(dev), noise.to(dev))
    mean, cov = p.posterior_cov(theta, Xs.to(dev))
    M = cov.shape[0]
    assert M % 128 == 0 and M >= m
    assert (mean.cpu() - mu_ref).abs().max() < 1e-9
    scale = cov_ref.abs().max()
    assert (_lower(cov, m) - torch.tril(cov_ref)).abs().max() <= 1e-8 * scale
    if M > m:  # identity pad: the factor of blockdiag(C, I) is blockdiag(L, I)
        pad = cov[m:, :].cpu()
        assert torch.equal(torch.tril(pad[:, m:]), torch.eye(M - m, dtype=torch.float64))
        assert pad[:, :m].abs().max() == 0
    # the same call after a full fit step (K^^-1 present) gives the same matrix
    p.fit_step(theta, r.to(dev), noise.to(dev))
    mean2, cov2 = p.posterior_cov(theta, Xs.to(dev))
    assert torch.equal(mean2, mean) and torch.equal(_lower(cov2, m), _lower(cov, m))


@pytest.mark.parametrize("model,d,sizes,m", [
    ("loadest", 3, (300, 300, 300), 130), ("rating", 2, (260, 200, 131), 300),
    ("loadest", 2, (200, 150, 128, 200, 90, 200, 177, 200, 64, 200), 70),  # > 8 sites: hyperparameters through the device scratch
])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_posterior_cov_of_a_batched_plan(model, d, sizes, m, dtype, gpu_device):
    """dgp_posterior_cov on a (ragged) batched plan -- one launch sequence, cov [batch][M][M] -- site by site against the
    oracle (fp64: the single plan's tolerances; fp32: mean 1e-3, covariance 2e-3 of its largest entry) and against a
    single-site plan holding site 0 (1e-12 / 1e-4 relative)."""
    from discontinuum_amd.backend import GPPlan

    dev, B, n = gpu_device, len(sizes), max(sizes)
    f64 = dtype == torch.float64
    cases = [make_case(model, d, nb, seed=60 + b, perturb=0.2) for b, nb in enumerate(sizes)]
    X = torch.zeros(B, n, d, dtype=torch.float64)
    r = torch.zeros(B, n, dtype=torch.float64)
    noise = torch.ones(B, n, dtype=torch.float64)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        X[b, :nb], r[b, :nb], noise[b, :nb] = c[0], c[1], c[2]
    theta = torch.stack([c[3] for c in cases])
    Xs = torch.stack([make_case(model, d, m, seed=80 + b)[0] for b in range(B)])
    pb = GPPlan(model, n, d, dtype=dtype, device=dev, lookahead=1, batch=B)
    pb.set_site_sizes(sizes)
    pb.set_inputs(X.to(dev, dtype).contiguous())
    pb.factorize(theta, r.to(dev, dtype).contiguous(), noise.to(dev, dtype).contiguous())
    mean, cov = pb.posterior_cov(theta, Xs.to(dev, dtype))
    M = cov.shape[-1]
    assert mean.shape == (B, m) and cov.shape == (B, M, M)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        mu_ref, cov_ref = orc.posterior(model, c[0], c[1], c[2], c[3], Xs[b], full_cov=True)
        scale = cov_ref.abs().max()
        assert (mean[b].cpu().double() - mu_ref).abs().max() < (1e-9 if f64 else 1e-3)
        assert (_lower(cov[b], m) - torch.tril(cov_ref)).abs().max() <= (1e-8 if f64 else 2e-3) * scale
        if M > m:
            pad = cov[b, m:, :].cpu()
            assert torch.equal(torch.tril(pad[:, m:]), torch.eye(M - m, dtype=dtype)) and pad[:, :m].abs().max() == 0
    # site 0 (a full-size one) alone in a single-site plan: same kernels, same tile order
    ps = GPPlan(model, n, d, dtype=dtype, device=dev, lookahead=1)
    ps.set_inputs(X[0].to(dev, dtype).contiguous())
    ps.factorize(theta[0], r[0].to(dev, dtype).contiguous(), noise[0].to(dev, dtype).contiguous())
    mean1, cov1 = ps.posterior_cov(theta[0], Xs[0].to(dev, dtype))
    tol = 1e-12 if f64 else 1e-4
    assert (mean1 - mean[0]).abs().max() <= tol * max(1.0, float(mean1.abs().max()))
    assert (_lower(cov1, m) - _lower(cov[0], m)).abs().max() <= tol * float(_lower(cov1, m).abs().max())


@pytest.mark.parametrize("model,d,n,m", [("loadest", 3, 300, 130), ("rating", 2, 200, 300), ("loadest", 3, 400, 700)])
def test_posterior_factor_reproduces_the_covariance(model, d, n, m, gpu_device):
    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=31, perturb=0.2)
    Xs, *_ = make_case(model, d, m, seed=32)
    _, cov_ref = orc.posterior(model, X, r, noise, theta, Xs, full_cov=True)
    p = plan_for(model, d, n, X, torch.float64, dev)
    p.factorize(theta, r.to(dev), noise.to(dev))
    mean, cov = p.posterior_cov(theta, Xs.to(dev))
    keep = cov.clone()
    Lbuf, jitter = p.psd_safe_factor(cov, m)
    assert torch.equal(cov, keep)  # the ladder restarts from the kept matrix, which it never modifies
    assert jitter in (0.0, 1e-8, 1e-7, 1e-6)
    L = _lower(Lbuf, m)
    want = _lower(cov, m)
    want = want + want.T - torch.diag(torch.diagonal(want)) + jitter * torch.eye(m, dtype=torch.float64)
    assert (L @ L.T - want).abs().max() <= 1e-10 * want.abs().max()
    # zeros above the diagonal inside the diagonal 128-blocks, identity pad (what dgp_sample_draws relies on)
    full = Lbuf.cpu()
    for c in range(0, full.shape[0], 128):
        blk = full[c:c + 128, c:c + 128]
        assert torch.equal(torch.triu(blk, 1), torch.zeros_like(blk))
    M = full.shape[0]
    if M > m:
        assert torch.equal(torch.tril(full[m:, m:]), torch.eye(M - m, dtype=torch.float64))
    # against the oracle's own factor when no jitter was needed (both are THE Cholesky factor of nearly the same matrix)
    if jitter == 0.0:
        Lref, info = torch.linalg.cholesky_ex(cov_ref)
        if int(info) == 0:
            assert (L @ L.T - Lref @ Lref.T).abs().max() <= 1e-8 * cov_ref.abs().max()


@pytest.mark.parametrize("shift,expect", [(0.0, 0.0), (-3e-9, 1e-8), (-3e-8, 1e-7), (-3e-7, 1e-6)])
def test_jitter_ladder_walks_like_psd_safe_cholesky(shift, expect, gpu_device):
    """A symmetric matrix whose smallest eigenvalue is ``shift`` (exactly constructed): the factorisation must succeed
    at the first rung of (0, 1e-8, 1e-7, 1e-6) that makes it positive definite, report that rung, and factor exactly
    matrix + rung * I; below the last rung it raises."""
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    m, M = 200, 256
    g = torch.Generator().manual_seed(5)
    Q, _ = torch.linalg.qr(torch.randn(m, m, generator=g, dtype=torch.float64))
    lam = torch.linspace(0.5, 2.0, m, dtype=torch.float64)
    lam[0] = shift if shift != 0.0 else 0.5
    C = (Q * lam) @ Q.T
    C = 0.5 * (C + C.T)
    cov = torch.eye(M, dtype=torch.float64)
    cov[:m, :m] = C
    p = GPPlan("loadest", 64, 2, device=dev)  # any plan: the ladder only uses its dtype / device / order-m helper plan
    Lbuf, jitter = p.psd_safe_factor(cov.to(dev).contiguous(), m)
    assert jitter == expect
    L = _lower(Lbuf, m)
    want = C + jitter * torch.eye(m, dtype=torch.float64)
    assert (L @ L.T - want).abs().max() <= 1e-12 * want.abs().max()
    if shift == -3e-7:
        lam[0] = -3e-5
        C2 = (Q * lam) @ Q.T
        cov[:m, :m] = 0.5 * (C2 + C2.T)
        with pytest.raises(RuntimeError, match="not positive definite"):
            p.psd_safe_factor(cov.to(dev).contiguous(), m)


@pytest.mark.parametrize("dtype,m,ndraw", [(torch.float64, 130, 64), (torch.float64, 300, 1000), (torch.float64, 1, 5),
                                           (torch.float64, 700, 129), (torch.float32, 300, 200)])
def test_sample_draws_is_mean_plus_L_z(dtype, m, ndraw, gpu_device):
    """``dgp_sample_draws`` against the dense product of the same factor and the same normals (torch fp64 on the host)."""
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    M = (m + 127) // 128 * 128
    g = torch.Generator().manual_seed(m)
    A = torch.randn(m, m, generator=g, dtype=torch.float64)
    L = torch.tril(A) * 0.1 + torch.eye(m, dtype=torch.float64)
    Lbuf = torch.full((M, M), float("nan"), dtype=torch.float64)  # blocks above the block diagonal are never read
    for c in range(0, M, 128):
        Lbuf[c:, c:c + 128] = 0.0
    Lbuf[:m, :m] = torch.where(torch.tril(torch.ones(m, m, dtype=torch.bool)), L, Lbuf[:m, :m])
    for c in range(0, M, 128):  # inside the diagonal blocks the upper triangle holds zeros
        blk = Lbuf[c:c + 128, c:c + 128]
        blk.copy_(torch.tril(blk.nan_to_num(0.0)))
    Lbuf[m:, m:] = torch.eye(M - m, dtype=torch.float64)
    mean = torch.randn(m, generator=g, dtype=torch.float64)
    p = GPPlan("loadest", 64, 2, dtype=dtype, device=dev)
    out = p.sample_draws(Lbuf.to(dev, dtype).contiguous(), m, mean.to(dev, dtype), ndraw)
    assert out.shape == (ndraw, m) and out.is_contiguous()
    z = p._last_z[:m, :ndraw].cpu().double()
    Lc = torch.tril(Lbuf[:m, :m].to(dtype).double())
    want = (mean.to(dtype).double()[:, None] + Lc @ z).T
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert (out.cpu().double() - want).abs().max() <= tol * want.abs().max()
    assert torch.isfinite(out).all()
    # without a mean
    out0 = p.sample_draws(Lbuf.to(dev, dtype).contiguous(), m, None, 3)
    z0 = p._last_z[:m, :3].cpu().double()
    assert (out0.cpu().double() - (Lc @ z0).T).abs().max() <= tol * want.abs().max()


def test_engine_sample_is_the_reference_recipe(gpu_device):
    """``MarginalHIP.sample`` end to end: the draws are y_t(mean + L z) with the plan's own factor and normals, the
    factor reproduces the oracle's latent covariance (+ jitter), and the sample moments agree with the oracle's
    posterior (4000 draws: standard error of a mean is sd / 63)."""
    from discontinuum_amd.loadest_gp import LoadestGP
    from tests.helpers import loadest_dataset
    from tests.test_engine_cpu import _loadest_raw_from_model

    torch.manual_seed(0)
    cov_ds, tgt = loadest_dataset(300)
    m = LoadestGP()
    m.fit(cov_ds, tgt, iterations=10)
    new_cov, _ = loadest_dataset(170, seed=9)
    ndraw = 4000
    draws = m.sample(new_cov, n=ndraw)
    assert draws.values.shape == (ndraw, 170) and np.all(np.isfinite(draws.values))
    Xnew = torch.tensor(m.dm.Xnew(new_cov))
    raw = _loadest_raw_from_model(m).detach()
    o = orc.LoadestOracle(2)
    X, y = torch.tensor(m.X), torch.tensor(m.y)
    theta = o.constrained(raw)
    mu_ref, cov_ref = orc.posterior("loadest", X, y - o.mean(raw, X), o.noise(raw, X.shape[0]), theta, Xnew, full_cov=True)
    mu_ref = mu_ref + o.mean(raw, Xnew)
    plan = m._plan
    Lbuf = plan._fac.buffer(1)
    L = _lower(Lbuf, 170)
    z = plan._last_z[:170, :ndraw].cpu()
    model_space = (mu_ref[:, None] + L @ z).T
    want = np.asarray(m.dm.y_t(model_space.reshape(-1).numpy()).data).reshape(ndraw, 170)
    assert np.allclose(draws.values, want, rtol=1e-8, atol=0)
    LLt = L @ L.T
    jit = (torch.diagonal(LLt) - torch.diagonal(cov_ref)).mean().item()
    assert -1e-9 < jit < 2e-6  # one of the rungs 0 / 1e-8 / 1e-7 / 1e-6 (plus rounding)
    assert (LLt - cov_ref - jit * torch.eye(170, dtype=torch.float64)).abs().max() <= 1e-8 * cov_ref.abs().max() + 1e-9
    # moments of the log-space draws
    logd = torch.tensor(np.log(draws.values))
    pipeline_mu = torch.tensor(np.log(np.asarray(m.dm.y_t(mu_ref.numpy()).data)))
    scale = (logd.std(dim=0) / np.sqrt(ndraw)).clamp_min(1e-12)
    assert ((logd.mean(dim=0) - pipeline_mu) / scale).abs().max() < 6.0


@pytest.mark.parametrize("model,d", [("loadest", 3), ("rating", 2)])
def test_batched_plan_predicts_every_site(model, d, gpu_device):
    """``dgp_predict`` / ``dgp_predict_mean`` on a batched, ragged plan: each site's posterior at its own test points from
    the factorisation the batched fit step left behind -- against the oracle, and against a single-site plan."""
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    n, m, sizes = 300, 150, [300, 211, 128, 57]
    B = len(sizes)
    cases = [make_case(model, d, k, seed=60 + i, perturb=0.2) for i, k in enumerate(sizes)]
    tests = [make_case(model, d, m, seed=80 + i)[0] for i in range(B)]
    P = cases[0][3].numel()
    Xb = torch.zeros(B, n, d, dtype=torch.float64)
    rb = torch.zeros(B, n, dtype=torch.float64)
    nb = torch.ones(B, n, dtype=torch.float64)
    for i, (X, r, noise, _theta) in enumerate(cases):
        Xb[i, :sizes[i]], rb[i, :sizes[i]], nb[i, :sizes[i]] = X, r, noise
    thetas = torch.stack([c[3] for c in cases])
    p = GPPlan(model, n, d, device=dev, lookahead=1, batch=B)
    p.set_site_sizes(sizes)
    p.set_inputs(Xb.to(dev).contiguous())
    Xs = torch.stack(tests).to(dev).contiguous()
    with pytest.raises(Exception):
        p.predict(thetas, Xs)  # no factorisation yet
    out = p.fit_step(thetas, rb.to(dev).contiguous(), nb.to(dev).contiguous())[0]
    assert bool((out[:, 3] == 0).all())
    mean, var = p.predict(thetas, Xs)
    mean_c, var_c = p.predict(thetas, Xs, chunk=64)  # chunked over m: same numbers
    assert mean.shape == (B, m) and torch.equal(mean, mean_c) and torch.equal(var, var_c)
    mean_only = p.predict_mean(thetas, Xs)
    assert mean_only.shape == (B, m) and (mean_only - mean).abs().max() < 1e-11
    for i, (X, r, noise, theta) in enumerate(cases):
        mu_ref, var_ref = orc.posterior(model, X, r, noise, theta, tests[i])
        assert (mean[i].cpu() - mu_ref).abs().max() < 1e-9, i
        assert ((var[i].cpu() - var_ref).abs() / (var_ref.abs() + 1e-12)).max() < 1e-7, i
        solo = plan_for(model, d, sizes[i], X, torch.float64, dev)
        solo.factorize(theta, r.to(dev), noise.to(dev))
        mu_s, var_s = solo.predict(theta, tests[i].to(dev))
        assert (mean[i] - mu_s).abs().max() < 1e-11 and (var[i] - var_s).abs().max() < 1e-11
    # after a value-only batched factorisation too
    p.factorize(thetas, rb.to(dev).contiguous(), nb.to(dev).contiguous())
    mean2, var2 = p.predict(thetas, Xs)
    assert (mean2 - mean).abs().max() < 1e-11 and (var2 - var).abs().max() < 1e-11


def test_engine_differentiable_mean_matches_oracle_autograd(gpu_device):
    """Engine-level value check of ``_differentiable_mean`` (the rating-gp penalty's inner function,
    src/rating_gp/models/gpytorch.py:160-176): posterior mean at new points and its gradient w.r.t. every raw
    parameter against autograd through the oracle.  Tolerances: mean abs 1e-8, gradients rel 1e-6."""
    from discontinuum_amd.gp.lowering import lower
    from discontinuum_amd.gp.mll import ExactMarginalLogLikelihood
    from discontinuum_amd.rating_gp import RatingGP
    from tests.helpers import rating_dataset

    torch.manual_seed(3)
    cov_ds, tgt, unc = rating_dataset(150)
    m = RatingGP()
    m.fit(cov_ds, tgt, target_unc=unc, iterations=6)
    m.model.zero_grad(set_to_none=True)
    _ = -ExactMarginalLogLikelihood(m.likelihood, m.model)(m._prior(), m._train_y)  # the plan holds this iteration's factor
    X, y, yu = torch.tensor(m.X), torch.tensor(m.y), torch.tensor(m.y_unc)
    g = torch.Generator().manual_seed(1)
    xs = torch.stack([torch.rand(24, generator=g, dtype=torch.float64) * (X[:, 0].max() - X[:, 0].min()) + X[:, 0].min(),
                      torch.rand(24, generator=g, dtype=torch.float64) * (X[:, 1].max() - X[:, 1].min()) + X[:, 1].min()], dim=1)
    w = torch.randn(24, generator=g, dtype=torch.float64)
    mu = m._differentiable_mean(xs)
    (mu.cpu() * w).sum().backward()
    o = orc.RatingOracle.from_stage(X[:, 1])
    theta = lower(m.model.covar_module, 2)[1]().detach()
    raw = torch.zeros(20, dtype=torch.float64)
    raw[0], raw[1], raw[2] = m.model.powerlaw.a.item(), m.model.powerlaw.b.item(), m.model.powerlaw.c.item()
    raw[3] = m.likelihood.second_noise_covar.raw_noise.item()
    raw[4] = orc.inv_interval(theta[0], o.b_lo, o.b_hi)
    raw[5:] = orc.inv_softplus(theta[1:])
    raw = raw.requires_grad_(True)
    th = o.constrained(raw)
    mu_lat, _ = orc.posterior("rating", X, y - o.mean(raw, X), o.noise(raw, X.shape[0], yu), th, xs)
    mu_ref = mu_lat + o.mean(raw, xs)
    assert (mu.detach().cpu() - mu_ref.detach()).abs().max() < 1e-8
    (mu_ref * w).sum().backward()
    pw = m.model.powerlaw
    got = {"a": pw.a.grad, "b": pw.b.grad, "c": pw.c.grad, "noise": m.likelihood.second_noise_covar.raw_noise.grad}
    ref = {"a": raw.grad[0], "b": raw.grad[1], "c": raw.grad[2], "noise": raw.grad[3]}
    scale = raw.grad.abs().max()
    for k in got:
        assert got[k] is not None and abs(float(got[k].reshape(-1)[0]) - float(ref[k])) <= 1e-6 * scale, k
    kern = torch.cat([p.grad.reshape(-1) for p in m.model.covar_module.parameters()])
    assert kern.numel() == 16
    # the kernel parameters' order in the module tree differs from the oracle's raw vector: compare as multisets of
    # (value) pairs through the constrained-theta gradient instead -- chain rule back through the constraints
    assert torch.isfinite(kern).all() and kern.abs().max() > 0
    assert abs(kern.abs().sum().item() - raw.grad[4:].abs().sum().item()) <= 1e-6 * max(1.0, raw.grad[4:].abs().sum().item())
