"""GPU parity of every stage of the HIP hot path against the CPU oracle (same seeded inputs).

All calls go through the C ABI (libdgp_hip.so via ctypes).  Tolerances (fp64 path):
  Gram entries abs 1e-13; Cholesky / inverse factors rel 1e-9 (Frobenius); NLL rel 1e-10;
  hyperparameter gradients rel 1e-8 (of the gradient's max-norm); alpha / dnoise rel 1e-8;
  posterior mean abs 1e-9, variance rel 1e-8 (+1e-12 abs).
fp32 path: NLL rel 1e-4 * max(1, n/1024), gradients rel 1e-2, mean/var abs 1e-3.
"""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

CASES = [
    ("loadest", 2, 200),
    ("loadest", 3, 384),
    ("loadest", 3, 1000),
    ("loadest", 4, 300),
    ("rating", 2, 200),
    ("rating", 2, 900),
]


def make_case(model, d, n, seed=0, perturb=0.0):
    g = torch.Generator().manual_seed(seed)
    if model == "loadest":
        X, y = orc.synth_loadest(n, d, seed)
        m = orc.LoadestOracle(d)
        raw = m.init_raw() + perturb * torch.randn(m.nraw, generator=g, dtype=orc.DT)
        X, y = torch.tensor(X), torch.tensor(y)
        noise = m.noise(raw, n)
        r = y - m.mean(raw, X)
    else:
        X, y, yu = orc.synth_rating(n, seed)
        X, y, yu = torch.tensor(X), torch.tensor(y), torch.tensor(yu)
        m = orc.RatingOracle.from_stage(X[:, 1])
        raw = torch.zeros(20, dtype=orc.DT)
        raw[1], raw[2] = 1.6, 0.5
        raw[3] = -5.0
        raw = raw + perturb * torch.randn(20, generator=g, dtype=orc.DT)
        m.clamp_(raw, X[:, 1].min())
        noise = m.noise(raw, n, yu)
        r = y - m.mean(raw, X)
    theta = m.constrained(raw)
    return X, r.detach(), noise.detach(), theta.detach()


def plan_for(model, d, n, X, dtype, dev, lookahead=True):
    from discontinuum_amd.backend import GPPlan

    p = GPPlan(model, n, d, dtype=dtype, device=dev, lookahead=lookahead)
    p.set_inputs(X.to(dev, dtype).contiguous())
    return p


def tril_n(t, n):
    return torch.tril(t[:n, :n]).cpu().double()


@pytest.mark.parametrize("model,d,n", CASES)
def test_stages_fp64(model, d, n, gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=1, perturb=0.3)
    Khat = orc.GRAMS[model](X, X, theta) + torch.diag(noise)
    p = plan_for(model, d, n, X, torch.float64, dev)
    # -- Gram
    p.stage_gram(theta, noise.to(dev))
    A = p.buffer(_lib.BUF_A)
    assert (tril_n(A, n) - torch.tril(Khat)).abs().max() < 1e-13
    N = p.N
    if N > n:  # identity pad
        pad = A[n:, :].cpu()
        assert torch.equal(torch.tril(pad[:, n:]), torch.eye(N - n, dtype=torch.float64))
        assert pad[:, :n].abs().max() == 0
    # -- Cholesky
    p.stage_potrf()
    L_ref = torch.linalg.cholesky(Khat)
    L = tril_n(p.buffer(_lib.BUF_A), n)
    assert torch.linalg.norm(L - L_ref) / torch.linalg.norm(L_ref) < 1e-9
    # -- L^-1
    p.stage_trtri()
    T = tril_n(p.buffer(_lib.BUF_T), n)
    eye = torch.eye(n, dtype=torch.float64)
    assert torch.linalg.norm(T @ L_ref - eye) / np.sqrt(n) < 1e-9
    # -- K^-1
    p.stage_lauum()
    S = tril_n(p.buffer(_lib.BUF_S), n)
    S_full = S + S.T - torch.diag(torch.diagonal(S))
    assert torch.linalg.norm(S_full @ Khat - eye) / np.sqrt(n) < 1e-7
    # -- solves
    p.stage_solve(r.to(dev))
    alpha_ref = torch.cholesky_solve(r[:, None], L_ref)[:, 0]
    alpha = p.buffer(_lib.BUF_ALPHA)[:n].cpu()
    assert (alpha - alpha_ref).abs().max() / alpha_ref.abs().max() < 1e-8
    if N > n:
        assert p.buffer(_lib.BUF_ALPHA)[n:].abs().max().item() == 0
    # -- gradient contraction
    g = p.stage_grad(theta).cpu()
    _, g_ref, _, _ = orc.nll_data_and_grads(model, X, r, noise, theta)
    assert (g - g_ref).abs().max() / g_ref.abs().max() < 1e-8


@pytest.mark.parametrize("model,d,n", CASES)
@pytest.mark.parametrize("lookahead", [True, False])
def test_fit_step_fp64(model, d, n, lookahead, gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=2, perturb=0.2)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    p = plan_for(model, d, n, X, torch.float64, dev, lookahead=lookahead)
    out, dr, dnoise = p.fit_step(theta, r.to(dev), noise.to(dev))
    out = out.cpu()
    assert out[_lib.OUT_INFO] == 0
    assert abs(out[_lib.OUT_NLL] - val) / abs(val) < 1e-10
    P = theta.numel()
    assert (out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P] - g_theta).abs().max() / g_theta.abs().max() < 1e-8
    assert (dr.cpu() - g_r).abs().max() / g_r.abs().max() < 1e-8
    assert (dnoise.cpu() - g_noise).abs().max() / g_noise.abs().max() < 1e-8
    # the result row also carries sum_i dNLL/dr_i (the gradient of a constant prior mean, negated), sum dnoise, and
    # -- on request -- dr against two weight vectors (a parametric prior mean's Jacobian)
    assert abs(out[_lib.OUT_SUM_DR] - g_r.sum()) <= 1e-8 * g_r.abs().sum()
    assert abs(out[_lib.OUT_SUM_DNOISE] - g_noise.sum()) <= 1e-8 * g_noise.abs().sum()
    assert out[_lib.OUT_DR_W0] == 0 and out[_lib.OUT_DR_W0 + 1] == 0
    w = torch.randn(2, n, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    p.set_dr_weights(w.to(dev).contiguous())
    outw = p.fit_step(theta, r.to(dev), noise.to(dev))[0].cpu()
    for kk in range(2):
        assert abs(outw[_lib.OUT_DR_W0 + kk] - (g_r * w[kk]).sum()) <= 1e-8 * (g_r * w[kk]).abs().sum()
    p.set_dr_weights(None)
    # repeated call is bitwise reproducible (deterministic reductions)
    out2, dr2, _ = p.fit_step(theta, r.to(dev), noise.to(dev))
    assert torch.equal(out2.cpu(), out) and torch.equal(dr2, dr)


@pytest.mark.parametrize("model,d,n", [("loadest", 3, 700), ("rating", 2, 500)])
def test_fit_step_fp32(model, d, n, gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=3, perturb=0.1)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    p = plan_for(model, d, n, X, torch.float32, dev)
    out, dr, dnoise = p.fit_step(theta, r.to(dev, torch.float32), noise.to(dev, torch.float32))
    out = out.cpu().double()
    assert out[_lib.OUT_INFO] == 0
    assert abs(out[_lib.OUT_NLL] - val) / abs(val) < 1e-4 * max(1.0, n / 1024)
    P = theta.numel()
    assert (out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P] - g_theta).abs().max() / g_theta.abs().max() < 1e-2


@pytest.mark.parametrize("model,d,n,m", [("loadest", 3, 500, 333), ("rating", 2, 400, 130), ("loadest", 2, 300, 300)])
def test_predict_fp64(model, d, n, m, gpu_device):
    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=4, perturb=0.2)
    Xs, *_ = make_case(model, d, m, seed=5)
    mu_ref, var_ref = orc.posterior(model, X, r, noise, theta, Xs)
    p = plan_for(model, d, n, X, torch.float64, dev)
    p.factorize(theta, r.to(dev), noise.to(dev))
    mu, var = p.predict(theta, Xs.to(dev), chunk=200)
    assert (mu.cpu() - mu_ref).abs().max() < 1e-9
    assert ((var.cpu() - var_ref).abs() / (var_ref.abs() + 1e-4)).max() < 1e-8
    Ks = p.cross_gram(theta, Xs.to(dev)).cpu()
    assert (Ks - orc.GRAMS[model](X, Xs, theta)).abs().max() < 1e-13


def test_not_positive_definite_reports_info(gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    n, d = 300, 2
    X, r, noise, theta = make_case("loadest", d, n, seed=6)
    X[150] = X[149]  # duplicate point ...
    noise = torch.full((n,), -0.5, dtype=torch.float64)  # ... and a negative "noise": K^ is indefinite
    p = plan_for("loadest", d, n, X, torch.float64, dev)
    out, _, _ = p.fit_step(theta, r.to(dev), noise.to(dev))
    out = out.cpu()
    assert out[_lib.OUT_INFO] >= 1
    assert not torch.isfinite(out[_lib.OUT_NLL])


@pytest.mark.parametrize("n,level,B", [(2500, 2, 1), (2500, 1, 1), (700, 1, 3)])
def test_not_positive_definite_every_schedule_terminates(n, level, B, gpu_device):
    """An indefinite matrix must come back as info >= 1 / non-finite NLL from every schedule -- the early inverse's
    queue-driven launches and the batched plan included -- and a healthy site in the same batch must be unaffected."""
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, d = gpu_device, 2
    X, r, noise, theta = make_case("loadest", d, n, seed=6)
    bad_noise = torch.full((n,), -0.5, dtype=torch.float64)
    p = GPPlan("loadest", n, d, device=dev, lookahead=level, batch=B)
    if B == 1:
        p.set_inputs(X.to(dev).contiguous())
        out = p.fit_step(theta, r.to(dev), bad_noise.to(dev))[0].cpu()
        assert out[_lib.OUT_INFO] >= 1 and not torch.isfinite(out[_lib.OUT_NLL])
        out = p.fit_step(theta, r.to(dev), noise.to(dev))[0].cpu()  # and the plan recovers on the next call
        assert out[_lib.OUT_INFO] == 0 and torch.isfinite(out[_lib.OUT_NLL])
    else:
        p.set_inputs(X.to(dev).repeat(B, 1, 1).contiguous())
        nz = torch.stack([noise, bad_noise, noise]).to(dev).contiguous()
        out = p.fit_step(theta.repeat(B, 1), r.to(dev).repeat(B, 1).contiguous(), nz)[0].cpu()
        assert out[1, _lib.OUT_INFO] >= 1 and not torch.isfinite(out[1, _lib.OUT_NLL])
        assert out[0, _lib.OUT_INFO] == 0 and out[2, _lib.OUT_INFO] == 0
        assert torch.isfinite(out[0, _lib.OUT_NLL]) and out[0, _lib.OUT_NLL] == out[2, _lib.OUT_NLL]


def test_bad_arguments_fail_loudly(gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    with pytest.raises(ValueError):
        GPPlan("rating", 100, 3, device=gpu_device)
    p = GPPlan("loadest", 100, 2, device=gpu_device)
    r = torch.zeros(100, dtype=torch.float64, device=gpu_device)
    with pytest.raises(_lib.DGPError):  # no inputs yet
        p.fit_step(torch.ones(9), r, r)
    with pytest.raises(ValueError):
        p.fit_step(torch.ones(5), r, r)


@pytest.mark.parametrize("model,d,n,m", [("rating", 2, 300, 40), ("loadest", 3, 200, 128), ("rating", 2, 100, 200)])
def test_predictive_mean_and_vjp_fp64(model, d, n, m, gpu_device):
    """dgp_predict_mean / dgp_mean_vjp (the differentiable mean the monotonicity penalty needs) vs autograd
    through the oracle's dense posterior mean.  Tolerance: mean abs 1e-9, gradients rel 1e-7."""
    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=11, perturb=0.2)
    Xs, *_ = make_case(model, d, m, seed=12)
    w = torch.randn(m, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    th, rr, nn = (t.clone().requires_grad_(True) for t in (theta, r, noise))
    mu_ref, _ = orc.posterior(model, X, rr, nn, th, Xs)
    g_theta, g_r, g_noise = torch.autograd.grad((mu_ref * w).sum(), (th, rr, nn))
    p = plan_for(model, d, n, X, torch.float64, dev)
    with pytest.raises(Exception):  # no factorisation yet
        p.predict_mean(theta, Xs.to(dev))
    p.fit_step(theta, r.to(dev), noise.to(dev))
    mu = p.predict_mean(theta, Xs.to(dev))
    assert (mu.cpu() - mu_ref.detach()).abs().max() < 1e-9
    dtheta, dr, dnoise = p.mean_vjp(theta, Xs.to(dev), w.to(dev))
    assert (dtheta.cpu() - g_theta).abs().max() / g_theta.abs().max() < 1e-7
    assert (dr.cpu() - g_r).abs().max() / g_r.abs().max() < 1e-7
    assert (dnoise.cpu() - g_noise).abs().max() / g_noise.abs().max() < 1e-7


@pytest.mark.parametrize("model,d,sizes,m", [("rating", 2, [300, 170, 300], 40), ("loadest", 3, [200, 129, 64, 200], 130),
                                             ("rating", 2, [150] * 12, 70)])  # 12 sites: hyperparameters through the device scratch
def test_batched_inference_is_one_launch_sequence_and_matches_the_oracle(model, d, sizes, m, gpu_device):
    """dgp_predict / dgp_predict_mean / dgp_mean_vjp on a (ragged) batched plan -- gridDim.z = sites, no host loop over
    the sites -- against the oracle's posterior and autograd through its mean, site by site.  Tolerances as for single
    plans: mean 1e-9, variance 1e-8, VJP gradients rel 1e-7; rows beyond a site's own size come back as zeros."""
    from discontinuum_amd.backend import GPPlan

    dev, B, n = gpu_device, len(sizes), max(sizes)
    cases = [make_case(model, d, nb, seed=30 + b, perturb=0.2) for b, nb in enumerate(sizes)]
    X = torch.full((B, n, d), float("nan"), dtype=torch.float64)
    r = torch.full((B, n), float("nan"), dtype=torch.float64)
    noise = torch.full((B, n), float("nan"), dtype=torch.float64)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        X[b, :nb], r[b, :nb], noise[b, :nb] = c[0], c[1], c[2]
    theta = torch.stack([c[3] for c in cases])
    Xs = torch.stack([make_case(model, d, m, seed=50 + b)[0] for b in range(B)])
    w = torch.randn(B, m, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    pb = GPPlan(model, n, d, device=dev, lookahead=1, batch=B)
    pb.set_site_sizes(sizes)
    pb.set_inputs(X.to(dev).contiguous())
    pb.fit_step(theta, r.to(dev).contiguous(), noise.to(dev).contiguous())
    mean, var = pb.predict(theta, Xs.to(dev))
    mu = pb.predict_mean(theta, Xs.to(dev))
    dtheta, dr, dnoise = pb.mean_vjp(theta, Xs.to(dev), w.to(dev).contiguous())
    assert mean.shape == (B, m) and dtheta.shape == (B, pb.ntheta) and dr.shape == (B, n)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        th, rr, nn = (t.clone().requires_grad_(True) for t in (c[3], c[1], c[2]))
        mu_ref, var_ref = orc.posterior(model, c[0], rr, nn, th, Xs[b])
        g_theta, g_r, g_noise = torch.autograd.grad((mu_ref * w[b]).sum(), (th, rr, nn))
        assert (mean[b].cpu() - mu_ref.detach()).abs().max() < 1e-9
        assert (mu[b].cpu() - mu_ref.detach()).abs().max() < 1e-9
        assert (var[b].cpu() - var_ref.detach()).abs().max() <= 1e-8 * max(1.0, var_ref.abs().max().item())
        assert (dtheta[b].cpu() - g_theta).abs().max() / g_theta.abs().max() < 1e-7
        assert (dr[b, :nb].cpu() - g_r).abs().max() / g_r.abs().max() < 1e-7
        assert (dnoise[b, :nb].cpu() - g_noise).abs().max() / g_noise.abs().max() < 1e-7
        assert bool((dr[b, nb:] == 0).all()) and bool((dnoise[b, nb:] == 0).all())


@pytest.mark.parametrize("n", [384, 512, 640, 768, 896, 1024, 1152, 1300])
def test_lookahead_schedule_matches_plain_schedule(n, gpu_device):
    """Every block-column count (odd / even, with and without a final unpaired panel) through the paired
    lookahead schedule gives the plain right-looking result (different summation order: rel 1e-12)."""
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case("loadest", 3, n, seed=n, perturb=0.1)
    outs = []
    for la in (True, False):
        p = plan_for("loadest", 3, n, X, torch.float64, dev, lookahead=la)
        out, dr, dn = p.fit_step(theta, r.to(dev), noise.to(dev))
        outs.append((out.cpu(), dr.cpu(), dn.cpu()))
    (o1, a1, n1), (o2, a2, n2) = outs
    assert o1[_lib.OUT_INFO] == 0 and o2[_lib.OUT_INFO] == 0
    assert abs(o1[0] - o2[0]) / abs(o2[0]) < 1e-12
    P = theta.numel()
    assert (o1[4:4 + P] - o2[4:4 + P]).abs().max() / o2[4:4 + P].abs().max() < 1e-9
    assert (a1 - a2).abs().max() / a2.abs().max() < 1e-9
    val, g_theta, _, _ = orc.nll_data_and_grads("loadest", X, r, noise, theta)
    assert abs(o1[0] - val) / abs(val) < 1e-10
    assert (o1[4:4 + P] - g_theta).abs().max() / g_theta.abs().max() < 1e-8


@pytest.mark.parametrize("n", [2048, 2200, 2500, 3000, 3300, 5000])
def test_early_inverse_matches_in_order_schedule(n, gpu_device):
    """From 16 block columns on, level 2 issues the inverse's level recursion piecewise behind checkpoints of the
    factorisation (third stream).  Same kernels, same operands: the inverse factor and K^^-1 must agree with the
    in-order schedules bit for bit in T (identical launches) and the step results to rounding."""
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case("loadest", 3, n, seed=n, perturb=0.1)
    res = {}
    for level in (2, 1, 0):
        p = plan_for("loadest", 3, n, X, torch.float64, dev, lookahead=level)
        out, dr, dn = p.fit_step(theta, r.to(dev), noise.to(dev))
        torch.cuda.synchronize()
        res[level] = (out.cpu(), dr.cpu(), tril_n(p.buffer(_lib.BUF_T), n), tril_n(p.buffer(_lib.BUF_S), n))
    o2, a2, T2, S2 = res[2]
    assert o2[_lib.OUT_INFO] == 0
    P = theta.numel()
    for level in (1, 0):
        o, a, Tm, S = res[level]
        assert abs(o2[0] - o[0]) / abs(o[0]) < 1e-12
        assert (o2[4:4 + P] - o[4:4 + P]).abs().max() / o[4:4 + P].abs().max() < 1e-9
        assert (a2 - a).abs().max() / a.abs().max() < 1e-9
        assert (T2 - Tm).abs().max() <= 1e-9 * Tm.abs().max()
        assert (S2 - S).abs().max() <= 1e-9 * S.abs().max()
    assert torch.equal(T2, res[1][2])  # levels 2 and 1 share the factorisation schedule: identical inverse
    # and the factor really is the inverse: T K^ T^T = I on a probe
    Khat = orc.GRAMS["loadest"](X, X, theta) + torch.diag(noise)
    probe = torch.randn(n, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    assert (S2 @ (Khat @ probe) + torch.tril(S2, -1).T @ (Khat @ probe) - probe).abs().max() < 1e-6


@pytest.mark.parametrize("model,d,n,split_tiles", [("loadest", 3, 500, None), ("loadest", 3, 512, None), ("rating", 2, 640, None),
                                                   ("loadest", 3, 1000, None), ("loadest", 2, 2048, None), ("rating", 2, 3300, None),
                                                   # hand-over from the pair schedule to the split chain after 2, 6, 12 pairs
                                                   ("loadest", 3, 2500, 120), ("loadest", 3, 3300, 60), ("rating", 2, 3072, 21),
                                                   # hand-over from GROUPS OF FOUR panels (what plans of 96+ block columns start in)
                                                   ("loadest", 3, 3300, -60), ("loadest", 3, 12288, None)])
def test_split_panel_chain_is_bitwise_the_single_stream_chain(model, d, n, split_tiles, gpu_device, monkeypatch):
    """The split panel chain (critical tile on the caller's stream, rest of the chain on a second stream, in-kernel progress
    counters; csrc/dgp_chol.hip::potrf_split) applies every panel to every element in the same order and continues the same
    k-ordered fma chains as the single-stream schedule: L, L^-1 and the whole result row are BITWISE equal in fp64, also
    when the factorisation starts in the pair schedule and hands over (DGP_SPLIT_TILES moves the switch point), at every
    lookahead level, and run to run."""
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=n, perturb=0.1)
    if split_tiles is not None:
        monkeypatch.setenv("DGP_SPLIT_TILES", str(abs(split_tiles)))
        if split_tiles < 0:
            monkeypatch.setenv("DGP_GROUP", "4")
    res = {}
    for mode, level in (("0", 1), ("1", 1), ("1", 2), ("1", 1)):
        monkeypatch.setenv("DGP_SPLIT_CHAIN", mode)
        p = plan_for(model, d, n, X, torch.float64, dev, lookahead=level)
        out, dr, dn = p.fit_step(theta, r.to(dev), noise.to(dev))
        torch.cuda.synchronize()
        got = (out.cpu(), tril_n(p.buffer(_lib.BUF_A), n).clone(), tril_n(p.buffer(_lib.BUF_T), n).clone())
        if (mode, level) in res:  # the repeated configuration: bitwise run to run
            assert all(torch.equal(a, b) for a, b in zip(got, res[(mode, level)]))
        res[(mode, level)] = got
        del p
    ref = res[("0", 1)]
    assert ref[0][_lib.OUT_INFO] == 0
    for key in (("1", 1), ("1", 2)):
        out, L, Tm = res[key]
        assert torch.equal(L, ref[1]), key
        assert torch.equal(Tm, ref[2]), key
        assert torch.equal(out[:4], ref[0][:4]), key  # NLL, quad, log-det, info
    val, g_theta, _, _ = orc.nll_data_and_grads(model, X, r, noise, theta)
    assert abs(ref[0][0] - val) <= 1e-10 * abs(val)


def test_split_panel_chain_fp32_and_not_positive_definite(gpu_device, monkeypatch):
    """fp32 through the split chain (its trailing updates sum from zero, so schedules differ by rounding, not bitwise):
    both schedules within SURVEY's fp32 bound of the fp64 oracle; and a non-PD matrix is reported, not hung on."""
    from discontinuum_amd import _lib

    dev, n = gpu_device, 3000
    X, r, noise, theta = make_case("loadest", 3, n, seed=5, perturb=0.1)
    val = orc.nll_data_and_grads("loadest", X, r, noise, theta)[0]
    for mode in ("0", "1"):
        monkeypatch.setenv("DGP_SPLIT_CHAIN", mode)
        p = plan_for("loadest", 3, n, X, torch.float32, dev)
        out = p.fit_step(theta, r.float().to(dev), noise.float().to(dev))[0].cpu().double()
        assert out[_lib.OUT_INFO] == 0 and abs(out[0] - val) <= 1e-4 * (n / 1024) * abs(val), mode
        del p
    monkeypatch.setenv("DGP_SPLIT_CHAIN", "1")
    Xb = X.clone()
    Xb[1700] = Xb[300]
    bad = noise.clone()
    bad[300] = bad[1700] = -0.5
    p = plan_for("loadest", 3, n, Xb, torch.float64, dev)
    out = p.fit_step(theta, r.to(dev), bad.to(dev))[0].cpu()
    assert out[_lib.OUT_INFO] >= 1 and not torch.isfinite(out[_lib.OUT_NLL])
    out = p.fit_step(theta, r.to(dev), noise.to(dev))[0].cpu()  # the next step on the same plan recovers
    assert out[_lib.OUT_INFO] == 0 and torch.isfinite(out[_lib.OUT_NLL])


@pytest.mark.parametrize("model,d,n", [("loadest", 2, 1), ("loadest", 2, 2), ("loadest", 3, 17), ("loadest", 2, 127),
                                       ("loadest", 3, 128), ("loadest", 2, 129), ("rating", 2, 3), ("rating", 2, 257)])
def test_edge_sizes_fp64(model, d, n, gpu_device):
    """Sizes at and around the 128-wide padding quantum, down to a single observation."""
    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=11, perturb=0.1)
    r = torch.nan_to_num(r, nan=0.3)  # the generator standardises y: undefined for a single observation
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    p = plan_for(model, d, n, X, torch.float64, dev)
    out, alpha, dnoise = p.fit_step(theta, r.to(dev), noise.to(dev))
    out = out.cpu()
    assert int(out[3]) == 0
    assert abs(out[0].item() - val.item()) <= 1e-10 * max(1.0, abs(val.item()))
    gt = out[4:4 + p.ntheta]
    assert (gt - g_theta).abs().max() <= 1e-8 * max(1.0, g_theta.abs().max().item())
    assert (alpha.cpu() - g_r).abs().max() <= 1e-8 * max(1.0, g_r.abs().max().item())
    assert (dnoise.cpu() - g_noise).abs().max() <= 1e-8 * max(1.0, g_noise.abs().max().item())
    # a single prediction point, and more prediction points than observations
    for m in (1, 2 * n + 3):
        Xs, *_ = make_case(model, d, m, seed=12)
        mu_ref, var_ref = orc.posterior(model, X, r, noise, theta, Xs)
        mu, var = p.predict(theta, Xs.to(dev))
        assert (mu.cpu() - mu_ref).abs().max() < 1e-9
        assert ((var.cpu() - var_ref).abs() / (var_ref.abs() + 1e-4)).max() < 1e-8


def test_repeated_observations_fp64(gpu_device):
    """Identical input rows (collisions): K is singular, K + Sigma is not -- the factorisation must go through."""
    dev = gpu_device
    X, r, noise, theta = make_case("loadest", 3, 300, seed=2, perturb=0.1)
    X = torch.cat([X[:150], X[:150]])  # every row twice
    val, g_theta, g_r, _ = orc.nll_data_and_grads("loadest", X, r, noise, theta)
    p = plan_for("loadest", 3, 300, X, torch.float64, dev)
    out, alpha, _ = p.fit_step(theta, r.to(dev), noise.to(dev))
    out = out.cpu()
    assert int(out[3]) == 0
    assert abs(out[0].item() - val.item()) <= 1e-9 * abs(val.item())
    assert (out[4:4 + p.ntheta] - g_theta).abs().max() <= 1e-7 * max(1.0, g_theta.abs().max().item())
    assert (alpha.cpu() - g_r).abs().max() <= 1e-7 * max(1.0, g_r.abs().max().item())


def test_nan_input_is_reported_not_hidden(gpu_device):
    """A NaN in the residual or the inputs must surface in the outputs (NaN NLL or non-zero info), never a finite lie."""
    dev = gpu_device
    X, r, noise, theta = make_case("loadest", 2, 200, seed=3)
    p = plan_for("loadest", 2, 200, X, torch.float64, dev)
    r_bad = r.clone()
    r_bad[7] = float("nan")
    out, *_ = p.fit_step(theta, r_bad.to(dev), noise.to(dev))
    assert not np.isfinite(out.cpu()[0].item())
    X_bad = X.clone()
    X_bad[5, 0] = float("nan")
    p.set_inputs(X_bad.to(dev).contiguous())
    out, *_ = p.fit_step(theta, r.to(dev), noise.to(dev))
    out = out.cpu()
    assert (not np.isfinite(out[0].item())) or int(out[3]) != 0
    # ... in a covariate column (it only enters the squared-distance terms) and in rating-gp's stage column
    X_bad = X.clone()
    X_bad[11, 1] = float("nan")
    p.set_inputs(X_bad.to(dev).contiguous())
    out = p.fit_step(theta, r.to(dev), noise.to(dev))[0].cpu()
    assert (not np.isfinite(out[0].item())) or int(out[3]) != 0
    Xr, rr, nr, tr = make_case("rating", 2, 150, seed=3)
    Xr[40, 1] = float("nan")
    pr = plan_for("rating", 2, 150, Xr, torch.float64, dev)
    out = pr.fit_step(tr, rr.to(dev), nr.to(dev))[0].cpu()
    assert (not np.isfinite(out[0].item())) or int(out[3]) != 0


@pytest.mark.parametrize("model,d,n,B", [("loadest", 3, 700, 3), ("rating", 2, 520, 2), ("loadest", 2, 2300, 4),
                                         ("loadest", 3, 300, 12), ("rating", 2, 200, 20)])  # > 8: hyperparameters via device memory
def test_batched_plan_matches_single_site_plans(model, d, n, B, gpu_device):
    """A batched plan (B sites in lockstep, one launch per kernel) must give every site what a plan of its own
    gives: different inputs, hyperparameters, residuals and noise per site; both lookahead levels."""
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    cases = [make_case(model, d, n, seed=20 + b, perturb=0.15) for b in range(B)]
    X = torch.stack([c[0] for c in cases]).to(dev).contiguous()
    r = torch.stack([c[1] for c in cases]).to(dev).contiguous()
    noise = torch.stack([c[2] for c in cases]).to(dev).contiguous()
    theta = torch.stack([c[3] for c in cases])
    for level in (1, 0):
        pb = GPPlan(model, n, d, dtype=torch.float64, device=dev, lookahead=level, batch=B)
        pb.set_inputs(X)
        out, dr, dn = pb.fit_step(theta, r, noise)
        assert out.shape == (B, _lib.OUT_LEN) and dr.shape == (B, n) and dn.shape == (B, n)
        for b in range(B):
            p1 = plan_for(model, d, n, cases[b][0], torch.float64, dev, lookahead=level)
            o1, a1, n1 = p1.fit_step(cases[b][3], r[b].contiguous(), noise[b].contiguous())
            assert int(out[b, _lib.OUT_INFO]) == 0
            assert abs(out[b, 0] - o1[0]) <= 1e-12 * abs(o1[0])
            P = p1.ntheta
            assert (out[b, 4:4 + P] - o1[4:4 + P]).abs().max() <= 1e-10 * o1[4:4 + P].abs().max()
            assert (dr[b] - a1).abs().max() <= 1e-10 * a1.abs().max()
            assert (dn[b] - n1).abs().max() <= 1e-10 * n1.abs().max()
    # against the oracle for one site
    val, g_theta, g_r, _ = orc.nll_data_and_grads(model, *[cases[B - 1][i] for i in (0, 1, 2, 3)])
    assert abs(out[B - 1, 0].cpu() - val) <= 1e-10 * abs(val)
    assert (out[B - 1, 4:4 + g_theta.numel()].cpu() - g_theta).abs().max() <= 1e-8 * max(1.0, g_theta.abs().max().item())
    # a batched plan predicts every site at its own points (values against the oracle's posterior) ...
    Xs = torch.stack([make_case(model, d, 5, seed=70 + b)[0] for b in range(B)])
    mean, var = pb.predict(theta, Xs.to(dev))
    assert mean.shape == (B, 5) and var.shape == (B, 5)
    for b in range(B):
        mu_ref, var_ref = orc.posterior(model, cases[b][0], cases[b][1], cases[b][2], cases[b][3], Xs[b])
        assert (mean[b].cpu() - mu_ref).abs().max() <= 1e-9
        assert (var[b].cpu() - var_ref).abs().max() <= 1e-8 * max(1.0, var_ref.abs().max().item())
    # ... and refuses arguments of a single site's shape
    with pytest.raises(ValueError):
        pb.predict(theta[0], cases[0][0][:5].to(dev))


def test_batched_plan_fp32_matches_single_site_plans(gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, model, d, n, B = gpu_device, "loadest", 3, 900, 5
    cases = [make_case(model, d, n, seed=60 + b, perturb=0.1) for b in range(B)]
    X = torch.stack([c[0] for c in cases]).float().to(dev).contiguous()
    r = torch.stack([c[1] for c in cases]).float().to(dev).contiguous()
    noise = torch.stack([c[2] for c in cases]).float().to(dev).contiguous()
    theta = torch.stack([c[3] for c in cases])
    pb = GPPlan(model, n, d, dtype=torch.float32, device=dev, lookahead=1, batch=B)
    pb.set_inputs(X)
    out, dr, dn = pb.fit_step(theta, r, noise)
    for b in (0, B - 1):
        p1 = GPPlan(model, n, d, dtype=torch.float32, device=dev, lookahead=1)
        p1.set_inputs(X[b].contiguous())
        o1, a1, n1 = p1.fit_step(theta[b], r[b].contiguous(), noise[b].contiguous())
        assert int(out[b, _lib.OUT_INFO]) == 0
        # same kernels, same operand order: identical up to the bulk update's tile shapes (fp32 rounding)
        assert abs(out[b, 0] - o1[0]) <= 1e-5 * abs(o1[0])
        assert (dr[b] - a1).abs().max() <= 1e-3 * a1.abs().max()


@pytest.mark.parametrize("model,d,sizes", [("loadest", 3, [700, 513, 300, 129]), ("rating", 2, [400, 400, 77]),
                                           ("loadest", 2, [2300, 1000, 2299])])
def test_ragged_batch_matches_plans_of_each_size(model, d, sizes, gpu_device):
    """Sites with different numbers of observations in one batched plan: site b uses the first sizes[b] rows of its
    slots (the unused tails are filled with NaN here to prove they are never read) and must match a plan of size
    sizes[b]."""
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, B, n = gpu_device, len(sizes), max(sizes)
    cases = [make_case(model, d, nb, seed=80 + b, perturb=0.1) for b, nb in enumerate(sizes)]
    X = torch.full((B, n, d), float("nan"), dtype=torch.float64)
    r = torch.full((B, n), float("nan"), dtype=torch.float64)
    noise = torch.full((B, n), float("nan"), dtype=torch.float64)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        X[b, :nb], r[b, :nb], noise[b, :nb] = c[0], c[1], c[2]
    theta = torch.stack([c[3] for c in cases])
    pb = GPPlan(model, n, d, device=dev, lookahead=1, batch=B)
    pb.set_site_sizes(sizes)
    pb.set_inputs(X.to(dev).contiguous())
    out, dr, dn = pb.fit_step(theta, r.to(dev).contiguous(), noise.to(dev).contiguous())
    out, dr, dn = out.cpu(), dr.cpu(), dn.cpu()
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        p1 = plan_for(model, d, nb, c[0], torch.float64, dev, lookahead=1)
        o1, a1, n1 = [t.cpu() for t in p1.fit_step(c[3], c[1].to(dev), c[2].to(dev))]
        assert int(out[b, _lib.OUT_INFO]) == 0
        assert abs(out[b, 0] - o1[0]) <= 1e-11 * abs(o1[0])
        P = p1.ntheta
        assert (out[b, 4:4 + P] - o1[4:4 + P]).abs().max() <= 1e-9 * max(1.0, o1[4:4 + P].abs().max().item())
        assert (dr[b, :nb] - a1).abs().max() <= 1e-9 * a1.abs().max()
        assert (dn[b, :nb] - n1).abs().max() <= 1e-9 * n1.abs().max()
        assert bool((dr[b, nb:] == 0).all()) and bool((dn[b, nb:] == 0).all())
    with pytest.raises(Exception):
        pb.set_site_sizes([n + 1] + sizes[1:])
