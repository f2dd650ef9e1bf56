"""Pin the CPU oracle: against the committed golden vectors, and against independent restatements that
do not share its code (autograd gradcheck, the analytic trace gradient, scipy's Cholesky solver,
scikit-learn's kernels).  PARITY UNPINNED at the gpytorch boundary -- see oracle/gp_oracle.py."""
import glob
import math
import os

import numpy as np
import pytest
import scipy.linalg
import torch
from sklearn.gaussian_process import kernels as sk

from oracle import gp_oracle as orc

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def test_golden_fixtures_exist():
    assert len(GOLD) == 18


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    model = os.path.basename(path).split("_")[0]
    X, y, raw = torch.tensor(z["X"]), torch.tensor(z["y"]), torch.tensor(z["raw"]).requires_grad_(True)
    if model == "loadest":
        m = orc.LoadestOracle(X.shape[1])
        obj = m.objective(raw, X, y)
    else:
        m = orc.RatingOracle.from_stage(X[:, 1])
        obj = m.objective(raw, X, y, torch.tensor(z["y_unc"]))
    (g,) = torch.autograd.grad(obj, raw)
    assert abs(obj.item() - float(z["objective"])) <= 1e-12 * max(1.0, abs(float(z["objective"])))
    assert np.allclose(g.numpy(), z["grad_raw"], rtol=1e-9, atol=1e-12)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, torch.tensor(z["r"]), torch.tensor(z["noise"]), torch.tensor(z["theta"]))
    assert abs(val.item() - float(z["nll_data"])) <= 1e-12 * abs(float(z["nll_data"]))
    assert np.allclose(g_theta.numpy(), z["grad_theta"], rtol=1e-9, atol=1e-11)
    assert np.allclose(g_r.numpy(), z["alpha"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("model,d", [("loadest", 2), ("loadest", 3), ("rating", 2)])
def test_trace_gradient_matches_autograd(model, d):
    n = 40
    if model == "loadest":
        X, y = orc.synth_loadest(n, d, 3)
        theta = orc.positive(0.4 * torch.randn(orc.loadest_ntheta(d), dtype=orc.DT, generator=torch.Generator().manual_seed(1)))
    else:
        X, y, _ = orc.synth_rating(n, 3)
        theta = orc.positive(0.4 * torch.randn(16, dtype=orc.DT, generator=torch.Generator().manual_seed(2)))
        theta[0] = 1.25
    X, y = torch.tensor(X), torch.tensor(y)
    noise = torch.full((n,), 0.02, dtype=orc.DT)
    _, g_auto, g_r, g_noise = orc.nll_data_and_grads(model, X, y, noise, theta)
    g_trace = orc.trace_gradient(model, X, y, noise, theta)
    assert torch.allclose(g_auto, g_trace, rtol=1e-8, atol=1e-10)
    Khat = orc.GRAMS[model](X, X, theta) + torch.diag(noise)
    S = torch.linalg.inv(Khat)
    alpha = S @ y
    assert torch.allclose(g_r, alpha, rtol=1e-8, atol=1e-10)
    assert torch.allclose(g_noise, 0.5 * (torch.diagonal(S) - alpha ** 2), rtol=1e-7, atol=1e-9)


def test_gradcheck_loadest_objective():
    X, y = orc.synth_loadest(12, 2, 5)
    X, y = torch.tensor(X), torch.tensor(y)
    m = orc.LoadestOracle(2)
    raw = (0.3 * torch.randn(m.nraw, dtype=orc.DT, generator=torch.Generator().manual_seed(0))).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda r: m.objective(r, X, y), (raw,), eps=1e-6, atol=1e-6, rtol=1e-5)


def test_nll_matches_scipy_cholesky():
    X, y = orc.synth_loadest(50, 3, 9)
    X, y = torch.tensor(X), torch.tensor(y)
    theta = torch.full((11,), 0.6931471805599453, dtype=orc.DT)
    K = (orc.loadest_gram(X, X, theta) + 0.01 * torch.eye(50, dtype=orc.DT)).numpy()
    c, low = scipy.linalg.cho_factor(K, lower=True)
    a = scipy.linalg.cho_solve((c, low), y.numpy())
    ref = 0.5 * y.numpy() @ a + np.log(np.diag(c)).sum() + 0.5 * 50 * math.log(2 * math.pi)
    assert abs(orc.nll_data(torch.tensor(K), y).item() - ref) < 1e-10 * abs(ref)


def test_kernels_match_sklearn():
    """Independent check of the closed forms (note sklearn's ExpSineSquared uses l^2 where gpytorch's
    PeriodicKernel uses l: SURVEY Appendix A.2)."""
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((7, 2)), rng.standard_normal((5, 2))
    ls = np.array([0.7, 1.9])
    A, B, L = torch.tensor(a), torch.tensor(b), torch.tensor(ls)
    assert np.allclose(orc.rbf(A, B, L).numpy(), sk.RBF(length_scale=ls)(a, b), atol=1e-14)
    assert np.allclose(orc.matern(A, B, L, 1.5).numpy(), sk.Matern(length_scale=ls, nu=1.5)(a, b), atol=1e-12)
    assert np.allclose(orc.matern(A, B, L, 2.5).numpy(), sk.Matern(length_scale=ls, nu=2.5)(a, b), atol=1e-12)
    t1, t2 = a[:, :1], b[:, :1]
    ell, p = 0.8, 1.3
    ours = orc.periodic(torch.tensor(t1), torch.tensor(t2), torch.tensor([ell]), torch.tensor(p)).numpy()
    assert np.allclose(ours, sk.ExpSineSquared(length_scale=math.sqrt(ell), periodicity=p)(t1, t2), atol=1e-13)


def test_constraints_and_priors_closed_forms():
    raw = torch.tensor([-1.3, 0.0, 2.1], dtype=orc.DT)
    assert torch.allclose(orc.inv_softplus(orc.positive(raw)), raw)
    assert torch.allclose(orc.inv_interval(orc.interval(raw, 1.1, 1.7), 1.1, 1.7), raw)
    assert abs(orc.positive(torch.zeros(1, dtype=orc.DT)).item() - math.log(2)) < 1e-15  # gpytorch default 0.6931
    x = torch.tensor([0.4, 1.5], dtype=orc.DT)
    from scipy import stats

    assert abs(orc.normal_lp(x, 1.0, 0.3).item() - stats.norm(1.0, 0.3).logpdf(x.numpy()).sum()) < 1e-12
    assert abs(orc.half_normal_lp(x, 0.7).item() - stats.halfnorm(scale=0.7).logpdf(x.numpy()).sum()) < 1e-12
    assert abs(orc.gamma_lp(x, 2.0, 3.0).item() - stats.gamma(a=2.0, scale=1 / 3.0).logpdf(x.numpy()).sum()) < 1e-12


def test_rating_posterior_noise_semantics():
    """loadest adds no noise at prediction unless m == n; rating adds the learned second_noise."""
    X, y, yu = orc.synth_rating(30, 1)
    X, y, yu = torch.tensor(X), torch.tensor(y), torch.tensor(yu)
    m = orc.RatingOracle.from_stage(X[:, 1])
    raw = torch.zeros(20, dtype=orc.DT)
    raw[1], raw[2], raw[3] = 1.6, 0.5, -3.0
    mu1, v1 = m.predict(raw.clone(), X, y, X[:10].clone(), yu)
    theta = m.constrained(raw)
    _, v_lat = orc.posterior("rating", X, y - m.mean(raw, X), m.noise(raw, 30, yu), theta, X[:10])
    assert torch.allclose(v1, v_lat + m.second_noise(raw))
