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


def test_loadest_nll_and_posterior_match_scikit_learn_gpr():
    """END-TO-END anchor on an independent GP implementation: scikit-learn's GaussianProcessRegressor with the loadest-gp
    covariance composed from scikit-learn's own kernels (ConstantKernel * ExpSineSquared * Matern52 on time + Constant *
    ARD-RBF on the covariates + Constant * ARD-Matern32 on all columns; the only glue is a column selector).  Its
    log-marginal likelihood is -NLL_data of the oracle and its latent predictive mean / standard deviation are the
    oracle's posterior -- kernel composition, A.6's likelihood and the prediction formulas all at once (gpytorch itself
    cannot run here: the gpytorch boundary stays parity unpinned)."""
    from sklearn.gaussian_process import GaussianProcessRegressor

    class Cols(sk.Kernel):
        """``kernel`` on the columns ``cols`` of the inputs (scikit-learn kernels have no active_dims)."""

        def __init__(self, kernel, cols):
            self.kernel, self.cols = kernel, cols

        def __call__(self, X, Y=None, eval_gradient=False):
            assert not eval_gradient
            return self.kernel(X[:, self.cols], None if Y is None else Y[:, self.cols])

        def diag(self, X):
            return self.kernel.diag(X[:, self.cols])

        def is_stationary(self):
            return True

    d, n, m = 3, 60, 25
    X, y = orc.synth_loadest(n, d, 11)
    Xs, _ = orc.synth_loadest(m, d, 12)
    theta = orc.positive(0.5 * torch.randn(orc.loadest_ntheta(d), dtype=orc.DT, generator=torch.Generator().manual_seed(4)))
    th = theta.numpy()
    os1, lp, per, lm, os2, l2a, l2b, os3, l3a, l3b, l3c = th
    fixed = "fixed"
    kernel = (sk.ConstantKernel(os1, fixed) * Cols(sk.ExpSineSquared(math.sqrt(lp), per, fixed, fixed), [0])
              * Cols(sk.Matern(lm, fixed, nu=2.5), [0])
              + sk.ConstantKernel(os2, fixed) * Cols(sk.RBF([l2a, l2b], fixed), [1, 2])
              + sk.ConstantKernel(os3, fixed) * sk.Matern([l3a, l3b, l3c], fixed, nu=1.5))
    noise = np.full(n, 0.01)
    gpr = GaussianProcessRegressor(kernel=kernel, alpha=noise, optimizer=None, normalize_y=False).fit(X, y)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    Khat = orc.loadest_gram(Xt, Xt, theta) + torch.diag(torch.tensor(noise))
    assert np.allclose(Khat.numpy(), kernel(X) + np.diag(noise), atol=1e-13)
    nll = orc.nll_data(Khat, yt).item()
    assert abs(-gpr.log_marginal_likelihood_value_ - nll) < 1e-10 * abs(nll)
    mu, var = orc.posterior("loadest", Xt, yt, torch.tensor(noise), theta, torch.tensor(Xs))
    mean_sk, std_sk = gpr.predict(Xs, return_std=True)
    assert np.allclose(mu.numpy(), mean_sk, atol=1e-10)
    assert np.allclose(var.numpy(), std_sk ** 2, atol=1e-10)


def test_rating_nll_and_posterior_match_scikit_learn_gpr():
    """The rating-gp covariance on scikit-learn's GaussianProcessRegressor: the Matern / periodic factors, their products and
    sums are scikit-learn's; the glue restates only the reference's own custom pieces -- the rank-one sigmoid gates with
    a = 20 and a shared switch point (src/rating_gp/models/kernels.py:242-360) and the log-warp of the stage column
    (:363-382).  Parameter order: DESIGN.md section 2."""
    from sklearn.gaussian_process import GaussianProcessRegressor

    class Wrap(sk.Kernel):
        def is_stationary(self):
            return False

        def diag(self, X):
            return np.diag(self(X))

    class Cols(Wrap):
        def __init__(self, kernel, cols):
            self.kernel, self.cols = kernel, cols

        def __call__(self, X, Y=None, eval_gradient=False):
            return self.kernel(X[:, self.cols], None if Y is None else Y[:, self.cols])

    class LogWarp(Wrap):
        def __init__(self, kernel):
            self.kernel = kernel

        def __call__(self, X, Y=None, eval_gradient=False):
            def warp(Z):
                Z = Z.copy()
                Z[:, 1] = np.log(Z[:, 1] + 1e-6)
                return Z
            return self.kernel(warp(X), None if Y is None else warp(Y))

    class Gate(Wrap):
        def __init__(self, b, inverted):
            self.b, self.inverted = b, inverted

        def __call__(self, X, Y=None, eval_gradient=False):
            def g(Z):
                v = 1.0 / (1.0 + np.exp(20.0 * (Z[:, 1] - self.b)))
                return 1.0 - v if self.inverted else v
            return np.outer(g(X), g(X if Y is None else Y))

    n, m = 70, 30
    X, y, yu = orc.synth_rating(n, 21)
    Xs, _, _ = orc.synth_rating(m, 22)
    theta = orc.positive(0.4 * torch.randn(16, dtype=orc.DT, generator=torch.Generator().manual_seed(6)))
    theta[0] = float(np.median(X[:, 1]))
    t = theta.numpy()
    f = "fixed"

    def m52(ls, col):
        return Cols(sk.Matern(ls, f, nu=2.5), [col])

    def shift(os, ls_s, ls_t):
        return sk.ConstantKernel(os, f) * m52(ls_s, 1) * Cols(sk.Matern(ls_t, f, nu=1.5), [0])

    lower = shift(*t[1:4]) + shift(*t[4:7])
    bend = sk.ConstantKernel(t[7], f) * m52(t[8], 1) * m52(t[9], 0)
    rest = (sk.ConstantKernel(t[10], f) * m52(t[11], 1)
            + sk.ConstantKernel(t[12], f) * Cols(sk.ExpSineSquared(math.sqrt(t[13]), t[14], f, f), [0]) * m52(t[15], 0))
    kernel = Gate(t[0], False) * LogWarp(lower) + Gate(t[0], True) * LogWarp(bend) + LogWarp(rest)
    noise = yu + 0.02
    Xt, yt = torch.tensor(X), torch.tensor(y)
    Khat = orc.rating_gram(Xt, Xt, theta) + torch.diag(torch.tensor(noise))
    assert np.allclose(Khat.numpy(), kernel(X) + np.diag(noise), atol=1e-13)
    gpr = GaussianProcessRegressor(kernel=kernel, alpha=noise, optimizer=None, normalize_y=False).fit(X, y)
    nll = orc.nll_data(Khat, yt).item()
    assert abs(-gpr.log_marginal_likelihood_value_ - nll) < 1e-10 * max(1.0, abs(nll))
    mu, var = orc.posterior("rating", Xt, yt, torch.tensor(noise), theta, torch.tensor(Xs))
    mean_sk, std_sk = gpr.predict(Xs, return_std=True)
    assert np.allclose(mu.numpy(), mean_sk, atol=1e-9)
    assert np.allclose(var.numpy(), std_sk ** 2, atol=1e-9)
