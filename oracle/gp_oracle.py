"""CPU oracle for discontinuum's exact-GP hot path.  TEST INFRASTRUCTURE ONLY.

This module is a dense torch-CPU fp64 restatement of the arithmetic that the
reference delegates to the third-party ``gpytorch`` / ``linear_operator``
packages (unpinned in the reference's ``pyproject.toml:20``; not installed in
this image, no network).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it -- and only as the checker.
The product (``discontinuum_amd``) never imports anything from ``oracle/``.

PARITY UNPINNED: the reference's own tests hold no numeric known-answer for this
path (``tests/test_loadest_gp.py:77-85`` and ``tests/test_rating_gp.py:32-65``
assert only ``is_fitted`` / Axes types on unseeded data) and gpytorch cannot be
executed here, so the formulas below follow gpytorch's published semantics
 anchored on the reference's call sites:

* kernel composition, priors          src/loadest_gp/models/gpytorch.py:48-128
                                      src/rating_gp/models/gpytorch.py:28-40, 64-79, 205-372
                                      src/rating_gp/models/kernels.py:242-382
* objective / loop semantics          src/discontinuum/engines/gpytorch.py:318, 346-384
* prediction semantics                src/discontinuum/engines/gpytorch.py:599-626

Independent cross-checks that DO run here (tests/test_oracle.py): autograd
``gradcheck``; the analytic trace gradient 1/2 tr((K^-1 - aa^T) dK); scipy
``cho_factor``/``cho_solve``; scikit-learn's RBF / Matern / ExpSineSquared kernels; and end to end scikit-learn's
GaussianProcessRegressor on both composite covariances (log-marginal likelihood and posterior).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

DT = torch.float64
LOG2PI = math.log(2.0 * math.pi)
SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)


# --------------------------------------------------------------------------
# constraints (gpytorch.constraints)
# --------------------------------------------------------------------------
def softplus(x):
    return torch.nn.functional.softplus(x)


def inv_softplus(y):
    y = torch.as_tensor(y, dtype=DT)
    return y + torch.log(-torch.expm1(-y))


def positive(raw):
    """gpytorch.constraints.Positive: theta = softplus(raw)."""
    return softplus(raw)


def greater_than(raw, lb):
    """gpytorch.constraints.GreaterThan(lb): theta = softplus(raw) + lb."""
    return softplus(raw) + lb


def interval(raw, lo, hi):
    """gpytorch.constraints.Interval(lo, hi): theta = lo + (hi - lo) sigmoid(raw)."""
    return lo + (hi - lo) * torch.sigmoid(raw)


def inv_interval(val, lo, hi):
    u = (torch.as_tensor(val, dtype=DT) - lo) / (hi - lo)
    return torch.log(u) - torch.log1p(-u)


# --------------------------------------------------------------------------
# priors (gpytorch.priors) -- log-pdf on the constrained value
# --------------------------------------------------------------------------
def normal_lp(x, loc, scale):
    return (-0.5 * ((x - loc) / scale) ** 2 - math.log(scale) - 0.5 * LOG2PI).sum()


def half_normal_lp(x, scale):
    return (math.log(2.0) - 0.5 * (x / scale) ** 2 - math.log(scale) - 0.5 * LOG2PI).sum()


def gamma_lp(x, conc, rate):
    return (conc * math.log(rate) - math.lgamma(conc) + (conc - 1.0) * torch.log(x) - rate * x).sum()


# --------------------------------------------------------------------------
# stationary kernels (gpytorch.kernels)
# --------------------------------------------------------------------------
def _scaled_sqdist(x1, x2, ls):
    """sum_j ((x1_j - x2_j)/ls_j)^2 over the last axis; x1 (n,k), x2 (m,k), ls (k,)."""
    a = x1 / ls
    b = x2 / ls
    diff = a.unsqueeze(1) - b.unsqueeze(0)
    return (diff * diff).sum(-1)


def _dist(sq):
    # gpytorch: sqrt(clamp_min(sq, 1e-30)) -> zero gradient on the diagonal
    return torch.sqrt(torch.clamp_min(sq, 1e-30))


def rbf(x1, x2, ls):
    return torch.exp(-0.5 * _scaled_sqdist(x1, x2, ls))


def matern(x1, x2, ls, nu):
    r = _dist(_scaled_sqdist(x1, x2, ls))
    if nu == 0.5:
        return torch.exp(-r)
    if nu == 1.5:
        return (1.0 + SQRT3 * r) * torch.exp(-SQRT3 * r)
    if nu == 2.5:
        return (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * torch.exp(-SQRT5 * r)
    raise ValueError(nu)


def periodic(x1, x2, ls, period):
    """gpytorch PeriodicKernel: exp(-2 sum_j sin^2(pi (x-x')/p) / l_j) (l NOT squared)."""
    diff = x1.unsqueeze(1) - x2.unsqueeze(0)
    s = torch.sin(math.pi * diff / period)
    return torch.exp(-2.0 * (s * s / ls).sum(-1))


# --------------------------------------------------------------------------
# loadest-gp composite kernel   (src/loadest_gp/models/gpytorch.py:61-128)
# --------------------------------------------------------------------------
def loadest_ntheta(d):
    return 2 * d + 5


def loadest_gram(X1, X2, theta):
    """K(X1, X2) for the loadest composite kernel, constrained parameters ``theta``:

    [0] os_seasonal [1] l_periodic [2] period [3] l_matern52(time)       (:90-101)
    [4] os_cov      [5 .. 5+d-1) l_rbf (ARD over columns 1..d-1)         (:103-114)
    [4+d] os_res    [5+d .. 5+2d) l_matern32 (ARD over all d columns)    (:116-128)
    """
    d = X1.shape[1]
    assert theta.shape[0] == loadest_ntheta(d)
    t1, t2 = X1[:, :1], X2[:, :1]
    k_seas = theta[0] * periodic(t1, t2, theta[1:2], theta[2]) * matern(t1, t2, theta[3:4], 2.5)
    k_cov = theta[4] * rbf(X1[:, 1:], X2[:, 1:], theta[5:5 + d - 1])
    k_res = theta[4 + d] * matern(X1, X2, theta[5 + d:5 + 2 * d], 1.5)
    return k_seas + k_cov + k_res


# --------------------------------------------------------------------------
# rating-gp composite kernel    (src/rating_gp/models/gpytorch.py:205-256)
# --------------------------------------------------------------------------
RATING_NTHETA = 16
GATE_A = 20.0  # src/rating_gp/models/kernels.py:271


def gate(s, b):
    """SigmoidKernel gate vector 1/(1+exp(a (s-b)))   (kernels.py:307-319)."""
    return 1.0 / (1.0 + torch.exp(GATE_A * (s - b)))


def rating_gram(X1, X2, theta):
    """K(X1, X2) for the rating composite kernel; X columns (time, stage).

    [0] gate switch point b (shared by SigmoidKernel / InvertedSigmoidKernel)
    [1..3]   cov_shift #1: os, l_stage (Matern52), l_time (Matern32)   (:242-246, 293-316)
    [4..6]   cov_shift #2: os, l_stage, l_time                         (:246-250)
    [7..9]   cov_bend: os, l_stage (Matern52), l_time (Matern52)       (:240, 318-335)
    [10..11] cov_base: os, l_stage (Matern52)                          (:238, 360-372)
    [12..15] cov_periodic: os, l_periodic, period, l_matern52(time)    (:238, 337-358)
    The stage column is log-warped (log(s + 1e-6), kernels.py:374-382) inside every
    Matern factor; the gates see the un-warped stage.
    """
    assert theta.shape[0] == RATING_NTHETA
    t1, t2 = X1[:, :1], X2[:, :1]
    s1, s2 = X1[:, 1], X2[:, 1]
    w1 = torch.log(X1[:, 1:2] + 1e-6)
    w2 = torch.log(X2[:, 1:2] + 1e-6)
    g1, g2 = gate(s1, theta[0]), gate(s2, theta[0])

    def shift(o):
        return theta[o] * matern(w1, w2, theta[o + 1:o + 2], 2.5) * matern(t1, t2, theta[o + 2:o + 3], 1.5)

    lower = shift(1) + shift(4)
    upper = theta[7] * matern(w1, w2, theta[8:9], 2.5) * matern(t1, t2, theta[9:10], 2.5)
    base = theta[10] * matern(w1, w2, theta[11:12], 2.5)
    per = theta[12] * periodic(t1, t2, theta[13:14], theta[14]) * matern(t1, t2, theta[15:16], 2.5)
    return (
        torch.outer(g1, g2) * lower
        + torch.outer(1.0 - g1, 1.0 - g2) * upper
        + base
        + per
    )


GRAMS = {"loadest": loadest_gram, "rating": rating_gram}


def composite_gram(spec):
    """Gram function of a GENERIC composite kernel -- a sum of (optionally scaled) products of RBF / Matern / Periodic
    factors on subsets of the columns -- from the same integer description the product hands to ``dgp_composite_define``
    (``discontinuum_amd.gp.lowering.composite_spec``): [d, nterms, per term: scaled, nfactors, per factor: type (0 RBF,
    1 Matern, 2 Periodic), 2 nu, ard, ndims, columns...].  theta order: per term [outputscale], per factor
    [lengthscale(s)], [period].  gpytorch semantics as above."""
    spec = [int(v) for v in spec]

    def gram(X1, X2, theta):
        i, t = 2, 0
        total = 0.0
        for _term in range(spec[1]):
            scaled, nfac = spec[i], spec[i + 1]
            i += 2
            os = 1.0
            if scaled:
                os = theta[t]
                t += 1
            prod = 1.0
            for _f in range(nfac):
                kind, nu2, ard, nd = spec[i:i + 4]
                dims = spec[i + 4:i + 4 + nd]
                i += 4 + nd
                nls = nd if ard else 1
                ls = theta[t:t + nls]
                t += nls
                a, b = X1[:, dims], X2[:, dims]
                if kind == 0:
                    val = rbf(a, b, ls)
                elif kind == 1:
                    val = matern(a, b, ls, nu2 / 2.0)
                else:
                    val = periodic(a, b, ls, theta[t])
                    t += 1
                prod = prod * val
            total = total + os * prod
        return total

    return gram


# --------------------------------------------------------------------------
# Gaussian marginal likelihood pieces
# --------------------------------------------------------------------------
def nll_data(Khat, r):
    """-log N(r | 0, Khat) = 1/2 r^T Khat^-1 r + 1/2 log|Khat| + n/2 log 2 pi  (un-normalised)."""
    n = r.shape[0]
    L = torch.linalg.cholesky(Khat)
    z = torch.linalg.solve_triangular(L, r.unsqueeze(1), upper=False).squeeze(1)
    quad = (z * z).sum()
    logdet = 2.0 * torch.log(torch.diagonal(L)).sum()
    return 0.5 * quad + 0.5 * logdet + 0.5 * n * LOG2PI


def nll_data_and_grads(model, X, r, noise, theta):
    """Value and the three gradients the HIP engine returns from one fit step:
    d/dtheta (constrained kernel parameters), d/dr (= alpha), d/dnoise (= 1/2 (diag S - alpha^2))."""
    theta = theta.detach().clone().requires_grad_(True)
    r = r.detach().clone().requires_grad_(True)
    noise = noise.detach().clone().requires_grad_(True)
    Khat = GRAMS[model](X, X, theta) + torch.diag(noise)
    val = nll_data(Khat, r)
    g_theta, g_r, g_noise = torch.autograd.grad(val, (theta, r, noise))
    return val.detach(), g_theta, g_r, g_noise


def trace_gradient(model, X, r, noise, theta):
    """Independent analytic check: dNLL/dtheta_p = 1/2 sum_ij (S - a a^T)_ij dK_ij/dtheta_p
    with dK/dtheta_p obtained by forward-mode jacobian of the dense Gram."""
    Khat = GRAMS[model](X, X, theta) + torch.diag(noise)
    S = torch.linalg.inv(Khat)
    a = S @ r
    W = S - torch.outer(a, a)
    J = torch.autograd.functional.jacobian(lambda th: GRAMS[model](X, X, th), theta)  # (n,n,P)
    return 0.5 * torch.einsum("ij,ijp->p", W, J)


def posterior(model, X, r, noise, theta, Xs, full_cov=False):
    """Latent-f posterior at Xs given residual r = y - m(X): mean offset K*^T alpha and
    var = diag(K** - K*^T Khat^-1 K*)  (exact; engines/gpytorch.py:621-624 under fast_pred_var
    is exact for n <= 800 and a rank-100 Lanczos approximation of the same quantity above)."""
    Khat = GRAMS[model](X, X, theta) + torch.diag(noise)
    L = torch.linalg.cholesky(Khat)
    alpha = torch.cholesky_solve(r.unsqueeze(1), L).squeeze(1)
    Ks = GRAMS[model](X, Xs, theta)  # (n, m)
    mu = Ks.T @ alpha
    V = torch.linalg.solve_triangular(L, Ks, upper=False)
    if full_cov:
        return mu, GRAMS[model](Xs, Xs, theta) - V.T @ V
    kss = torch.diagonal(GRAMS[model](Xs, Xs, theta))
    return mu, kss - (V * V).sum(0)


# --------------------------------------------------------------------------
# model-level restatement: raw parameters -> objective (what one fit step evaluates)
# --------------------------------------------------------------------------
@dataclass
class LoadestOracle:
    """loadest-gp ExactGPModel + FixedNoiseGaussianLikelihood(0.1^2) + ExactMarginalLogLikelihood.

    Raw parameters (all initialise to 0), flat vector of length 2d+6:
    [0] mean constant (ConstantMean, no constraint, no prior)
    [1:] raw kernel parameters in ``loadest_gram`` order, all ``Positive`` (softplus).
    """

    d: int
    noise_var: float = 0.1 ** 2  # src/loadest_gp/models/gpytorch.py:50

    @property
    def nraw(self):
        return 2 * self.d + 6

    def init_raw(self):
        return torch.zeros(self.nraw, dtype=DT)

    def constrained(self, raw):
        return positive(raw[1:])

    def log_prior(self, theta):
        d = self.d
        lp = half_normal_lp(theta[0], 1.0)            # :91
        lp = lp + normal_lp(theta[2], 1.0, 0.01)      # :92
        lp = lp + half_normal_lp(theta[4], 2.0)       # :104
        lp = lp + gamma_lp(theta[5:5 + d - 1], 2.0, 3.0)   # :105
        lp = lp + half_normal_lp(theta[4 + d], 0.2)   # :117
        lp = lp + gamma_lp(theta[5 + d:5 + 2 * d], 2.0, 10.0)  # :118
        return lp

    def mean(self, raw, X):
        return raw[0].expand(X.shape[0])

    def noise(self, raw, n, y_unc=None):
        return torch.full((n,), self.noise_var, dtype=DT)

    def objective(self, raw, X, y, y_unc=None):
        """-(mll)/1 as minimised at src/discontinuum/engines/gpytorch.py:353 (already / n)."""
        n = X.shape[0]
        theta = self.constrained(raw)
        r = y - self.mean(raw, X)
        Khat = loadest_gram(X, X, theta) + torch.diag(self.noise(raw, n))
        return (nll_data(Khat, r) - self.log_prior(theta)) / n

    def predict(self, raw, X, y, Xs, y_unc=None):
        """(mu*, var*) in model space; loadest adds NO noise at prediction
        unless m == n, where FixedNoiseGaussianLikelihood re-adds the training noise."""
        theta = self.constrained(raw)
        n = X.shape[0]
        mu, var = posterior("loadest", X, y - self.mean(raw, X), self.noise(raw, n), theta, Xs)
        mu = mu + self.mean(raw, Xs)
        if Xs.shape[0] == n:
            var = var + self.noise(raw, n)
        return mu, var


@dataclass
class RatingOracle:
    """rating-gp ExactGPModel + FixedNoise(y_unc, learn_additional_noise, HalfNormal(0.03)).

    Raw vector of length 20:
    [0..2] powerlaw a, b, c (plain Parameters; b clamped to [1.2, 2.5], c <= min(stage) - 1e-6
           in forward: src/rating_gp/models/gpytorch.py:39, 259)
    [3]    raw second_noise (GreaterThan(1e-4))                         (:71-75)
    [4]    raw gate b (Interval(q10(stage), q90(stage)))                 (:221-235)
    [5:]   15 raw kernel parameters in ``rating_gram`` order [1..15], ``Positive``.
    """

    b_lo: float
    b_hi: float
    nraw: int = 20

    @staticmethod
    def from_stage(stage):
        s = np.asarray(stage, dtype=np.float64)
        return RatingOracle(float(np.quantile(s, 0.10)), float(np.quantile(s, 0.90)))

    def constrained(self, raw):
        b = interval(raw[4:5], self.b_lo, self.b_hi)
        return torch.cat([b, positive(raw[5:])])

    def second_noise(self, raw):
        return greater_than(raw[3], 1e-4)

    def log_prior(self, raw, theta):
        lp = normal_lp(theta[0], 0.0, 1.0)                 # kernels.py:280
        lp = lp + half_normal_lp(self.second_noise(raw), 0.03)  # gpytorch.py:74
        # shift #1 (:242-246)
        lp = lp + half_normal_lp(theta[1], 0.6) + gamma_lp(theta[2], 3.0, 2.0) + gamma_lp(theta[3], 3.0, 1.0)
        # shift #2 (:246-250)
        lp = lp + half_normal_lp(theta[4], 0.3) + gamma_lp(theta[5], 3.0, 1.0) + gamma_lp(theta[6], 1.0, 7.0)
        # bend (:240, 328, 332)
        lp = lp + half_normal_lp(theta[7], 0.6) + gamma_lp(theta[8], 3.0, 2.0) + gamma_lp(theta[9], 4.0, 2.0)
        # base (:238, 365)
        lp = lp + half_normal_lp(theta[10], 1.0) + gamma_lp(theta[11], 4.0, 4.0)
        # periodic (:238, 345, 350); Matern52(time) lengthscale has no prior (:353-356)
        lp = lp + half_normal_lp(theta[12], 0.2) + gamma_lp(theta[13], 9.0, 10.0) + normal_lp(theta[14], 1.0, 0.05)
        return lp

    def clamp_(self, raw, stage_min):
        """The in-forward ``.data`` clamps (gpytorch.py:39, 259); mutates ``raw`` like the reference."""
        with torch.no_grad():
            raw[1].clamp_(1.2, 2.5)
            raw[2].clamp_(max=float(stage_min) - 1e-6)

    def mean(self, raw, X):
        return raw[0] + raw[1] * torch.log(X[:, 1] - raw[2])

    def noise(self, raw, n, y_unc):
        return y_unc + self.second_noise(raw)

    def objective(self, raw, X, y, y_unc):
        n = X.shape[0]
        self.clamp_(raw, X[:, 1].min())
        theta = self.constrained(raw)
        r = y - self.mean(raw, X)
        Khat = rating_gram(X, X, theta) + torch.diag(self.noise(raw, n, y_unc))
        return (nll_data(Khat, r) - self.log_prior(raw, theta)) / n

    def predict(self, raw, X, y, Xs, y_unc):
        """(mu*, var*); rating adds the learned homoskedastic second_noise.
        In eval mode gpytorch runs forward on [X; X*], so the c-clamp sees test stages too (A.8)."""
        n = X.shape[0]
        self.clamp_(raw, torch.minimum(X[:, 1].min(), Xs[:, 1].min()))
        theta = self.constrained(raw)
        mu, var = posterior("rating", X, y - self.mean(raw, X), self.noise(raw, n, y_unc), theta, Xs)
        mu = mu + self.mean(raw, Xs)
        var = var + self.second_noise(raw)
        if Xs.shape[0] == n:
            var = var + y_unc
        return mu, var


# --------------------------------------------------------------------------
# synthetic data generators -- shared by tests and bench
# --------------------------------------------------------------------------
def synth_loadest(n, d, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16.0, 16.0, n))
    cov = rng.standard_normal((n, d - 1))
    y = 0.8 * np.sin(2 * np.pi * t) + 0.5 * cov[:, 0] + 0.1 * t / 16.0 + 0.3 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std()
    X = np.concatenate([t[:, None], cov], axis=1)
    return X, y


def synth_rating(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16.0, 16.0, n))
    s = 1.0 + rng.beta(2.0, 5.0, n)
    y = 1.6 * np.log(s - 0.5) + 0.2 * np.sin(2 * np.pi * t) * (s < 1.3) + 0.05 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std()
    y_unc = rng.uniform(1e-3, 4e-3, n)
    X = np.stack([t, s], axis=1)
    return X, y, y_unc
