"""Test doubles and data builders shared by the CPU and GPU suites (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import numpy as np
import torch

from discontinuum_amd import _lib
from discontinuum_amd.xr_compat import DataArray, Dataset
from oracle import gp_oracle as orc


class OraclePlan:
    """CPU stand-in for ``backend.GPPlan`` that answers from the oracle.  It exists so the host logic
    (fit loop, lowering, priors, constraints, checkpointing) can be exercised without a GPU; the product
    never constructs it (``MarginalHIP._plan_factory`` is ``GPPlan``)."""

    def __init__(self, model, n, d, dtype=torch.float64, device="cpu", lookahead=True, batch=1):
        self.model, self.n, self.d, self.dtype, self.device = model, n, d, dtype, torch.device("cpu")
        self.ntheta = orc.loadest_ntheta(d) if model == "loadest" else orc.RATING_NTHETA
        self.calls = 0
        self.batch = int(batch)
        # batch > 1 (``GPPlan(batch=B)``'s surface for ``fit_many``): one single-site double per site, ragged sizes
        self._sites = [OraclePlan(model, n, d, dtype) for _ in range(self.batch)] if self.batch > 1 else None
        self._sizes = [n] * self.batch

    def set_site_sizes(self, sizes):
        self._sizes = [int(v) for v in sizes]

    def set_inputs(self, X):
        if self._sites:
            for b, p in enumerate(self._sites):
                p.n = self._sizes[b]
                p.set_inputs(X[b, : self._sizes[b]])
            return
        self.X = X.double()

    def set_dr_weights(self, w):
        if self._sites:
            for b, p in enumerate(self._sites):
                p.set_dr_weights(None if w is None else w[b][:, : self._sizes[b]])
            return
        self._dr_w = None if w is None else w.double()

    def fit_step(self, theta, r, noise):
        self.calls += 1
        if self._sites:
            rows = [p.fit_step(theta[b], r[b, : self._sizes[b]], noise[b, : self._sizes[b]]) for b, p in enumerate(self._sites)]
            pad = lambda v: torch.cat([v, torch.zeros(self.n - v.shape[0], dtype=v.dtype)])  # noqa: E731
            return (torch.stack([o for o, _, _ in rows]), torch.stack([pad(a) for _, a, _ in rows]),
                    torch.stack([pad(g) for _, _, g in rows]))
        theta = torch.as_tensor(theta, dtype=torch.float64).detach()
        out = torch.zeros(_lib.OUT_LEN, dtype=self.dtype)
        try:
            with torch.enable_grad():  # autograd.Function.forward runs under no_grad
                val, g_theta, g_r, g_noise = orc.nll_data_and_grads(
                    self.model, self.X, r.detach().double(), noise.detach().double(), theta)
        except torch.linalg.LinAlgError:  # not positive definite
            out[_lib.OUT_NLL] = float("nan")
            out[_lib.OUT_INFO] = 1
            return out, torch.zeros(self.n, dtype=self.dtype), torch.zeros(self.n, dtype=self.dtype)
        out[_lib.OUT_NLL] = val
        out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + self.ntheta] = g_theta
        out[_lib.OUT_SUM_DR] = g_r.sum()
        out[_lib.OUT_SUM_DNOISE] = g_noise.sum()
        if getattr(self, "_dr_w", None) is not None:
            out[_lib.OUT_DR_W0] = (g_r * self._dr_w[0]).sum()
            out[_lib.OUT_DR_W0 + 1] = (g_r * self._dr_w[1]).sum()
        self._state = (theta, r.double().detach(), noise.double().detach())
        return out, g_r.to(self.dtype), g_noise.to(self.dtype)

    def factorize(self, theta, r, noise):
        theta = torch.as_tensor(theta, dtype=torch.float64).detach()
        self._state = (theta, r.double().detach(), noise.double().detach())
        out = torch.zeros(_lib.OUT_LEN, dtype=self.dtype)
        Khat = orc.GRAMS[self.model](self.X, self.X, theta) + torch.diag(noise.double())
        out[_lib.OUT_NLL] = orc.nll_data(Khat, r.double())
        return out

    def predict(self, theta, Xs, chunk=4096):
        theta, r, noise = self._state
        mu, var = orc.posterior(self.model, self.X, r, noise, theta, Xs.double())
        return mu.to(self.dtype), var.to(self.dtype)

    def predict_mean(self, theta, Xs):
        theta, r, noise = self._state
        mu, _ = orc.posterior(self.model, self.X, r, noise, theta, Xs.double())
        return mu.to(self.dtype)

    def mean_vjp(self, theta, Xs, w):
        th, r, noise = (t.clone().requires_grad_(True) for t in self._state)
        with torch.enable_grad():
            mu, _ = orc.posterior(self.model, self.X, r, noise, th, Xs.double())
            g = torch.autograd.grad((mu * w.double()).sum(), (th, r, noise))
        return tuple(t.to(self.dtype) for t in g)

    def posterior_factor(self, theta, Xs):
        theta, r, noise = self._state
        mu, cov = orc.posterior(self.model, self.X, r, noise, theta, Xs.double(), full_cov=True)
        eye = torch.eye(cov.shape[0], dtype=torch.float64)
        for jitter in (0.0, 1e-8, 1e-7, 1e-6):
            L, info = torch.linalg.cholesky_ex(cov + jitter * eye)
            if int(info) == 0:
                return mu.to(self.dtype), L.to(self.dtype), jitter
        raise RuntimeError("posterior covariance not positive definite")

    def sample_draws(self, L, m, mean, ndraw, generator=None):
        z = torch.randn(m, ndraw, dtype=self.dtype, generator=generator)
        return (mean[:, None] + L @ z).T.contiguous()


def loadest_dataset(n=40, seed=0):
    """Synthetic sampling record shaped like the reference fixtures (tests/test_loadest_gp.py:12-28)."""
    rng = np.random.default_rng(seed)
    time = np.sort(rng.choice(np.arange("2010-01-01", "2016-01-01", dtype="datetime64[D]"), n, replace=False)).astype("datetime64[ns]")
    flow = np.exp(rng.standard_normal(n)) * 10
    conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
    covariates = Dataset({"flow": ("time", flow)}, coords={"time": time})
    target = Dataset({"concentration": ("time", conc)}, coords={"time": time})
    return covariates, target["concentration"]


def rating_dataset(n=40, seed=0):
    rng = np.random.default_rng(seed)
    time = np.sort(rng.choice(np.arange("2010-01-01", "2016-01-01", dtype="datetime64[D]"), n, replace=False)).astype("datetime64[ns]")
    stage = 1.0 + 3.0 * rng.beta(2, 5, n)
    q = np.exp(1.6 * np.log(stage) + 0.05 * rng.standard_normal(n))
    unc = np.full(n, 1.05)  # geometric standard error of the measurement
    covariates = Dataset({"stage": ("time", stage)}, coords={"time": time})
    target = DataArray(q, dims=("time",), coords={"time": time}, name="discharge", attrs={"units": "cfs"})
    target_unc = DataArray(unc, dims=("time",), coords={"time": time}, name="discharge_unc")
    return covariates, target, target_unc
