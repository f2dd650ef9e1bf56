"""Seeded random sweep of the fit step and the prediction against the oracle: sizes that are not multiples of anything,
both kernels, every input dimension of loadest-gp, all three lookahead levels, single and ragged batched plans.
Tolerances are those of the fixed-size tests: NLL rel 1e-10, gradients rel 1e-8."""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc
from tests.test_gpu_stages import make_case

pytestmark = pytest.mark.gpu


def _oracle(model, X, r, noise, theta):
    with torch.enable_grad():
        return orc.nll_data_and_grads(model, X, r, noise, theta)


def _check(tag, model, X, r, noise, theta, out, dr, dn, P):
    from discontinuum_amd import _lib

    val, g_theta, g_r, g_noise = _oracle(model, X, r, noise, theta)
    assert int(out[_lib.OUT_INFO]) == 0, tag
    assert abs(float(out[_lib.OUT_NLL]) - float(val)) <= 1e-10 * max(1.0, abs(float(val))), tag
    got = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P]
    assert (got - g_theta).abs().max() <= 1e-8 * max(1.0, float(g_theta.abs().max())), tag
    assert (dr - g_r).abs().max() <= 1e-8 * max(1.0, float(g_r.abs().max())), tag
    assert (dn - g_noise).abs().max() <= 1e-8 * max(1.0, float(g_noise.abs().max())), tag


def test_random_single_site_plans_match_oracle(gpu_device):
    from discontinuum_amd.backend import GPPlan

    rng = np.random.default_rng(2024)
    for trial in range(24):
        model = "rating" if rng.random() < 0.4 else "loadest"
        d = 2 if model == "rating" else int(rng.integers(2, 7))
        n = int(rng.choice([rng.integers(1, 130), rng.integers(130, 700), rng.integers(700, 1500)]))
        level = int(rng.integers(0, 3))
        X, r, noise, theta = make_case(model, d, n, seed=300 + trial, perturb=0.15)
        p = GPPlan(model, n, d, device=gpu_device, lookahead=level)
        p.set_inputs(X.to(gpu_device).contiguous())
        out, dr, dn = [t.cpu() for t in p.fit_step(theta, r.to(gpu_device), noise.to(gpu_device))]
        _check(f"trial {trial}: {model} d={d} n={n} level={level}", model, X, r, noise, theta, out, dr, dn, p.ntheta)
        m = int(rng.integers(1, 300))
        Xs = X[rng.integers(0, n, m)] + 0.01 * torch.randn(m, d, dtype=torch.float64)
        if model == "rating":
            Xs[:, 1] = Xs[:, 1].clamp(min=float(X[:, 1].min()))
        p.factorize(theta, r.to(gpu_device), noise.to(gpu_device))
        mu, var = [t.cpu() for t in p.predict(theta, Xs.to(gpu_device).contiguous())]
        mu0, var0 = orc.posterior(model, X, r, noise, theta, Xs)
        assert (mu - mu0).abs().max() <= 1e-9 * max(1.0, float(mu0.abs().max())), (trial, "mean")
        assert (var - var0).abs().max() <= 1e-8 * max(1.0, float(var0.abs().max())), (trial, "variance")


def test_random_ragged_batches_match_oracle(gpu_device):
    from discontinuum_amd.backend import GPPlan

    rng = np.random.default_rng(7)
    for trial in range(6):
        model = "rating" if trial % 3 == 2 else "loadest"
        d = 2 if model == "rating" else int(rng.integers(2, 5))
        B = int(rng.integers(2, 12))
        sizes = [int(rng.integers(1, 900)) for _ in range(B)]
        n = max(sizes)
        cases = [make_case(model, d, nb, seed=500 + 20 * trial + b, perturb=0.1) for b, nb in enumerate(sizes)]
        X = torch.full((B, n, d), float("nan"), dtype=torch.float64)
        r = torch.full((B, n), float("nan"), dtype=torch.float64)
        noise = torch.full((B, n), float("nan"), dtype=torch.float64)
        for b, (nb, c) in enumerate(zip(sizes, cases)):
            X[b, :nb], r[b, :nb], noise[b, :nb] = c[0], c[1], c[2]
        theta = torch.stack([c[3] for c in cases])
        pb = GPPlan(model, n, d, device=gpu_device, lookahead=int(rng.integers(0, 2)), batch=B)
        pb.set_site_sizes(sizes)
        pb.set_inputs(X.to(gpu_device).contiguous())
        out, dr, dn = [t.cpu() for t in pb.fit_step(theta, r.to(gpu_device).contiguous(), noise.to(gpu_device).contiguous())]
        for b, (nb, c) in enumerate(zip(sizes, cases)):
            _check(f"trial {trial} site {b}: {model} d={d} n={nb} of {n}", model, c[0], c[1], c[2], c[3], out[b], dr[b, :nb],
                   dn[b, :nb], pb.ntheta)
