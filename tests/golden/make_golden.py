#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the CPU oracle (oracle/gp_oracle.py).

PARITY UNPINNED at the gpytorch boundary: the reference holds no numeric known-answer for this path and
gpytorch is not installable here, so these vectors pin the ORACLE against itself
across code changes and give the GPU tests fixed inputs/outputs that travel to the GPU box.  If a later
environment has gpytorch, run these inputs through it under `gpytorch.settings.max_cholesky_size(10**9)`
and record the deltas.

    python tests/golden/make_golden.py     # rewrites tests/golden/*.npz (small: n <= 300)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gp_oracle as orc  # noqa: E402

CASES = [("loadest", 2, 8), ("loadest", 2, 64), ("loadest", 3, 64), ("loadest", 3, 300), ("rating", 2, 64), ("rating", 2, 300)]


def build(model, d, n, point):
    """point 0: gpytorch defaults (raw = 0); 1, 2: raw ~ N(0, 0.5^2) with fixed seeds."""
    g = torch.Generator().manual_seed(100 + point)
    if model == "loadest":
        X, y = orc.synth_loadest(n, d, seed=n)
        X, y = torch.tensor(X), torch.tensor(y)
        m = orc.LoadestOracle(d)
        raw = m.init_raw()
        if point:
            raw = raw + 0.5 * torch.randn(m.nraw, generator=g, dtype=orc.DT)
        yu = None
    else:
        X, y, yu = orc.synth_rating(n, seed=n)
        X, y, yu = torch.tensor(X), torch.tensor(y), torch.tensor(yu)
        m = orc.RatingOracle.from_stage(X[:, 1])
        raw = torch.zeros(20, dtype=orc.DT)
        raw[1], raw[2], raw[3] = 1.6, 0.5, -4.0
        if point:
            raw = raw + 0.5 * torch.randn(20, generator=g, dtype=orc.DT)
        m.clamp_(raw, X[:, 1].min())
    raw = raw.clone().requires_grad_(True)
    obj = m.objective(raw, X, y, yu) if model == "rating" else m.objective(raw, X, y)
    (graw,) = torch.autograd.grad(obj, raw)
    rawd = raw.detach()
    theta = m.constrained(rawd)
    n_ = X.shape[0]
    noise = m.noise(rawd, n_, yu) if model == "rating" else m.noise(rawd, n_)
    r = y - m.mean(rawd, X)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    rng = np.random.default_rng(7 + point)
    idx = rng.choice(n_, size=min(16, n_), replace=False)
    Xs = X[idx] + 0.01
    mu, var = (m.predict(rawd.clone(), X, y, Xs, yu) if model == "rating" else m.predict(rawd, X, y, Xs))
    return dict(X=X.numpy(), y=y.numpy(), y_unc=(yu.numpy() if yu is not None else np.zeros(0)), raw=rawd.numpy(),
                theta=theta.numpy(), noise=noise.numpy(), r=r.numpy(), objective=np.array(obj.item()),
                grad_raw=graw.numpy(), nll_data=np.array(val.item()), grad_theta=g_theta.numpy(),
                alpha=g_r.numpy(), grad_noise=g_noise.numpy(), Xs=Xs.numpy(), mu=mu.numpy(), var=var.numpy())


def main():
    for model, d, n in CASES:
        for point in range(3):
            out = build(model, d, n, point)
            np.savez_compressed(os.path.join(HERE, f"{model}_d{d}_n{n}_p{point}.npz"), **out)
    print("wrote", len(CASES) * 3, "fixtures to", HERE)


if __name__ == "__main__":
    main()
