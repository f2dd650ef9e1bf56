"""float32 as a first-class path (it is the reference's dtype: src/discontinuum/engines/gpytorch.py:221-222).

The fp32 HIP path against the fp64 oracle at n = 4096 for both models (the dense CPU oracle takes seconds there), with
the tolerance tied to what fp32 can deliver on THIS matrix: the oracle measures cond(K^) and the test requires

    alpha:      ||alpha32 - alpha64|| / ||alpha64||  <=  4 cond(K^) eps32        (backward-stable solve)
    NLL:        |NLL32 - NLL64| / scale              <=  1e-4 max(1, n / 1024)    (SURVEY.md section 8d), scale = the
                sum of the magnitudes of the NLL's three terms (1/2 quad, 1/2 log-det, n/2 log 2 pi): the log-det of a
                small-noise matrix is negative and cancels against the other two, so |NLL| itself is not the
                conditioning scale.  BOTH normalisations are asserted (against |NLL|, round 3: loadest 1e-6, rating
                3e-5 .. 3e-4 over six seeds at n = 4096; before the accumulators started from zero: 4.2e-4)
    gradients:  max |g32 - g64| / max |g64|          <=  1e-2                     
    posterior:  mean abs <= 1e-3, variance abs <= 1e-3                            

and records what it measured (``gpurun_out/fp32_parity.jsonl`` when that directory exists; DESIGN.md section 5 quotes
it).  Size-independent checks at the full sizes of BASELINE configs 3 and 5 live in tests/test_gpu_fullsize.py.
"""
import json
import os

import pytest
import torch

from oracle import gp_oracle as orc
from tests.test_gpu_stages import make_case, plan_for

pytestmark = pytest.mark.gpu

EPS32 = 2.0 ** -24
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(**row):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fp32_parity.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")


@pytest.mark.parametrize("model,d,n", [("loadest", 3, 4096), ("rating", 2, 4096), ("loadest", 2, 1500), ("rating", 2, 1500)])
def test_fp32_fit_step_and_posterior_against_the_fp64_oracle(model, d, n, gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=7, perturb=0.1)
    Xs, *_ = make_case(model, d, 300, seed=8)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    Khat = orc.GRAMS[model](X, X, theta) + torch.diag(noise)
    ev = torch.linalg.eigvalsh(Khat)
    cond = (ev[-1] / ev[0]).item()
    mu_ref, var_ref = orc.posterior(model, X, r, noise, theta, Xs)
    p = plan_for(model, d, n, X, torch.float32, dev)
    out, dr, dnoise = p.fit_step(theta, r.to(dev, torch.float32), noise.to(dev, torch.float32))
    out = out.cpu().double()
    assert out[_lib.OUT_INFO] == 0
    P = theta.numel()
    Lref = torch.linalg.cholesky(Khat)
    logdet_ref = 2.0 * torch.log(torch.diagonal(Lref)).sum().item()
    quad_ref = float(r @ g_r)  # r^T K^^-1 r (dNLL/dr = alpha)
    scale = 0.5 * abs(quad_ref) + 0.5 * abs(logdet_ref) + 0.5 * n * 1.8378770664093453
    e_nll_rel = abs(out[_lib.OUT_NLL] - val).item() / abs(val).item()
    e_nll = abs(out[_lib.OUT_NLL] - val).item() / scale
    e_grad = ((out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P] - g_theta).abs().max() / g_theta.abs().max()).item()
    e_alpha = (torch.linalg.norm(dr.cpu().double() - g_r) / torch.linalg.norm(g_r)).item()
    e_dnoise = ((dnoise.cpu().double() - g_noise).abs().max() / g_noise.abs().max()).item()
    mean, var = p.predict(theta, Xs.to(dev, torch.float32))
    e_mean = (mean.cpu().double() - mu_ref).abs().max().item()
    e_var = (var.cpu().double() - var_ref).abs().max().item()
    _record(test="fp32_vs_fp64_oracle", model=model, d=d, n=n, cond=cond, nll_rel=e_nll_rel, nll_over_term_scale=e_nll, grad_rel=e_grad, alpha_rel=e_alpha,
            dnoise_rel=e_dnoise, mean_abs=e_mean, var_abs=e_var, alpha_bound=4 * cond * EPS32)
    assert e_nll <= 1e-4 * max(1.0, n / 1024), e_nll
    # SURVEY.md section 8d's bound as written, against |NLL| itself (restored in round 3: the trailing updates of fp32
    # plans sum each pass from zero, csrc/dgp_gemm.h::trailing_begin -- before that the log-determinant carried a
    # systematic +1e-5 n error and this assertion failed at rating n = 4096)
    assert e_nll_rel <= 1e-4 * max(1.0, n / 1024), e_nll_rel
    assert e_grad <= 1e-2, e_grad
    assert e_alpha <= 4 * cond * EPS32, (e_alpha, cond)
    assert e_dnoise <= 1e-2 + 4 * cond * EPS32, e_dnoise
    assert e_mean <= 1e-3 and e_var <= 1e-3, (e_mean, e_var)
    # Round 4 -- far inside those bounds: alpha is refined once against an fp64 residual (DGP_OPT_REFINE) and K^^-1 = L^-T L^-1
    # sums its products small-to-large (csrc/dgp_gemm.h REV).  Measured at n = 4096 (gpurun_out/fp32_parity.jsonl): alpha
    # 3.8e-6 (was 1.5e-3 for rating), gradients 8e-6 (6.8e-4), dnoise 4e-6 (1.4e-3), NLL 1e-5 of |NLL| (3e-4).
    assert e_alpha <= 1e-4, e_alpha
    assert e_grad <= 5e-4, e_grad
    assert e_dnoise <= 5e-4, e_dnoise
    assert e_nll_rel <= 1e-4, e_nll_rel
    # without the refinement the same plan is where round 3 left it: the step, not luck, buys the digits
    p.set_option(_lib.OPT_REFINE, 0)
    _, dr0, _ = p.fit_step(theta, r.to(dev, torch.float32), noise.to(dev, torch.float32))
    e_alpha0 = (torch.linalg.norm(dr0.cpu().double() - g_r) / torch.linalg.norm(g_r)).item()
    assert e_alpha0 <= 4 * cond * EPS32 and e_alpha < 0.2 * e_alpha0, (e_alpha, e_alpha0)


@pytest.mark.parametrize("model,d", [("loadest", 3), ("rating", 2)])
def test_fp32_batched_and_single_plans_agree_at_n4096(model, d, gpu_device):
    """The same fp32 kernels under the batched schedule (groups of four panels) and the single-site schedule (pairs,
    early inverse): NLL to 2e-5, alpha to 5e-5 of each other at n = 4096 (measured 2.2e-6 / 2.8e-6)."""
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, n, B = gpu_device, 4096, 4
    cases = [make_case(model, d, n, seed=90 + b, perturb=0.1) for b in range(B)]
    X = torch.stack([c[0] for c in cases]).float().to(dev).contiguous()
    r = torch.stack([c[1] for c in cases]).float().to(dev).contiguous()
    noise = torch.stack([c[2] for c in cases]).float().to(dev).contiguous()
    theta = torch.stack([c[3] for c in cases])
    pb = GPPlan(model, n, d, dtype=torch.float32, device=dev, lookahead=1, batch=B)
    pb.set_inputs(X)
    out, dr, _ = pb.fit_step(theta, r, noise)
    p1 = GPPlan(model, n, d, dtype=torch.float32, device=dev)
    for b in (0, B - 1):
        p1.set_inputs(X[b].contiguous())
        o1, a1, _ = p1.fit_step(theta[b], r[b].contiguous(), noise[b].contiguous())
        assert int(out[b, _lib.OUT_INFO]) == 0 and int(o1[_lib.OUT_INFO]) == 0
        e_nll = (abs(out[b, 0] - o1[0]) / abs(o1[0])).item()
        e_alpha = (torch.linalg.norm((dr[b] - a1).double()) / torch.linalg.norm(a1.double())).item()
        _record(test="fp32_batched_vs_single", model=model, n=n, site=b, nll_rel=e_nll, alpha_rel=e_alpha)
        # two roundings of the same fp64 truth.  Until round 3 the two schedules were bitwise equal in fp32 (every pass
        # continued ONE fmaf chain started at -C -- which is what made the log-determinant drift, csrc/dgp_gemm.h::
        # trailing_begin); round 3 had to loosen this bound to 2e-4 (measured 6e-5: the quadratic form of the unrefined
        # alpha differs by that much between the schedules).  With the refinement both schedules sit within 1e-5 of the
        # truth and the bound is back at 2e-5 (measured 2.2e-6, alpha 2.8e-6)
        assert e_nll <= 2e-5, e_nll
        assert e_alpha <= 5e-5, e_alpha
