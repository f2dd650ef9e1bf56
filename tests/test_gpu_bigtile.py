"""The kernels that carry the benchmark -- `lauum_kernel`, the 128 x 128 rounds of the bulk `syrk_kernel` and the
128-tile `trtri_level_kernel`, all on the direct-to-LDS core -- against the DENSE CPU oracle.

By default the library picks them from the problem size (K^^-1 = L^-T L^-1 needs > 1000 tiles x batch, i.e. one site of
n > 5 700; bulk-update rounds of 512 tiles; inverse levels of >= 1024 tiles), which no dense-oracle test reaches.  The
plan-level options `DGP_OPT_LAUUM64_MAX_TILES = 0`, `DGP_OPT_SYRK_SLOTS = 4`, `DGP_OPT_TRTRI_SMALL = 0`
(include/dgp_hip.h) force them on at n = 1000 and n = 1300 (ragged: N = 1408, 11 block columns), where the oracle runs in
a second: with 4 slots a bulk launch of t tiles has t // 4 whole rounds of 128 x 128 tiles AND a cut remainder.

Tolerances are those of tests/test_gpu_stages.py: fp64 Gram 1e-13 abs, factors 1e-9, NLL 1e-10, gradients / alpha /
dnoise 1e-8; fp32 (the reference's dtype, src/discontinuum/engines/gpytorch.py:221-222; with the fp64 refinement of
alpha) NLL 1e-4 max(1, n / 1024), gradients 1e-2.  What the fit step replaces:
src/discontinuum/engines/gpytorch.py:350-384."""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc
from tests.test_gpu_stages import make_case, tril_n

pytestmark = pytest.mark.gpu

SIZES = [("loadest", 3, 1000), ("loadest", 3, 1300), ("rating", 2, 1000), ("rating", 2, 1300)]


def force_big_tiles(p):
    from discontinuum_amd import _lib

    p.set_option(_lib.OPT_LAUUM64_MAX_TILES, 0)
    p.set_option(_lib.OPT_SYRK_SLOTS, 4)
    p.set_option(_lib.OPT_TRTRI_SMALL, 0)
    assert p.get_option(_lib.OPT_LAUUM64_MAX_TILES) == 0 and p.get_option(_lib.OPT_SYRK_SLOTS) == 4
    assert p.get_option(_lib.OPT_TRTRI_SMALL) == 0
    return p


def forced_plan(model, d, n, X, dtype, dev, lookahead=True, batch=1):
    from discontinuum_amd.backend import GPPlan

    p = force_big_tiles(GPPlan(model, n, d, dtype=dtype, device=dev, lookahead=lookahead, batch=batch))
    if batch == 1:
        p.set_inputs(X.to(dev, dtype).contiguous())
    return p


@pytest.mark.parametrize("model,d,n", SIZES)
@pytest.mark.parametrize("lookahead", [True, False])
def test_stages_fp64_on_the_128_tile_kernels(model, d, n, lookahead, gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=1, perturb=0.3)
    Khat = orc.GRAMS[model](X, X, theta) + torch.diag(noise)
    p = forced_plan(model, d, n, X, torch.float64, dev, lookahead)
    p.stage_gram(theta, noise.to(dev))
    assert (tril_n(p.buffer(_lib.BUF_A), n) - torch.tril(Khat)).abs().max() < 1e-13
    p.stage_potrf()  # bulk update: whole rounds of 128 x 128 tiles + a cut remainder
    L_ref = torch.linalg.cholesky(Khat)
    L = tril_n(p.buffer(_lib.BUF_A), n)
    assert torch.linalg.norm(L - L_ref) / torch.linalg.norm(L_ref) < 1e-9
    p.stage_trtri()  # every level in 128 x 128 tiles
    T = tril_n(p.buffer(_lib.BUF_T), n)
    eye = torch.eye(n, dtype=torch.float64)
    assert torch.linalg.norm(T @ L_ref - eye) / np.sqrt(n) < 1e-9
    p.stage_lauum()  # lauum_kernel
    S = tril_n(p.buffer(_lib.BUF_S), n)
    S_full = S + S.T - torch.diag(torch.diagonal(S))
    assert torch.linalg.norm(S_full @ Khat - eye) / np.sqrt(n) < 1e-7
    S_ref = torch.cholesky_inverse(L_ref)
    assert (S - torch.tril(S_ref)).abs().max() / S_ref.abs().max() < 1e-9
    p.stage_solve(r.to(dev))
    alpha_ref = torch.cholesky_solve(r[:, None], L_ref)[:, 0]
    assert (p.buffer(_lib.BUF_ALPHA)[:n].cpu() - alpha_ref).abs().max() / alpha_ref.abs().max() < 1e-8
    g = p.stage_grad(theta).cpu()
    _, g_ref, _, _ = orc.nll_data_and_grads(model, X, r, noise, theta)
    assert (g - g_ref).abs().max() / g_ref.abs().max() < 1e-8


@pytest.mark.parametrize("model,d,n", SIZES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fit_step_on_the_128_tile_kernels(model, d, n, dtype, gpu_device):
    from discontinuum_amd import _lib

    dev = gpu_device
    X, r, noise, theta = make_case(model, d, n, seed=2, perturb=0.2)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    P = theta.numel()
    for lookahead in (2, 1, 0):
        p = forced_plan(model, d, n, X, dtype, dev, lookahead)
        out, dr, dnoise = p.fit_step(theta, r.to(dev, dtype), noise.to(dev, dtype))
        out = out.cpu().double()
        assert out[_lib.OUT_INFO] == 0
        e_nll = (abs(out[_lib.OUT_NLL] - val) / abs(val)).item()
        e_g = ((out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P] - g_theta).abs().max() / g_theta.abs().max()).item()
        e_a = ((dr.cpu().double() - g_r).abs().max() / g_r.abs().max()).item()
        e_n = ((dnoise.cpu().double() - g_noise).abs().max() / g_noise.abs().max()).item()
        if dtype == torch.float64:
            assert e_nll < 1e-10 and e_g < 1e-8 and e_a < 1e-8 and e_n < 1e-8, (lookahead, e_nll, e_g, e_a, e_n)
        else:
            assert e_nll < 1e-4 * max(1.0, n / 1024) and e_g < 1e-2, (lookahead, e_nll, e_g)
            assert e_a < 1e-2 and e_n < 2e-2, (lookahead, e_a, e_n)


@pytest.mark.parametrize("model,d,sizes", [("loadest", 3, [1300, 1000, 1171]), ("rating", 2, [1000, 1300, 640, 1300])])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_ragged_batch_on_the_128_tile_kernels(model, d, sizes, dtype, gpu_device):
    """A ragged batched plan (gridDim.z = sites; groups of four panels; the group's columns in 128-tiles too) with the
    128-tile kernels forced on, site by site against the oracle."""
    from discontinuum_amd import _lib

    dev, B, n = gpu_device, len(sizes), max(sizes)
    cases = [make_case(model, d, nb, seed=40 + b, perturb=0.2) for b, nb in enumerate(sizes)]
    X = torch.full((B, n, d), float("nan"), dtype=torch.float64)
    r = torch.full((B, n), float("nan"), dtype=torch.float64)
    noise = torch.full((B, n), float("nan"), dtype=torch.float64)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        X[b, :nb], r[b, :nb], noise[b, :nb] = c[0], c[1], c[2]
    theta = torch.stack([c[3] for c in cases])
    pb = forced_plan(model, d, n, None, dtype, dev, lookahead=1, batch=B)
    pb.set_site_sizes(sizes)
    pb.set_inputs(X.to(dev, dtype).contiguous())
    out, dr, dnoise = pb.fit_step(theta, r.to(dev, dtype).contiguous(), noise.to(dev, dtype).contiguous())
    out = out.cpu().double()
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, c[0], c[1], c[2], c[3])
        P = c[3].numel()
        assert out[b, _lib.OUT_INFO] == 0
        e_nll = (abs(out[b, _lib.OUT_NLL] - val) / abs(val)).item()
        e_g = ((out[b, _lib.OUT_DTHETA:_lib.OUT_DTHETA + P] - g_theta).abs().max() / g_theta.abs().max()).item()
        e_a = ((dr[b, :nb].cpu().double() - g_r).abs().max() / g_r.abs().max()).item()
        e_n = ((dnoise[b, :nb].cpu().double() - g_noise).abs().max() / g_noise.abs().max()).item()
        if dtype == torch.float64:
            assert e_nll < 1e-10 and e_g < 1e-8 and e_a < 1e-8 and e_n < 1e-8, (b, e_nll, e_g, e_a, e_n)
        else:
            assert e_nll < 1e-4 * max(1.0, nb / 1024) and e_g < 1e-2 and e_a < 1e-2 and e_n < 2e-2, (b, e_nll, e_g, e_a, e_n)
        assert bool((dr[b, nb:] == 0).all()) and bool((dnoise[b, nb:] == 0).all())


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_chain_yield_is_a_scheduling_hint_only(dtype, gpu_device):
    """`DGP_OPT_CHAIN_YIELD` (single-site plans: the bulk update's waves that share the diagonal-block kernel's compute unit
    sleep while it runs; csrc/dgp_common.h `yield_if_asked`) changes WHEN tiles are computed, never what: the whole result
    row, alpha and dnoise are bitwise the same with the hint off -- on the 128-tile bulk kernel that carries it (forced by
    the tile selectors) and with the default selectors."""
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, model, d, n = gpu_device, "loadest", 3, 2600
    X, r, noise, theta = make_case(model, d, n, seed=4, perturb=0.3)
    rd, nd = r.to(dev, dtype).contiguous(), noise.to(dev, dtype).contiguous()
    for forced in (True, False):
        rows = []
        for hint in (1, 0, 1):
            p = GPPlan(model, n, d, dtype=dtype, device=dev)
            if forced:
                force_big_tiles(p)
            p.set_option(_lib.OPT_CHAIN_YIELD, hint)
            assert p.get_option(_lib.OPT_CHAIN_YIELD) == hint
            p.set_inputs(X.to(dev, dtype).contiguous())
            out, alpha, dnoise = p.fit_step(theta, rd, nd)
            out2 = p.fit_step(theta, rd, nd)[0]  # (and repeatable)
            assert torch.equal(out, out2)
            rows.append((out.cpu(), alpha.cpu(), dnoise.cpu()))
        assert rows[0][0][_lib.OUT_INFO] == 0 and bool(torch.isfinite(rows[0][0][:15]).all())
        for o, a, dn in rows[1:]:
            assert torch.equal(o, rows[0][0]) and torch.equal(a, rows[0][1]) and torch.equal(dn, rows[0][2])


@pytest.mark.parametrize("model,d,n,B", [("loadest", 3, 1300, 1), ("rating", 2, 1000, 1), ("loadest", 4, 900, 3), ("loadest", 3, 1100, 10)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_fused_gradient_epilogue_against_the_two_launch_path(model, d, n, B, dtype, gpu_device):
    """`DGP_OPT_FUSED_GRAD` (round 5, csrc/dgp_fused.hip): every 128 x 128 tile of K^^-1 = L^-T L^-1 contracts itself with
    dK/dtheta right after it is stored, instead of a second kernel streaming K^^-1 again.  Against the two-launch path
    (lauum_kernel + gram_grad_kernel) on the same plan: K^^-1, the NLL, alpha and dnoise are BITWISE the same (the k-loop and
    the store are lauum_kernel's), the hyperparameter gradient is the same sum of the same products in a different order
    (128- instead of 64-tiles in the first stage).  Error model for the bound: ~n^2 / 2 = 5e5 terms whose sum cancels to
    ~1e-2 of their absolute sum, each product exact to eps: relative difference <= 100 eps sqrt(n^2 / 2) ~ 1e-11 (fp64),
    6e-3 worst case in fp32 where every pair evaluation itself carries ~1e-6 (asserted at 2e-4, like the oracle bound / 50).
    Site 0's fused gradient is also held against the dense oracle at the usual 1e-8 (fp32: 1e-2).  B = 10: hyperparameters
    through device scratch.  Reference: the backward of engines/gpytorch.py:384."""
    from discontinuum_amd import _lib

    dev = gpu_device
    cases = [make_case(model, d, n, seed=60 + b, perturb=0.25) for b in range(B)]
    X = torch.stack([c[0] for c in cases]).to(dev, dtype).contiguous()
    r = torch.stack([c[1] for c in cases]).to(dev, dtype).contiguous()
    noise = torch.stack([c[2] for c in cases]).to(dev, dtype).contiguous()
    theta = torch.stack([c[3] for c in cases])
    P = theta.shape[1]
    if B == 1:
        X, r, noise, theta = X[0].contiguous(), r[0].contiguous(), noise[0].contiguous(), theta[0]
    p = forced_plan(model, d, n, X if B == 1 else None, dtype, dev, lookahead=1 if B > 1 else 2, batch=B)
    p.set_inputs(X)
    rows = {}
    for fused in (1, 0):
        p.set_option(_lib.OPT_FUSED_GRAD, fused)
        assert p.get_option(_lib.OPT_FUSED_GRAD) == fused
        out, alpha, dnoise = p.fit_step(theta, r, noise)
        S = [torch.tril(p.buffer(_lib.BUF_S, site=b))[:n, :n].clone() for b in (0, B - 1)]
        rows[fused] = (out.reshape(B, -1).cpu().double(), alpha.cpu(), dnoise.cpu(), S)
        assert bool((rows[fused][0][:, _lib.OUT_INFO] == 0).all())
    (of, af, nf, Sf), (o2, a2, n2, S2) = rows[1], rows[0]
    assert torch.equal(af, a2) and torch.equal(nf, n2) and all(torch.equal(x, y) for x, y in zip(Sf, S2))
    keep = [k for k in range(_lib.OUT_LEN) if not _lib.OUT_DTHETA <= k < _lib.OUT_DTHETA + P]
    assert torch.equal(of[:, keep], o2[:, keep])
    g = slice(_lib.OUT_DTHETA, _lib.OUT_DTHETA + P)
    err = ((of[:, g] - o2[:, g]).abs().max(dim=1).values / o2[:, g].abs().max(dim=1).values).max().item()
    assert err <= (1e-11 if dtype == torch.float64 else 2e-4), err
    _, g_ref, _, _ = orc.nll_data_and_grads(model, *cases[0][:3], cases[0][3])
    e_g = ((of[0, g] - g_ref).abs().max() / g_ref.abs().max()).item()
    assert e_g < (1e-8 if dtype == torch.float64 else 1e-2), e_g
    # and repeatable bit for bit
    p.set_option(_lib.OPT_FUSED_GRAD, 1)
    again = p.fit_step(theta, r, noise)[0].reshape(B, -1).cpu().double()
    assert torch.equal(again, of)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("model,d,sizes", [("loadest", 3, [1300, 1000, 1171, 1300, 900]), ("rating", 2, [1408, 1300, 1408, 1100])])
def test_group_panel_gemm_option(model, d, sizes, dtype, gpu_device):
    """`DGP_OPT_GROUP_GEMM` (round 5; batched plans of >= 4 sites): a panel group's rows below its 512 x 512 diagonal block are
    solved by ONE GEMM against that block's inverse, L[i, group] = A[i, group] T_D^T (csrc/dgp_chol.hip::trsm_group_kernel),
    instead of panel-by-panel trsm / column-update launches.  N = 1408 = 11 block columns: two full groups take the new path,
    the last (three panels, nothing below) the old one.  Against the dense oracle at the usual tolerances (fp64 NLL 1e-10,
    gradients / alpha / dnoise 1e-8; fp32 NLL 1e-4 max(1, n / 1024), rest 1e-2), and in fp64 against the same plan with the
    option off: a different association of the same sums.  Error model for that bound: the panel entries differ by
    ~cond(L_D) eps <= 1e3 x 1.1e-16 relative, the NLL (a sum over n pivots and n residuals) by less: asserted 1e-11 on the
    NLL and 1e-9 on gradients / alpha, a tenth of the oracle tolerances.  Measured neutral in speed (default off).
    What it restates: the Cholesky inside the reference's `mll(output, y)`, engines/gpytorch.py:350-353."""
    from discontinuum_amd import _lib

    dev, B, n = gpu_device, len(sizes), max(sizes)
    cases = [make_case(model, d, nb, seed=80 + b, perturb=0.2) for b, nb in enumerate(sizes)]
    X = torch.full((B, n, d), float("nan"), dtype=torch.float64)
    r = torch.full((B, n), float("nan"), dtype=torch.float64)
    noise = torch.full((B, n), float("nan"), dtype=torch.float64)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        X[b, :nb], r[b, :nb], noise[b, :nb] = c[0], c[1], c[2]
    theta = torch.stack([c[3] for c in cases])
    pb = forced_plan(model, d, n, None, dtype, dev, lookahead=1, batch=B)
    pb.set_site_sizes(sizes)
    pb.set_inputs(X.to(dev, dtype).contiguous())
    rd, nd = r.to(dev, dtype).contiguous(), noise.to(dev, dtype).contiguous()
    rows = {}
    for opt in (1, 0):
        pb.set_option(_lib.OPT_GROUP_GEMM, opt)
        assert pb.get_option(_lib.OPT_GROUP_GEMM) == opt
        out, dr, dnoise = pb.fit_step(theta, rd, nd)
        rows[opt] = (out.cpu().double(), dr.cpu().double(), dnoise.cpu().double())
    out, dr, dnoise = rows[1]
    P = theta.shape[1]
    g = slice(_lib.OUT_DTHETA, _lib.OUT_DTHETA + P)
    for b, (nb, c) in enumerate(zip(sizes, cases)):
        val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, c[0], c[1], c[2], c[3])
        assert out[b, _lib.OUT_INFO] == 0
        e_nll = (abs(out[b, _lib.OUT_NLL] - val) / abs(val)).item()
        e_g = ((out[b, g] - g_theta).abs().max() / g_theta.abs().max()).item()
        e_a = ((dr[b, :nb] - g_r).abs().max() / g_r.abs().max()).item()
        e_n = ((dnoise[b, :nb] - g_noise).abs().max() / g_noise.abs().max()).item()
        if dtype == torch.float64:
            assert e_nll < 1e-10 and e_g < 1e-8 and e_a < 1e-8 and e_n < 1e-8, (b, e_nll, e_g, e_a, e_n)
            o0, a0, _ = rows[0]
            assert abs(out[b, 0] - o0[b, 0]) <= 1e-11 * abs(o0[b, 0])
            assert (out[b, g] - o0[b, g]).abs().max() <= 1e-9 * o0[b, g].abs().max()
            assert (dr[b] - a0[b]).abs().max() <= 1e-9 * a0[b].abs().max()
        else:
            assert e_nll < 1e-4 * max(1.0, nb / 1024) and e_g < 1e-2 and e_a < 1e-2 and e_n < 2e-2, (b, e_nll, e_g, e_a, e_n)
