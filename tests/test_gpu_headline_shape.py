"""The SHAPE the headline runs, under parity (ADVICE r4 bench.py:669).

The bench line is 32 sites of n = 8192 in one batched plan: hyperparameters of more than 8 sites travel through device
scratch, and the default tile selectors pick `lauum_kernel` (TRI_ROW_LE / TRI_LOWER zero-work skipping, reversed k order),
whole 128 x 128 rounds of `syrk_kernel` and the 128-tile `trtri_level_kernel` BY THEMSELVES.  The dense oracle cannot reach
n = 8192 in test time, so every checked site is compared

  (a) with a single-site plan on the same inputs (a different schedule: pairs + split chain + early inverse against groups
      of four panels; different launches; the same matrix) -- that plan is the one `test_full_size_factor_identities_and_
      gradient` and the stage tests pin to the oracle; and
  (b) with the size-independent factor identities of tests/test_gpu_fullsize.py:58-68 on ITS OWN buffers.

Bounds, written before the first GPU run from the error model: both plans apply the same k-ordered fma chains per element,
so the expected difference is rounding noise of a few ulp amplified by cond(K^) ~ 1e5 in alpha: NLL 1e-11 (oracle tolerance
1e-10 / 10), gradients / alpha / dnoise 1e-9 (1e-8 / 10).  Identities as in test_gpu_fullsize.py: ||L L^T v - K v|| 1e-12,
||T L v - v|| 1e-9, ||S K v - v|| 1e-8, ||K alpha - r|| 1e-9, NLL from the factor 1e-12.
Reference: the loop body /root/reference/src/discontinuum/engines/gpytorch.py:350-384.
"""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _sym_matvec(M, v):
    L = torch.tril(M)
    return L @ v + torch.tril(M, -1).T @ v


@pytest.mark.parametrize("n,sites", [(8192, (0, 5, 11)), (4096, (0, 7, 11))])
def test_batch_of_12_with_default_selectors(n, sites, gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, d, B = gpu_device, 3, 12
    P = 2 * d + 5
    Xs, ys = zip(*[orc.synth_loadest(n, d, seed=40 + b) for b in range(B)])
    X = torch.tensor(np.stack(Xs), device=dev).contiguous()
    y = torch.tensor(np.stack(ys), device=dev).contiguous()
    noise = torch.full((B, n), 0.01, dtype=torch.float64, device=dev)
    g = torch.Generator().manual_seed(11)
    theta = orc.positive(0.3 * torch.randn(B, P, dtype=torch.float64, generator=g))  # a different theta per site
    pb = GPPlan("loadest", n, d, device=dev, lookahead=1, batch=B)
    # the defaults, not forced selectors: at these sizes they must already pick the benchmark's kernels
    nb = pb.N // 128
    assert pb.get_option(_lib.OPT_LAUUM64_MAX_TILES) < nb * (nb + 1) // 2 * B   # -> lauum_kernel (128 x 128, skipping)
    assert B > 8                                                                # -> theta through device scratch
    pb.set_inputs(X)
    out, alpha, dnoise = pb.fit_step(theta, y, noise)
    host = out.cpu()
    assert out.shape == (B, _lib.OUT_LEN) and bool((host[:, _lib.OUT_INFO] == 0).all())
    V = torch.randn(n, 4, dtype=torch.float64, generator=g).to(dev)
    p1 = GPPlan("loadest", n, d, device=dev)
    for b in sites:
        p1.set_inputs(X[b].contiguous())
        p1.stage_gram(theta[b], noise[b].contiguous())
        K = p1.buffer(_lib.BUF_A).clone()
        KV = _sym_matvec(K, V)
        # (b) the batched plan's OWN buffers of site b
        L = torch.tril(pb.buffer(_lib.BUF_A, site=b))
        T = torch.tril(pb.buffer(_lib.BUF_T, site=b))
        S = pb.buffer(_lib.BUF_S, site=b)
        assert (torch.linalg.norm(L @ (L.T @ V) - KV) / torch.linalg.norm(KV)).item() < 1e-12
        assert (torch.linalg.norm(T @ (L @ V) - V) / torch.linalg.norm(V)).item() < 1e-9
        assert (torch.linalg.norm(_sym_matvec(S, KV) - V) / torch.linalg.norm(V)).item() < 1e-8
        assert (torch.linalg.norm(_sym_matvec(K, alpha[b][:, None]) - y[b][:, None]) / torch.linalg.norm(y[b])).item() < 1e-9
        logdet = 2.0 * torch.log(torch.diagonal(L)).sum().item()
        nll = 0.5 * float((y[b] * alpha[b]).sum()) + 0.5 * logdet + 0.5 * n * np.log(2 * np.pi)
        assert abs(host[b, _lib.OUT_NLL].item() - nll) / abs(nll) < 1e-12
        assert torch.allclose(dnoise[b], 0.5 * (torch.diagonal(S) - alpha[b] ** 2), rtol=1e-12, atol=1e-14)
        del L, T, KV
        # (a) a plan of its own
        o1, a1, n1 = p1.fit_step(theta[b], y[b].contiguous(), noise[b].contiguous())
        o1 = o1.cpu()
        assert o1[_lib.OUT_INFO] == 0
        assert abs(host[b, 0] - o1[0]) <= 1e-11 * abs(o1[0])
        gs = slice(_lib.OUT_DTHETA, _lib.OUT_DTHETA + P)
        assert (host[b, gs] - o1[gs]).abs().max() <= 1e-9 * o1[gs].abs().max()
        assert (alpha[b] - a1).abs().max() <= 1e-9 * a1.abs().max()
        assert (dnoise[b] - n1).abs().max() <= 1e-9 * n1.abs().max()
        for k in (_lib.OUT_SUM_DR, _lib.OUT_SUM_DNOISE):
            assert abs(host[b, k] - o1[k]) <= 1e-9 * max(abs(o1[k]), a1.abs().max().item())
