"""Parity at BASELINE.json's full size (n = 8192, d = 3, fp64) through size-independent properties --
the dense oracle would take minutes here, so the factors are checked against the matrix they came from
with random probe vectors, and the gradient against central finite differences of the NLL itself.

Tolerances: ||L L^T v - K v|| / ||K v|| < 1e-12; ||T L v - v|| / ||v|| < 1e-9; ||S K v - v|| / ||v|| < 1e-8;
alpha: ||K alpha - r|| / ||r|| < 1e-9; directional derivative vs finite difference rel 1e-6."""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _sym_matvec(M, v):
    """y = M v for a symmetric matrix stored in its lower triangle."""
    L = torch.tril(M)
    return L @ v + torch.tril(M, -1).T @ v


def test_full_size_factor_identities_and_gradient(gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    n, d = 8192, 3
    X, y = orc.synth_loadest(n, d, seed=0)
    Xd, yd = torch.tensor(X, device=dev), torch.tensor(y, device=dev)
    noise = torch.full((n,), 0.01, dtype=torch.float64, device=dev)
    theta = torch.tensor([0.9, 0.7, 1.0, 1.5, 0.6, 0.8, 1.2, 0.3, 0.9, 0.7, 1.1], dtype=torch.float64)
    g = torch.Generator(device="cpu").manual_seed(0)
    V = torch.randn(n, 4, dtype=torch.float64, generator=g).to(dev)
    p = GPPlan("loadest", n, d, device=dev)
    p.set_inputs(Xd)
    assert p.N == n
    # K^ itself (lower) before it is overwritten
    p.stage_gram(theta, noise)
    K = p.buffer(_lib.BUF_A).clone()
    KV = _sym_matvec(K, V)
    out, alpha, dnoise = p.fit_step(theta, yd, noise)
    out = out.cpu()
    assert out[_lib.OUT_INFO] == 0
    L = torch.tril(p.buffer(_lib.BUF_A))
    T = torch.tril(p.buffer(_lib.BUF_T))
    S = p.buffer(_lib.BUF_S)
    assert (torch.linalg.norm(L @ (L.T @ V) - KV) / torch.linalg.norm(KV)).item() < 1e-12
    assert (torch.linalg.norm(T @ (L @ V) - V) / torch.linalg.norm(V)).item() < 1e-9
    assert (torch.linalg.norm(_sym_matvec(S, KV) - V) / torch.linalg.norm(V)).item() < 1e-8
    assert (torch.linalg.norm(_sym_matvec(K, alpha[:, None]) - yd[:, None]) / torch.linalg.norm(yd)).item() < 1e-9
    # NLL pieces recomputed from the factor
    logdet = 2.0 * torch.log(torch.diagonal(L)).sum().item()
    quad = float((yd * alpha).sum())
    nll = 0.5 * quad + 0.5 * logdet + 0.5 * n * np.log(2 * np.pi)
    assert abs(out[_lib.OUT_NLL].item() - nll) / abs(nll) < 1e-12
    # dnoise_i = 1/2 (S_ii - alpha_i^2)
    assert torch.allclose(dnoise, 0.5 * (torch.diagonal(S) - alpha ** 2), rtol=1e-12, atol=1e-14)
    # directional derivative of the NLL w.r.t. theta vs central finite differences
    grad = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + 11]
    direction = torch.randn(11, dtype=torch.float64, generator=g)
    direction /= direction.norm()
    h = 1e-5
    fp = p.fit_step(theta + h * direction, yd, noise)[0][_lib.OUT_NLL].item()
    fm = p.fit_step(theta - h * direction, yd, noise)[0][_lib.OUT_NLL].item()
    fd = (fp - fm) / (2 * h)
    an = float(grad @ direction)
    assert abs(fd - an) / abs(an) < 1e-6
