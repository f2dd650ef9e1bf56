"""The HIP path against the committed golden vectors (tests/golden/*.npz), through the C ABI.

Pure data comparison -- no oracle code runs here: each fixture holds the inputs (X, theta, noise, r) and the
expected data-term NLL, its gradients w.r.t. theta / r / noise, and (loadest) the latent posterior at 16 points.
Tolerances as in test_gpu_stages.py (fp64 path): NLL rel 1e-10, gradients rel 1e-8 of the max-norm, alpha and
dnoise rel 1e-8, posterior mean abs 1e-9, variance rel 1e-8.
"""
import glob
import os

import numpy as np
import pytest
import torch

from discontinuum_amd import _lib

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_reproduces_golden(path, gpu_device):
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    z = np.load(path)
    model = os.path.basename(path).split("_")[0]
    X = torch.tensor(z["X"], device=dev)
    n, d = X.shape
    theta = torch.tensor(z["theta"])
    r, noise = torch.tensor(z["r"], device=dev), torch.tensor(z["noise"], device=dev)
    p = GPPlan(model, n, d, dtype=torch.float64, device=dev)
    p.set_inputs(X.contiguous())
    out, alpha, dnoise = p.fit_step(theta, r, noise)
    out = out.cpu()
    assert int(out[_lib.OUT_INFO]) == 0
    nll = float(z["nll_data"])
    assert abs(float(out[_lib.OUT_NLL]) - nll) <= 1e-10 * abs(nll)
    g = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + p.ntheta].numpy()
    assert np.abs(g - z["grad_theta"]).max() <= 1e-8 * max(1.0, np.abs(z["grad_theta"]).max())
    assert np.abs(alpha.cpu().numpy() - z["alpha"]).max() <= 1e-8 * max(1.0, np.abs(z["alpha"]).max())
    assert np.abs(dnoise.cpu().numpy() - z["grad_noise"]).max() <= 1e-8 * max(1.0, np.abs(z["grad_noise"]).max())
    if model == "loadest" and z["Xs"].shape[0] != n:
        # loadest: constant mean raw[0], no noise added at prediction when m != n
        mu, var = p.predict(theta, torch.tensor(z["Xs"], device=dev))
        assert np.abs(mu.cpu().numpy() + z["raw"][0] - z["mu"]).max() < 1e-9
        assert (np.abs(var.cpu().numpy() - z["var"]) / (np.abs(z["var"]) + 1e-4)).max() < 1e-8
