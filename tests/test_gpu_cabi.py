"""The C ABI from a host written in plain C (examples/c_host.c): built with gcc against include/dgp_hip.h, run as its
own process, checked against the oracle on the same LCG-generated inputs."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import gp_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_c_host(out_path):
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
           os.path.join(ROOT, "examples", "c_host.c"), "-L", os.path.join(ROOT, "discontinuum_amd"), "-ldgp_hip",
           "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "discontinuum_amd"),
           "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", str(out_path)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]


def test_c_host_compiles_against_the_header(tmp_path):
    """No GPU needed: the header is valid C11 and every symbol the example uses links."""
    build_c_host(tmp_path / "c_host")


def lcg_inputs(n, d=3, m=5):
    state = 12345
    mask = (1 << 64) - 1

    def uniform():
        nonlocal state
        state = (state * 6364136223846793005 + 1442695040888963407) & mask
        return ((state >> 11) + 0.5) / 9007199254740992.0

    X, r = np.empty((n, d)), np.empty(n)
    for i in range(n):
        X[i, 0] = -16.0 + 32.0 * i / n + 0.01 * uniform()
        for j in range(1, d):
            X[i, j] = 4.0 * uniform() - 2.0
        r[i] = 2.0 * uniform() - 1.0
    theta = np.array([0.5 + 0.05 * p for p in range(orc.loadest_ntheta(d))])
    Xs = np.stack([X[7 * j] + 0.05 for j in range(m)])
    return X, r, np.full(n, 0.01), theta, Xs


@pytest.mark.gpu
def test_c_host_matches_the_oracle(tmp_path, gpu_device):
    exe = tmp_path / "c_host"
    build_c_host(exe)
    n = 333
    run = subprocess.run([str(exe), str(n)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    got = [float(v) for v in run.stdout.split()]
    X, r, noise, theta, Xs = (torch.tensor(a) for a in lcg_inputs(n))
    val, g_theta, g_r, _ = orc.nll_data_and_grads("loadest", X, r, noise, theta)
    Khat = orc.GRAMS["loadest"](X, X, theta) + torch.diag(noise)
    mu, var = orc.posterior("loadest", X, r, noise, theta, Xs)
    assert int(got[0]) == n and int(got[4]) == 0
    assert abs(got[1] - float(val)) <= 1e-10 * abs(float(val))
    assert abs(got[2] - float(torch.logdet(Khat))) <= 1e-9 * abs(float(torch.logdet(Khat)))
    assert abs(got[5] - float(g_theta[0])) <= 1e-8 * max(1.0, float(g_theta.abs().max()))
    assert abs(got[6] - float(g_r.sum())) <= 1e-8 * float(g_r.abs().sum())
    assert abs(got[7] - float(mu[0])) <= 1e-9 * max(1.0, float(mu.abs().max()))
    assert abs(got[8] - float(var[0])) <= 1e-8 * max(1.0, float(var.abs().max()))
