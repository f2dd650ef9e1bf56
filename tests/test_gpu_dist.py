"""One matrix factored, inverted and differentiated across ranks (BASELINE config 5; discontinuum_amd/dist_chol.py).

world = 1 runs the whole column-slab pipeline on one rank against the single-GPU fit step and the CPU oracle;
world = 2 and 3 run real ranks -- separate processes, each holding ONLY its own column groups -- that share the one GPU
of the test box and talk over gloo (RCCL refuses several ranks on one device; the multi-GPU run uses nccl with the same
code).  Tolerances (fp64): NLL / quad / log-det rel 1e-11 against the single-plan fit step (different summation order),
NLL rel 1e-10 and gradients (theta, r, noise) rel 1e-8 against the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gp_oracle as orc
from tests.test_gpu_stages import make_case

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(model, d, n, W, lookahead, dtype, dev, seed=7, with_grad=True):
    from discontinuum_amd.dist_chol import DistributedFit

    X, r, noise, theta = make_case(model, d, n, seed=seed, perturb=0.2)
    ctx = DistributedFit(model, n, d, dtype=dtype, device=dev, group_panels=W, lookahead=lookahead)
    ctx.set_inputs(X.to(dev, dtype).contiguous())
    out = ctx.fit_step(theta, r.to(dev, dtype).contiguous(), noise.to(dev, dtype).contiguous(), with_grad=with_grad)
    torch.cuda.synchronize()
    return ctx, out.cpu().double(), (X, r, noise, theta)


def _worker(rank, world, port, model, d, n, W, lookahead, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        ctx, out, _ = _run(model, d, n, W, lookahead, torch.float64, dev)
        q.put((rank, out.numpy(), ctx.alpha.cpu().numpy(), ctx.dnoise.cpu().numpy(), ctx.hbm_bytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lookahead", [True, False])
@pytest.mark.parametrize("model,d,n,W", [("loadest", 3, 300, 1), ("loadest", 3, 1500, 2), ("loadest", 3, 2600, 4),
                                         ("loadest", 2, 1000, 3), ("rating", 2, 1100, 4), ("rating", 2, 700, 2)])
def test_single_rank_pipeline_matches_the_fit_step_and_the_oracle(model, d, n, W, lookahead, gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev = gpu_device
    ctx, out, (X, r, noise, theta) = _run(model, d, n, W, lookahead, torch.float64, dev)
    assert out[_lib.OUT_INFO] == 0
    p = GPPlan(model, n, d, device=dev)
    p.set_inputs(X.to(dev).contiguous())
    ref, ref_dr, ref_dn = p.fit_step(theta, r.to(dev), noise.to(dev))
    ref = ref.cpu()
    P = theta.numel()
    for k in (_lib.OUT_NLL, _lib.OUT_QUAD, _lib.OUT_LOGDET):
        assert abs(out[k] - ref[k]) <= 1e-11 * abs(ref[k]), k
    g, gr = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P], ref[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P]
    assert (g - gr).abs().max() <= 1e-9 * gr.abs().max()
    assert (ctx.alpha - ref_dr).abs().max() <= 1e-9 * ref_dr.abs().max()
    assert (ctx.dnoise - ref_dn).abs().max() <= 1e-9 * ref_dn.abs().max()
    assert abs(out[_lib.OUT_SUM_DR] - ref[_lib.OUT_SUM_DR]) <= 1e-9 * ref_dr.abs().sum().item()
    assert abs(out[_lib.OUT_SUM_DNOISE] - ref[_lib.OUT_SUM_DNOISE]) <= 1e-9 * ref_dn.abs().sum().item()
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    assert abs(out[_lib.OUT_NLL] - val) <= 1e-10 * abs(val)
    assert (g - g_theta).abs().max() <= 1e-8 * g_theta.abs().max()
    # value only: same NLL, no gradient work
    _, out0, _ = _run(model, d, n, W, lookahead, torch.float64, dev, with_grad=False)
    assert out0[_lib.OUT_NLL] == out[_lib.OUT_NLL] and out0[_lib.OUT_DTHETA:].abs().max() == 0


@pytest.mark.parametrize("world,model,d,n,W,lookahead", [(2, "loadest", 3, 1500, 2, True), (3, "loadest", 3, 2000, 2, True),
                                                          (2, "rating", 2, 900, 4, True), (3, "loadest", 3, 1300, 1, True),
                                                          (2, "loadest", 3, 1500, 2, False), (4, "loadest", 3, 2100, 1, True)])
def test_ranks_on_one_gpu_match_the_oracle(world, model, d, n, W, lookahead, gpu_device):
    from discontinuum_amd import _lib

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model, d, n, W, lookahead, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    outs = {rk: (torch.tensor(o), torch.tensor(a), torch.tensor(dn), hb) for rk, o, a, dn, hb in got}
    X, r, noise, theta = make_case(model, d, n, seed=7, perturb=0.2)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    P = theta.numel()
    full = 3 * (((n + 127) // 128 * 128) ** 2) * 8  # A, T, S of a single-GPU plan
    for rk in range(world):
        out, alpha, dnoise, hbm = outs[rk]
        assert out[_lib.OUT_INFO] == 0
        assert abs(out[_lib.OUT_NLL] - val.item()) <= 1e-10 * abs(val.item())
        g = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P]
        assert (g - g_theta).abs().max() <= 1e-8 * g_theta.abs().max()
        assert (alpha - g_r).abs().max() <= 1e-8 * g_r.abs().max()
        assert (dnoise - g_noise).abs().max() <= 1e-8 * g_noise.abs().max()
        assert torch.equal(out, outs[0][0])  # every rank holds the same reduced results
    # a rank holds its own column groups only (+ two panel buffers and the gradient partials), not a replica
    N = -(-n // (128 * W)) * 128 * W
    cl = -(-(N // (128 * W)) // world) * 128 * W
    slabs = 3 * N * cl * 8
    assert all(o[3] >= slabs for o in outs.values())
    if world >= 3:
        assert max(o[3] for o in outs.values()) < 0.8 * full + 2 * N * 128 * W * 8 + 64 * (N // 64) * (cl // 64) * 8


def test_single_rank_fp32(gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev, n, d = gpu_device, 3000, 3
    ctx, out, (X, r, noise, theta) = _run("loadest", d, n, 4, True, torch.float32, dev, seed=11)
    p = GPPlan("loadest", n, d, dtype=torch.float32, device=dev)
    p.set_inputs(X.float().to(dev).contiguous())
    ref = p.fit_step(theta, r.float().to(dev), noise.float().to(dev))[0].cpu().double()
    assert out[_lib.OUT_INFO] == 0
    assert abs(out[_lib.OUT_NLL] - ref[_lib.OUT_NLL]) <= 1e-4 * abs(ref[_lib.OUT_NLL])
    g, gr = out[4:15], ref[4:15]
    assert (g - gr).abs().max() <= 2e-2 * gr.abs().max()


def test_not_positive_definite_is_reported(gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.dist_chol import DistributedFit

    dev, n, d = gpu_device, 700, 3
    X, r, noise, theta = make_case("loadest", d, n, seed=5, perturb=0.1)
    X[400] = X[100]  # repeated observation with (nearly) no noise: singular
    bad = noise.clone()
    bad[100] = bad[400] = -0.5
    ctx = DistributedFit("loadest", n, d, device=dev, group_panels=2)
    ctx.set_inputs(X.to(dev).contiguous())
    out = ctx.fit_step(theta, r.to(dev), bad.to(dev)).cpu()
    assert out[_lib.OUT_INFO] >= 1 and not torch.isfinite(out[_lib.OUT_NLL])
    out = ctx.fit_step(theta, r.to(dev), noise.to(dev)).cpu()  # and the next call recovers
    assert out[_lib.OUT_INFO] == 0 and torch.isfinite(out[_lib.OUT_NLL])
