"""One matrix factored across ranks (BASELINE config 5; discontinuum_amd/dist_chol.py).

world = 1 checks the building blocks (group chain, owned-column update, block forward solve) against the fit step;
world = 2 and 3 run real ranks -- separate processes, each with a full-size plan of its own -- that share the one GPU
of the test box and talk over gloo (RCCL refuses several ranks on one device; the multi-GPU run uses nccl with
the same code).  Tolerances: NLL / quad / log-det rel 1e-11 against the single-plan fit step (different summation
order), NLL rel 1e-10 against the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _case(n, d, seed):
    X, y = orc.synth_loadest(n, d, seed)
    theta = orc.positive(0.3 * torch.randn(2 * d + 5, dtype=torch.float64, generator=torch.Generator().manual_seed(seed)))
    return torch.tensor(X), torch.tensor(y), torch.full((n,), 0.01, dtype=torch.float64), theta


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, W, lookahead, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from discontinuum_amd.backend import GPPlan
        from discontinuum_amd.dist_chol import distributed_nll

        dev = torch.device("cuda", 0)
        X, y, noise, theta = _case(n, d, 7)
        p = GPPlan("loadest", n, d, device=dev)
        p.set_inputs(X.to(dev).contiguous())
        out = distributed_nll(p, theta, y.to(dev).contiguous(), noise.to(dev).contiguous(), group_panels=W,
                              lookahead=lookahead)
        torch.cuda.synchronize()
        q.put((rank, out.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lookahead", [True, False])
@pytest.mark.parametrize("n,W", [(300, 1), (1500, 2), (2600, 4), (1000, 3)])
def test_single_rank_building_blocks(n, W, lookahead, gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan
    from discontinuum_amd.dist_chol import distributed_nll

    dev, d = gpu_device, 3
    X, y, noise, theta = _case(n, d, 3)
    p = GPPlan("loadest", n, d, device=dev)
    p.set_inputs(X.to(dev).contiguous())
    ref = p.fit_step(theta, y.to(dev), noise.to(dev))[0].cpu()
    out = distributed_nll(p, theta, y.to(dev).contiguous(), noise.to(dev).contiguous(), group_panels=W,
                          lookahead=lookahead).cpu()
    assert out[_lib.OUT_INFO] == 0
    for k in (_lib.OUT_NLL, _lib.OUT_QUAD, _lib.OUT_LOGDET):
        assert abs(out[k] - ref[k]) <= 1e-11 * abs(ref[k])


@pytest.mark.parametrize("world,n,W,lookahead", [(2, 1500, 2, True), (3, 2000, 2, True), (2, 900, 4, True), (3, 1300, 1, True),
                                                  (2, 1500, 2, False)])
def test_ranks_on_one_gpu_match_the_oracle(world, n, W, lookahead, gpu_device):
    from discontinuum_amd import _lib

    d, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, d, W, lookahead, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = dict(q.get(timeout=300) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    X, y, noise, theta = _case(n, d, 7)
    val, *_ = orc.nll_data_and_grads("loadest", X, y, noise, theta)
    for r in range(world):
        assert outs[r][_lib.OUT_INFO] == 0
        assert abs(outs[r][_lib.OUT_NLL] - val.item()) <= 1e-10 * abs(val.item())
        assert abs(outs[r][_lib.OUT_NLL] - outs[0][_lib.OUT_NLL]) <= 1e-13 * abs(outs[0][_lib.OUT_NLL])


def test_single_rank_fp32(gpu_device):
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan
    from discontinuum_amd.dist_chol import distributed_nll

    dev, n, d = gpu_device, 3000, 3
    X, y, noise, theta = _case(n, d, 11)
    p = GPPlan("loadest", n, d, dtype=torch.float32, device=dev)
    p.set_inputs(X.float().to(dev).contiguous())
    ref = p.fit_step(theta, y.float().to(dev), noise.float().to(dev))[0].cpu()
    out = distributed_nll(p, theta, y.float().to(dev).contiguous(), noise.float().to(dev).contiguous()).cpu()
    assert out[_lib.OUT_INFO] == 0
    assert abs(out[_lib.OUT_NLL] - ref[_lib.OUT_NLL]) <= 1e-4 * abs(ref[_lib.OUT_NLL])
