"""N > 1 path on CPU: two gloo ranks shard a batch of sites, evaluate them (oracle-backed plan double)
and gather the result table -- the same ``site_partition`` / ``gather_site_results`` / ``fit_sites``
code the multi-GPU bench runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from discontinuum_amd import _lib
from discontinuum_amd.sites import fit_sites, gather_site_results, site_partition
from oracle import gp_oracle as orc
from tests.helpers import OraclePlan

N_SITES, N, D = 5, 24, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _site(i):
    X, y = orc.synth_loadest(N, D, seed=i)
    return torch.tensor(X), torch.tensor(y), torch.full((N,), 0.01, dtype=torch.float64)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = site_partition(N_SITES, world, rank)
        data = [_site(i) for i in mine]
        plan = OraclePlan("loadest", N, D)
        theta = torch.full((orc.loadest_ntheta(D),), 0.6931471805599453, dtype=torch.float64)
        local = fit_sites(plan, [d[0] for d in data], [d[1] for d in data], [d[2] for d in data], theta)
        table = gather_site_results(local, N_SITES)
        q.put((rank, table.numpy()))
    finally:
        dist.destroy_process_group()


def test_partition_is_round_robin_and_complete():
    parts = [site_partition(11, 4, r) for r in range(4)]
    assert parts[0] == [0, 4, 8] and parts[3] == [3, 7]
    assert sorted(i for p in parts for i in p) == list(range(11))
    with pytest.raises(ValueError):
        site_partition(4, 2, 2)


def test_two_rank_gloo_gather_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: all sites in one process
    theta = torch.full((orc.loadest_ntheta(D),), 0.6931471805599453, dtype=torch.float64)
    ref = []
    for i in range(N_SITES):
        X, y, noise = _site(i)
        val, g, _, _ = orc.nll_data_and_grads("loadest", X, y, noise, theta)
        ref.append((val.item(), g.numpy()))
    for rank in range(world):
        table = results[rank]
        assert table.shape == (N_SITES, _lib.OUT_LEN)
        for i in range(N_SITES):
            assert np.isclose(table[i, _lib.OUT_NLL], ref[i][0], rtol=1e-12)
            assert np.allclose(table[i, _lib.OUT_DTHETA:_lib.OUT_DTHETA + 9], ref[i][1], rtol=1e-10)
    assert np.array_equal(results[0], results[1])
