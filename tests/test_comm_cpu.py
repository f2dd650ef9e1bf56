"""The communicators of the distributed fit (discontinuum_amd/dist_chol.py) on CPU tensors: ``ThreadComm`` (ranks as
threads: the world-8 rehearsal of tests/test_gpu_dist.py) and ``TorchComm`` over a two-rank gloo group (the class the
multi-GPU run uses over RCCL).  No device arithmetic here -- only that every rank ends with the same, correct data."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from discontinuum_amd.dist_chol import ThreadComm, TorchComm, run_thread_ranks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _exercise(comm):
    """The three collectives the fit step uses, on rank-dependent data -> what every rank must agree on."""
    r, w = comm.rank, comm.world
    got = []
    for owner in range(w):  # broadcast from every owner in turn, as the block-cyclic schedule does
        buf = torch.full((5,), float(100 * owner + r)) if r != owner else torch.arange(5.0) + owner
        work = comm.broadcast(buf, owner)
        if work is not None:
            work.wait()
        got.append(buf.clone())
    t = torch.arange(4.0) * (r + 1)
    comm.all_reduce_sum(t)
    ga = comm.all_gather(torch.tensor([float(r), float(r * r)]))
    return torch.stack(got), t, ga, dict(comm.calls)


def _check(results, world):
    tri = world * (world + 1) / 2
    for got, t, ga, calls in results:
        for owner in range(world):
            assert torch.equal(got[owner], torch.arange(5.0) + owner)
        assert torch.equal(t, torch.arange(4.0) * tri)
        assert torch.equal(ga, torch.tensor([[float(k), float(k * k)] for k in range(world)]))
        assert calls == {"broadcast": world, "all_reduce": 1, "all_gather": 1}


@pytest.mark.parametrize("world", [2, 8])
def test_thread_ranks_collectives(world):
    _check(run_thread_ranks(world, _exercise), world)


def test_one_rank_skips_collectives_unless_forced():
    (got, t, ga, calls), = run_thread_ranks(1, _exercise)
    assert calls == {"broadcast": 0, "all_reduce": 0, "all_gather": 0} and ga.shape == (1, 2)
    c = TorchComm()  # no process group: one rank, nothing to do, nothing forced
    assert c.world == 1 and not c.active and c.broadcast(torch.zeros(3), 0) is None
    assert TorchComm(force=True).active is False  # forcing needs an initialised group to call into


def test_a_failing_rank_does_not_hang_the_others():
    def body(comm):
        if comm.rank == 2:
            raise ValueError("rank 2 fails")
        comm.all_reduce_sum(torch.ones(2))

    with pytest.raises(ValueError, match="rank 2 fails"):
        run_thread_ranks(4, body)


def _gloo_worker(rank, world, port, force, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _exercise(TorchComm(force=force))
        q.put((rank, [x.numpy() if torch.is_tensor(x) else x for x in res]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,force", [(2, False), (1, True)])
def test_torch_comm_over_gloo(world, force):
    """world 2: the N > 1 path; world 1 with force: what DGP_DIST_FORCE_COLLECTIVES=1 does (every collective issued)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, force, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _check([tuple(torch.as_tensor(x) if not isinstance(x, dict) else x for x in res[r]) for r in range(world)], world)


def test_stage_flops_add_up_to_the_fit_step():
    """DistributedFit.stage_flops (bench.py --config 5 prices its roofline with these counts): summed over the ranks the
    counts do not depend on how many ranks share the matrix -- the same tiles, dealt differently -- and each stage is
    the N^3/3 of its single-GPU counterpart up to the 128-block granularity (the owners' panel chains and diagonal-group
    inverses are not counted in `update` / `invert`)."""
    from discontinuum_amd.dist_chol import DistributedFit

    class Shape:  # stage_flops only reads the shape fields
        stage_flops = DistributedFit.stage_flops

    def total(world, W, nbk):
        tot = {"update": 0.0, "invert": 0.0, "product": 0.0}
        for rank in range(world):
            s = Shape()
            s.W, s.N, s.world, s.rank, s.ngroups = W, nbk * 128, world, rank, nbk // W
            for k, v in s.stage_flops().items():
                tot[k] += v
        return tot

    for W, nbk in ((4, 32), (4, 64), (1, 33), (2, 20), (4, 512)):
        ref = total(1, W, nbk)
        for world in (2, 3, 8):
            assert total(world, W, nbk) == ref, (world, W, nbk)
        third = (nbk * 128.0) ** 3 / 3
        assert 0.75 * third < ref["update"] < third
        assert 0.75 * third < ref["invert"] < 1.15 * third
        assert third < ref["product"] < 1.25 * third
    big = total(8, 4, 512)  # config 5: n = 65536
    assert abs(big["update"] / ((65536.0 ** 3) / 3) - 1) < 0.03 and abs(big["product"] / ((65536.0 ** 3) / 3) - 1) < 0.03
