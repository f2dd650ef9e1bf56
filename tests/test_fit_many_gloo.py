"""Training across ranks on CPU: ``fit_many_distributed`` over a two-rank gloo group.

Five ragged loadest sites (and three rating sites) are partitioned round-robin over two processes, every process trains
its share with ``fit_many`` on the oracle-backed plan double (tests/helpers.OraclePlan, batch surface), ONE ``all_gather``
at the end of the fit moves the fitted raw parameters / final objectives / iteration counts / stop reasons, and every rank
loads every site.  Checked against a single-process ``fit_many`` over all sites: bit for bit (the sites are independent,
the host algebra is evaluated per site under ``vmap``), identical tables on both ranks, ``predict`` on a site the rank did
not train, the failure path (a rank whose ``fit_many`` raises must not leave the other waiting in the gather), and a
world with more ranks than sites' worth of work (a rank that owns nothing).
Reference: one site per cloud worker, /root/reference/examples/nwqn-loadest-example/nwqn-loadest-example.py:38-125, 156-159.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import OraclePlan, loadest_dataset, rating_dataset

SIZES = (31, 44, 27, 38, 35)
ITERS = 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_engine():
    from discontinuum_amd import multisite_fit
    from discontinuum_amd.engines.hip import MarginalHIP

    MarginalHIP._plan_factory = staticmethod(OraclePlan)
    MarginalHIP.device = "cpu"
    multisite_fit.GPPlan = OraclePlan


def _sites(family, sizes=SIZES):
    if family == "loadest":
        from discontinuum_amd.loadest_gp import LoadestGP

        return [LoadestGP() for _ in sizes], [loadest_dataset(n, seed=10 + i) for i, n in enumerate(sizes)]
    from discontinuum_amd.rating_gp import RatingGP

    return [RatingGP() for _ in sizes], [rating_dataset(n, seed=20 + i) for i, n in enumerate(sizes)]


def _flat_params(m):
    return torch.cat([p.detach().reshape(-1).double() for _, p in sorted(m.model.named_parameters())]
                     + [p.detach().reshape(-1).double() for _, p in sorted(m.likelihood.named_parameters())])


def _worker(rank, world, port, q, family, sizes, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _cpu_engine()
        from discontinuum_amd import multisite_fit

        models, data = _sites(family, sizes)
        if mode == "fail" and rank == 1:  # this rank's fit_many raises before any collective
            def boom(*a, **k):
                raise RuntimeError("more than 10 consecutive NaN/Inf objectives (injected)")
            multisite_fit.fit_many = boom
        try:
            objs, table = multisite_fit.fit_many_distributed(models, data, iterations=ITERS, early_stopping=(mode == "early"),
                                                             patience=2 if mode == "early" else 60,
                                                             load=mode[5:] if mode.startswith("load=") else "all")
        except RuntimeError as e:
            q.put((rank, "error", str(e)))
            return
        if mode.startswith("load="):  # who holds which fitted site
            q.put((rank, "ok", {"table": table.numpy(), "fitted": [bool(m.is_fitted) for m in models]}))
            return
        other = 1 if rank == 0 else 0  # a site this rank did NOT train
        mu, se = models[other].predict(data[other][0])
        q.put((rank, "ok", {"objs": objs.numpy(), "table": table.numpy(), "params": [_flat_params(m).numpy() for m in models],
                            "fitted": [bool(m.is_fitted) for m in models], "its": [int(m._current_iteration) for m in models],
                            "pred": np.asarray(mu.values)}))
    finally:
        dist.destroy_process_group()


def _run(world, family, sizes=SIZES, mode="plain"):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, family, sizes, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return {r: (kind, payload) for r, kind, payload in got}


@pytest.mark.parametrize("family", ["loadest", "rating"])
def test_two_ranks_match_single_process_fit_many_bit_for_bit(family, monkeypatch):
    from discontinuum_amd import multisite_fit
    from discontinuum_amd.engines.hip import MarginalHIP

    sizes = SIZES if family == "loadest" else SIZES[:3]
    res = _run(2, family, sizes)
    assert all(kind == "ok" for kind, _ in res.values()), res
    # single process, all sites in one batch
    monkeypatch.setattr(MarginalHIP, "_plan_factory", staticmethod(OraclePlan))
    monkeypatch.setattr(MarginalHIP, "device", "cpu")
    monkeypatch.setattr(multisite_fit, "GPPlan", OraclePlan)
    torch.set_num_threads(2)
    models, data = _sites(family, sizes)
    objs = multisite_fit.fit_many(models, data, iterations=ITERS, site_seeds=list(range(len(sizes))))  # seed 0 + site index
    ref = [_flat_params(m).numpy() for m in models]
    for rank in (0, 1):
        out = res[rank][1]
        assert np.array_equal(out["objs"], objs.numpy()), (out["objs"] - objs.numpy())
        for i in range(len(sizes)):
            assert np.array_equal(out["params"][i], ref[i]), (rank, i, np.abs(out["params"][i] - ref[i]).max())
        assert out["fitted"] == [True] * len(sizes) and out["its"] == [ITERS - 1] * len(sizes)
        t = out["table"]
        assert t.shape[0] == len(sizes) and np.array_equal(t[:, -3], objs.numpy())
        assert np.array_equal(t[:, -2], np.full(len(sizes), ITERS)) and np.array_equal(t[:, -1], np.zeros(len(sizes)))
    assert np.array_equal(res[0][1]["table"], res[1][1]["table"])
    # predictions of a site on the rank that did not train it = the single process's
    for rank, other in ((0, 1), (1, 0)):
        mu, _ = models[other].predict(data[other][0])
        assert np.array_equal(res[rank][1]["pred"], np.asarray(mu.values))


def test_early_stopping_reason_and_iteration_count_travel():
    res = _run(2, "loadest", SIZES[:3], mode="early")
    assert all(kind == "ok" for kind, _ in res.values()), res
    t = res[0][1]["table"]
    assert np.array_equal(t, res[1][1]["table"])
    assert set(np.unique(t[:, -1])) <= {0.0, 1.0} and np.all(t[:, -2] >= 1) and np.all(t[:, -2] <= ITERS)
    assert [int(v) - 1 for v in t[:, -2]] == res[0][1]["its"]


def test_a_failing_rank_raises_everywhere_and_nobody_hangs():
    res = _run(2, "loadest", SIZES[:4], mode="fail")
    assert all(kind == "error" for kind, _ in res.values()), res
    assert "rank(s) [1]" in res[0][1] and "rank(s) [1]" in res[1][1]
    assert "injected" in res[1][1]  # the failing rank also says why


def test_a_rank_without_sites_takes_part():
    res = _run(3, "loadest", SIZES[:2])
    assert all(kind == "ok" for kind, _ in res.values()), res
    assert np.array_equal(res[0][1]["table"], res[2][1]["table"]) and res[2][1]["fitted"] == [True, True]


def test_load_modes_decide_who_holds_the_other_ranks_sites():
    """`load="rank0"`: only rank 0 loads the sites it did not train; `load="own"`: nobody does -- the table is gathered everywhere."""
    res = _run(2, "loadest", SIZES[:3], mode="load=rank0")
    assert res[0][1]["fitted"] == [True, True, True] and res[1][1]["fitted"] == [False, True, False]
    assert np.array_equal(res[0][1]["table"], res[1][1]["table"])
    res = _run(2, "loadest", SIZES[:3], mode="load=own")
    assert res[0][1]["fitted"] == [True, False, True] and res[1][1]["fitted"] == [False, True, False]
    assert np.array_equal(res[0][1]["table"], res[1][1]["table"]) and np.isfinite(res[0][1]["table"]).all()


def test_without_a_process_group_it_is_fit_many(monkeypatch):
    from discontinuum_amd import multisite_fit
    from discontinuum_amd.engines.hip import MarginalHIP

    monkeypatch.setattr(MarginalHIP, "_plan_factory", staticmethod(OraclePlan))
    monkeypatch.setattr(MarginalHIP, "device", "cpu")
    monkeypatch.setattr(multisite_fit, "GPPlan", OraclePlan)
    models, data = _sites("loadest", SIZES[:2])
    objs, table = multisite_fit.fit_many_distributed(models, data, iterations=3)
    m2, d2 = _sites("loadest", SIZES[:2])
    ref = multisite_fit.fit_many(m2, d2, iterations=3, site_seeds=[0, 1])
    assert torch.equal(objs, ref) and table.shape[0] == 2
    assert all(torch.equal(_flat_params(a), _flat_params(b)) for a, b in zip(models, m2))
    with pytest.raises(ValueError):
        multisite_fit.fit_many_distributed(models, data, iterations=1, return_state=True)
