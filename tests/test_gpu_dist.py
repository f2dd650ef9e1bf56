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
                                                          (2, "loadest", 3, 1500, 2, False), (4, "loadest", 3, 2100, 1, True),
                                                          # more ranks than column groups: rank 3 owns nothing
                                                          (4, "loadest", 3, 300, 1, True)])
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


# ---------------------------------------------------------------------------------------------------------------------
# world = 8 (BASELINE config 5's rank count).  The GPU boxes allow at most six processes on a card, so eight gloo ranks
# cannot share the test box's one GPU; the eight ranks are THREADS of this process instead (dist_chol.ThreadComm: the
# same DistributedFit code, every rank with its own handle, slabs and panel buffers; collectives are barriers around
# device copies).  What this pins at world = 8: the block-cyclic ownership maps, ragged last groups, ranks with several
# / one / no groups, the lookahead schedule and the partial-sum reductions.  It does NOT pin torch.distributed / RCCL
# behaviour at eight ranks -- that stays unmeasured until a node exists.
def _thread_world(world, model, d, n, W, dtype, dev, seed=7, lookahead=True, **kw):
    from discontinuum_amd.dist_chol import DistributedFit, run_thread_ranks

    X, r, noise, theta = make_case(model, d, n, seed=seed, perturb=0.2)
    Xd, rd, nd = (t.to(dev, dtype).contiguous() for t in (X, r, noise))

    def rank_body(comm):
        ctx = DistributedFit(model, n, d, dtype=dtype, device=dev, group_panels=W, lookahead=lookahead, comm=comm, **kw)
        ctx.set_inputs(Xd)
        out = ctx.fit_step(theta, rd, nd)
        return out.cpu().double(), ctx.alpha.cpu().double(), ctx.dnoise.cpu().double(), ctx.hbm_bytes(), dict(comm.calls), ctx.ngroups

    res = run_thread_ranks(world, rank_body, device=dev)
    torch.cuda.synchronize()
    return res, (X, r, noise, theta)


@pytest.mark.parametrize("model,d,n,W,lookahead", [("loadest", 3, 4200, 1, True), ("loadest", 3, 4200, 4, True),
                                                   ("rating", 2, 4100, 4, True), ("loadest", 3, 4200, 1, False),
                                                   ("loadest", 3, 300, 1, True)])  # the last: 3 groups, 5 idle ranks
def test_world8_thread_ranks_match_the_oracle(model, d, n, W, lookahead, gpu_device):
    from discontinuum_amd import _lib

    world = 8
    res, (X, r, noise, theta) = _thread_world(world, model, d, n, W, torch.float64, gpu_device, lookahead=lookahead)
    val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
    P = theta.numel()
    ng = res[0][5]
    assert ng == -(-n // (128 * W))
    for rk, (out, alpha, dnoise, hbm, calls, _) in enumerate(res):
        assert out[_lib.OUT_INFO] == 0
        assert abs(out[_lib.OUT_NLL] - val.item()) <= 1e-10 * abs(val.item())
        g = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P]
        assert (g - g_theta).abs().max() <= 1e-8 * g_theta.abs().max()
        assert (alpha - g_r).abs().max() <= 1e-8 * g_r.abs().max()
        assert (dnoise - g_noise).abs().max() <= 1e-8 * g_noise.abs().max()
        assert torch.equal(out, res[0][0])  # bitwise the same reduced row on every rank
        # two passes of one broadcast per group; z, alpha and the (dnoise, dtheta) tail; one gather of (log-det, info)
        assert calls == {"broadcast": 2 * ng, "all_reduce": 3, "all_gather": 1}
    # ranks hold their own column groups only
    N = ng * 128 * W
    cl = -(-ng // world) * 128 * W
    assert all(o[3] >= 3 * N * cl * 8 for o in res)
    if ng >= world:
        assert max(o[3] for o in res) < 0.5 * 3 * N * N * 8


def test_world8_thread_ranks_fp32_n16384_match_the_single_plan(gpu_device):
    """fp32 at a quarter of config 5's order, 32 groups of four panels over eight ranks, against the single-GPU plan in
    the same precision (measured: gpurun_out/fullsize_parity.jsonl)."""
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan
    from tests.test_gpu_fullsize import _record

    dev, n, d = gpu_device, 16384, 3
    res, (X, r, noise, theta) = _thread_world(8, "loadest", d, n, 4, torch.float32, dev, seed=11)
    p = GPPlan("loadest", n, d, dtype=torch.float32, device=dev)
    p.set_inputs(X.float().to(dev).contiguous())
    ref, ref_a, ref_dn = p.fit_step(theta, r.float().to(dev), noise.float().to(dev))
    ref, ref_a = ref.cpu().double(), ref_a.cpu().double()
    out, alpha = res[0][0], res[0][1]
    assert out[_lib.OUT_INFO] == 0 and ref[_lib.OUT_INFO] == 0
    e_nll = (abs(out[_lib.OUT_NLL] - ref[_lib.OUT_NLL]) / abs(ref[_lib.OUT_NLL])).item()
    g, gr = out[4:15], ref[4:15]
    e_grad = ((g - gr).abs().max() / gr.abs().max()).item()
    e_alpha = (torch.linalg.norm(alpha - ref_a) / torch.linalg.norm(ref_a)).item()
    _record(test="world8_threads_n16384_fp32_vs_single_plan", nll_rel=e_nll, grad_rel=e_grad, alpha_rel=e_alpha)
    # both sides refine alpha and the quadratic form against an fp64 residual since round 4 (before: 2.9e-5 / 6.7e-5 / 1.3e-3)
    assert e_nll <= 2e-6, e_nll  # measured 7.5e-8
    assert e_grad <= 2e-4, e_grad  # measured 2.0e-5
    assert e_alpha <= 1e-5, e_alpha  # measured 2.8e-7
    assert all(torch.equal(o[0], out) for o in res)


@pytest.mark.parametrize("world", [1, 3])
def test_fp32_refinement_across_ranks(world, gpu_device):
    """fp32 handles refine alpha and the quadratic form once against an fp64 residual (dgp_dist_residual; the single plan's
    DGP_OPT_REFINE): against the fp64 ORACLE, rating-gp n = 3000 (the ill-conditioned kernel of BASELINE config 3).
    Bounds = SURVEY section 8d's fp32 row (NLL rel 1e-4 n / 1024, gradients rel 1e-2) with the measured values beside them."""
    from discontinuum_amd import _lib
    from tests.test_gpu_fullsize import _record

    model, d, n, W = "rating", 2, 3000, 2
    errs = {}
    for refine in (False, True):
        res, (X, r, noise, theta) = _thread_world(world, model, d, n, W, torch.float32, gpu_device, seed=5, refine=refine)
        val, g_theta, g_r, g_noise = orc.nll_data_and_grads(model, X, r, noise, theta)
        out, alpha, dnoise, _, calls, _ = res[0]
        P = theta.numel()
        assert out[_lib.OUT_INFO] == 0
        assert all(torch.equal(o[0], out) and torch.equal(o[1], alpha) for o in res)  # the same bits on every rank
        if world > 1:
            assert calls["all_reduce"] == (5 if refine else 3)  # + the refinement's two solves
        g = out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + P]
        errs[refine] = dict(nll=(abs(out[_lib.OUT_NLL] - val.item()) / abs(val.item())).item(),
                            alpha=(torch.linalg.norm(alpha - g_r) / torch.linalg.norm(g_r)).item(),
                            grad=((g - g_theta).abs().max() / g_theta.abs().max()).item(),
                            dnoise=((dnoise - g_noise).abs().max() / g_noise.abs().max()).item())
    _record(test=f"dist_fp32_refinement_world{world}", refined=errs[True], unrefined=errs[False])
    e = errs[True]
    assert e["nll"] <= 1e-4 * n / 1024, errs
    assert e["alpha"] <= 1e-4 and e["alpha"] <= 0.2 * errs[False]["alpha"], errs
    assert e["grad"] <= 1e-2 and e["dnoise"] <= 1e-2, errs
    assert e["nll"] <= errs[False]["nll"] + 1e-7, errs


# ---------------------------------------------------------------------------------------------------------------------
# RCCL with ONE rank: `force_collectives` makes a world-1 run issue every broadcast / all_reduce / all_gather on the
# "nccl" backend (a one-rank communicator is legal).  This is the only way RCCL can see this code on a one-GPU box; it
# pins the call signatures, dtypes, buffer slicing and the async-work handling -- not multi-rank ordering or timing.
def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from discontinuum_amd.dist_chol import DistributedFit
        from discontinuum_amd.sites import gather_site_results

        rows = {}
        for dtype in (torch.float64, torch.float32):
            X, r, noise, theta = make_case("loadest", 3, 1500, seed=7, perturb=0.2)
            ctx = DistributedFit("loadest", 1500, 3, dtype=dtype, device=dev, group_panels=2, force_collectives=True)
            ctx.set_inputs(X.to(dev, dtype).contiguous())
            out = ctx.fit_step(theta, r.to(dev, dtype).contiguous(), noise.to(dev, dtype).contiguous())
            torch.cuda.synchronize()
            rows[str(dtype)] = (out.cpu().double().numpy(), ctx.alpha.cpu().double().numpy(), dict(ctx.comm.calls), ctx.ngroups)
        table = gather_site_results(torch.arange(12.0, device=dev).reshape(3, 4), 3)  # the batch gather of bench.py / sites.py
        q.put((dist.get_backend(), rows, table.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_rccl_is_really_called_with_one_rank(gpu_device):
    from discontinuum_amd import _lib

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    pr.start()
    backend, rows, table = q.get(timeout=600)
    pr.join(timeout=120)
    assert pr.exitcode == 0 and backend == "nccl"
    X, r, noise, theta = make_case("loadest", 3, 1500, seed=7, perturb=0.2)
    val, g_theta, g_r, _ = orc.nll_data_and_grads("loadest", X, r, noise, theta)
    for key, tol_nll, tol_g in ((str(torch.float64), 1e-10, 1e-8), (str(torch.float32), 1e-4, 1e-2)):
        out, alpha, calls, ng = rows[key]
        # (fp32 handles refine alpha once: two more solves, i.e. two more sums)
        assert calls == {"broadcast": 2 * ng, "all_reduce": 5 if key == str(torch.float32) else 3, "all_gather": 1}, calls
        assert out[_lib.OUT_INFO] == 0
        assert abs(out[_lib.OUT_NLL] - val.item()) <= tol_nll * abs(val.item())
        g = torch.tensor(out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + 11])
        assert (g - g_theta).abs().max() <= tol_g * g_theta.abs().max()
    assert (table == torch.arange(12.0).reshape(3, 4).numpy()).all()
