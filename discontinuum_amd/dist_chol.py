"""One exact-GP matrix factored across the GPUs of a node (BASELINE config 5).

Nothing in the reference corresponds to this: its engines hold one matrix on one device.  The layout follows the
blueprint of SURVEY 8e with one simplification that 288 GB of HBM per GPU allows: every rank keeps a FULL-SIZE copy
of the matrix (n = 65536 fp32: 16 GiB) and of the plan's workspace, so the single-GPU kernels apply unchanged and
nothing but factored panels ever moves:

* block columns are dealt to the ranks in groups of ``W`` 128-wide panels, group g to rank g % world
  (1-D block-cyclic);
* every rank builds K^ itself (O(n^2), the inputs are replicated);
* for each group: the owner runs the panel chain of its W columns (``dgp_dist_factor_group``), the factored
  columns -- rows from the group's diagonal down, plus the inverses of their diagonal blocks -- are broadcast (RCCL over xGMI; gloo in the tests), and every
  rank applies them to the block columns IT owns right of the group (``dgp_dist_update``, K = 128 W);
* after the last group every rank holds all of L; the log-determinants of the owners are summed with one
  all-reduce and the forward solve L z = r for the quadratic form runs locally (``dgp_dist_finish``).

Per rank that is n^3 / (3 world) flops of MFMA work and n^2 / 2 elements received over the whole factorisation
(8.6 GB at n = 65536 fp32).  The result is the data term of the marginal likelihood (NLL, r^T K^^-1 r, log|K^|);
the gradient path (K^^-1) stays single-GPU, where one MI355X already factors and inverts this matrix.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib


def _pack(plan, A, Tm, c0, c1, w):
    inv = torch.stack([Tm[c:c + 128, c:c + 128] for c in range(c0, c1, 128)])
    return torch.cat([A[c0:, c0:c1].reshape(-1), inv.reshape(-1)])


def _unpack(plan, A, Tm, c0, c1, w, payload):
    npan = (plan.N - c0) * (c1 - c0)
    A[c0:, c0:c1].copy_(payload[:npan].view(plan.N - c0, c1 - c0))
    inv = payload[npan:].view(w, 128, 128)
    for i, c in enumerate(range(c0, c1, 128)):
        Tm[c:c + 128, c:c + 128].copy_(inv[i])


def distributed_nll(plan, theta, r, noise, group_panels: int = 4, group=None, lookahead: bool = True):
    """-> ``out[32]`` (DGP_OUT_NLL / QUAD / LOGDET / INFO) on every rank.  ``plan`` is this rank's full-size single-site
    ``GPPlan`` with the (replicated) inputs already set; ``theta``, ``r``, ``noise`` are the same on every rank.

    With ``lookahead`` the owner of group g+1 applies group g to that group's columns FIRST, factors it and starts its
    broadcast (asynchronous collective) while every rank -- the owner included -- is still applying group g to the
    rest of its columns; the panel chain and the transfer then hide behind the updates."""
    if plan.batch != 1:
        raise ValueError("distributed_nll needs a plain (unbatched) plan")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    W = int(group_panels)
    nbk = plan.N // 128
    ngroups = (nbk + W - 1) // W
    plan.stage_gram(theta, noise)
    plan.dist_begin()
    A, Tm = plan.buffer(_lib.BUF_A), plan.buffer(_lib.BUF_T)

    def span(g):
        k0 = g * W
        w = min(W, nbk - k0)
        return k0, w, k0 * 128, (k0 + w) * 128

    def src(g):
        owner = g % world
        return dist.get_global_rank(group, owner) if group is not None else owner

    def start_broadcast(g):  # collective: every rank calls it; returns (payload, work)
        k0, w, c0, c1 = span(g)
        if rank == g % world:
            payload = _pack(plan, A, Tm, c0, c1, w)
        else:
            payload = torch.empty((plan.N - c0) * (c1 - c0) + w * 128 * 128, dtype=plan.dtype, device=plan.device)
        return payload, dist.broadcast(payload, src=src(g), group=group, async_op=True)

    def finish_broadcast(g, payload, work):
        work.wait()
        if rank != g % world:
            k0, w, c0, c1 = span(g)
            _unpack(plan, A, Tm, c0, c1, w, payload)

    if rank == 0 % world:
        plan.dist_factor_group(0, span(0)[1])
    if world > 1:
        finish_broadcast(0, *start_broadcast(0))
    for g in range(ngroups):
        k0 = g * W
        nxt = g + 1
        pending = None
        if nxt < ngroups and lookahead:
            if rank == nxt % world:  # bring the next group up to date, factor it
                plan.dist_update(k0, W, rank, world, nxt * W, min((nxt + 1) * W, nbk))
                plan.dist_factor_group(nxt * W, span(nxt)[1])
            if world > 1:
                pending = start_broadcast(nxt)
            begin = (nxt + 1) * W if rank == nxt % world else 0
            plan.dist_update(k0, W, rank, world, begin, 0 if begin < nbk else nbk + 1)
            if pending is not None:
                finish_broadcast(nxt, *pending)
        else:
            plan.dist_update(k0, W, rank, world)
            if nxt < ngroups:
                if rank == nxt % world:
                    plan.dist_factor_group(nxt * W, span(nxt)[1])
                if world > 1:
                    finish_broadcast(nxt, *start_broadcast(nxt))
    stats = torch.tensor([plan.local_logdet(), float(plan.potrf_info())], dtype=torch.float64, device=plan.device)
    if world > 1:
        logdet = stats[:1].clone()
        info = stats[1:].clone()
        dist.all_reduce(logdet, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(info, op=dist.ReduceOp.MAX, group=group)
        stats = torch.cat([logdet, info])
    return plan.dist_finish(r, float(stats[0].item()), int(stats[1].item()))
