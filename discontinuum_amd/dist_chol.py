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


def distributed_nll(plan, theta, r, noise, group_panels: int = 4, group=None):
    """-> ``out[32]`` (DGP_OUT_NLL / QUAD / LOGDET / INFO) on every rank.  ``plan`` is this rank's full-size single-site
    ``GPPlan`` with the (replicated) inputs already set; ``theta``, ``r``, ``noise`` are the same on every rank."""
    if plan.batch != 1:
        raise ValueError("distributed_nll needs a plain (unbatched) plan")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    W = int(group_panels)
    nbk = plan.N // 128
    plan.stage_gram(theta, noise)
    plan.dist_begin()
    A, Tm = plan.buffer(_lib.BUF_A), plan.buffer(_lib.BUF_T)
    for g in range((nbk + W - 1) // W):
        k0 = g * W
        w = min(W, nbk - k0)
        owner = g % world
        c0, c1 = k0 * 128, (k0 + w) * 128
        if rank == owner:
            plan.dist_factor_group(k0, w)
        if world > 1:
            # payload: the factored columns from the diagonal down + the inverses of their diagonal blocks
            npan, ninv = (plan.N - c0) * (c1 - c0), w * 128 * 128
            if rank == owner:
                inv = torch.stack([Tm[c:c + 128, c:c + 128] for c in range(c0, c1, 128)])
                payload = torch.cat([A[c0:, c0:c1].reshape(-1), inv.reshape(-1)])
            else:
                payload = torch.empty(npan + ninv, dtype=plan.dtype, device=plan.device)
            dist.broadcast(payload, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
            if rank != owner:
                A[c0:, c0:c1].copy_(payload[:npan].view(plan.N - c0, c1 - c0))
                inv = payload[npan:].view(w, 128, 128)
                for i, c in enumerate(range(c0, c1, 128)):
                    Tm[c:c + 128, c:c + 128].copy_(inv[i])
        plan.dist_update(k0, W, rank, world)
    stats = torch.tensor([plan.local_logdet(), float(plan.potrf_info())], dtype=torch.float64, device=plan.device)
    if world > 1:
        logdet = stats[:1].clone()
        info = stats[1:].clone()
        dist.all_reduce(logdet, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(info, op=dist.ReduceOp.MAX, group=group)
        stats = torch.cat([logdet, info])
    return plan.dist_finish(r, float(stats[0].item()), int(stats[1].item()))
