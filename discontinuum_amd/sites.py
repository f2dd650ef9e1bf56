"""Independent sites across the GPUs of one node (BASELINE config 4).

The reference's only parallelism is "one site = one fit" mapped over AWS Lambda
(``examples/nwqn-loadest-example/nwqn-loadest-example.py:156-159``).  Here one process drives one GPU;
site i belongs to rank i mod G, every rank runs its sites through its own device plan, and the only
communication is one ``all_gather`` of the per-site result rows (NLL, info, gradients) -- RCCL over xGMI
when the tensors live on GPUs, gloo for the CPU tests.  There is no data-path collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def site_partition(n_sites: int, world: int, rank: int) -> list[int]:
    """Sites owned by ``rank``: round-robin, so a sweep of equally sized sites is balanced."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_sites, world))


def gather_site_results(local: torch.Tensor, n_sites: int, group=None) -> torch.Tensor:
    """All ranks contribute ``local`` (rows = their sites in ``site_partition`` order, R columns) and
    receive the full ``(n_sites, R)`` table in site order.  Ragged shares are padded for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != n_sites:
            raise ValueError("single-process gather needs all sites locally")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_sites + world - 1) // world
    mine = site_partition(n_sites, world, rank)
    if local.shape[0] != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} sites but passed {local.shape[0]} rows")
    pad = torch.zeros(per, local.shape[1], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty(n_sites, local.shape[1], dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = site_partition(n_sites, world, r)
        out[idx] = parts[r][: len(idx)]
    return out


def _side_stream(plan):
    """ONE side stream per plan object, created at first use and kept: the library keys a process-lifetime set of
    internal streams (bulk / rest / early) on every caller stream it sees (csrc/dgp_api.hip::stream_set), and a process has
    few hardware queues -- fresh streams per call would walk torch's pool of 32 and leave up to 32 such sets behind."""
    st = getattr(plan, "_side_stream", None)
    if st is None:
        st = plan._side_stream = torch.cuda.Stream(device=plan.device)
    return st


def _fit_sites_batched_plans(plans, Xs, rs, noises, theta):
    """Several batched plans, each with a contiguous share of the sites on its own stream; rows come back in input order."""
    dev = plans[0].device
    share = -(-len(Xs) // len(plans))
    ready = torch.cuda.Event()
    ready.record()  # inputs were produced on the caller's stream
    streams = [_side_stream(p) for p in plans]
    parts = []
    for k, (plan, st) in enumerate(zip(plans, streams)):
        lo, hi = k * share, min((k + 1) * share, len(Xs))
        if lo >= hi:
            break
        with torch.cuda.stream(st):
            st.wait_event(ready)
            parts.append(_fit_sites_batched(plan, Xs[lo:hi], rs[lo:hi], noises[lo:hi], theta))
    for st in streams:
        torch.cuda.current_stream(dev).wait_stream(st)
    if not parts:
        return torch.empty(0, 32, dtype=plans[0].dtype, device=dev)
    return torch.cat(parts)


def _fit_sites_batched(plan, Xs, rs, noises, theta):
    """Chunks of ``plan.batch`` sites per launch; sites may have fewer observations than the plan's n (ragged batch:
    their slots are zero-padded and ``set_site_sizes`` tells the kernels); a short last chunk repeats its last site."""
    B, n, rows = plan.batch, plan.n, []
    th = torch.as_tensor(theta, dtype=torch.float64).reshape(1, -1)

    def slot(t, width=None):  # one site's array in an n-row slot
        if t.shape[0] == n:
            return t
        if t.shape[0] > n:
            raise ValueError(f"a site has {t.shape[0]} observations but the plan holds {n}")
        pad = torch.zeros((n - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat([t, pad])

    for lo in range(0, len(Xs), B):
        idx = list(range(lo, min(lo + B, len(Xs))))
        pad = idx + [idx[-1]] * (B - len(idx))
        plan.set_site_sizes([Xs[i].shape[0] for i in pad])
        plan.set_inputs(torch.stack([slot(Xs[i]) for i in pad]).contiguous())
        out = plan.fit_step(th.repeat(B, 1), torch.stack([slot(rs[i]) for i in pad]).contiguous(),
                            torch.stack([slot(noises[i]) for i in pad]).contiguous())[0]
        rows.append(out[: len(idx)])
    if not rows:
        return torch.empty(0, 32, dtype=plan.dtype, device=plan.device)
    return torch.cat(rows)


def fit_sites(plans, Xs, rs, noises, theta):
    """One fit step for each local site; returns the stacked ``(B_local, OUT_LEN)`` result rows on the plans'
    device, in input order.  ``plans`` is

    * a BATCHED plan (``GPPlan(..., batch=B)``): B sites per launch in lockstep -- the MI355X-native form of the
      reference's map over sites: the sequential panel chain and the launch rate are amortised over the batch
      (measured on one MI355X, sites/s: n = 8192 80 -> 103 and n = 4096 275 -> 623 at B = 8, n = 300 3300 -> 180 000
      at B = 256); sites may have FEWER observations than the plan's n (ragged batch);
    * a list of BATCHED plans: the sites are cut into one contiguous share per plan and every plan runs its share on its
      own HIP stream -- at mid sizes one plan's sequential panel chain hides under the other's bulk updates (64 sites of
      n = 4096 as 2 x 32: 793 -> 822 fits/s; at n = 8192 a single plan of 32 already fills the GPU: no gain);
    * a plain plan, or a list of plain plans for the same (model, n, d): sites are dealt round-robin over them, each
      plan on its own HIP stream (two plans in flight: n = 4096 275 -> 400 sites/s; more add nothing)."""
    if not isinstance(plans, (list, tuple)) and getattr(plans, "batch", 1) > 1:
        return _fit_sites_batched(plans, list(Xs), list(rs), list(noises), theta)
    if isinstance(plans, (list, tuple)) and plans and all(getattr(p, "batch", 1) > 1 for p in plans):
        return _fit_sites_batched_plans(list(plans), list(Xs), list(rs), list(noises), theta)
    if not isinstance(plans, (list, tuple)):
        plans = [plans]
    on_gpu = plans[0].device.type == "cuda"
    if len(plans) > 1:  # several plans in flight: no third stream per plan (include/dgp_hip.h, dgp_plan_set_lookahead)
        for p in plans:
            if hasattr(p, "set_lookahead"):
                p.set_lookahead(1)
    streams = [_side_stream(p) for p in plans] if on_gpu else [None] * len(plans)
    if on_gpu:
        ready = torch.cuda.Event()
        ready.record()  # inputs were produced on the caller's stream
    rows = []
    for i, (X, r, noise) in enumerate(zip(Xs, rs, noises)):
        plan, st = plans[i % len(plans)], streams[i % len(plans)]
        if on_gpu:
            with torch.cuda.stream(st):
                st.wait_event(ready)
                plan.set_inputs(X)
                rows.append(plan.fit_step(theta, r, noise)[0])
        else:
            plan.set_inputs(X)
            rows.append(plan.fit_step(theta, r, noise)[0])
    if on_gpu:
        for st in streams:
            torch.cuda.current_stream(plans[0].device).wait_stream(st)
    if not rows:
        return torch.empty(0, 32, dtype=plans[0].dtype, device=plans[0].device)
    return torch.stack(rows)
