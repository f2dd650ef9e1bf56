"""One exact-GP matrix factored, inverted and differentiated across the GPUs of a node (BASELINE config 5,
SURVEY.md section 8e): a whole FIT STEP -- NLL and every gradient -- of a matrix that is spread over the ranks.

Nothing in the reference corresponds to this: its engines hold one matrix on one device.  The layout is SURVEY 8e's:
1-D block-cyclic by column groups of ``group_panels`` 128-wide panels, group g on rank g % world, and a rank holds ONLY
its own groups (three column slabs for K^ -> L, L^-1 and K^^-1; n = 65536 fp32 on 8 ranks: 3 x 2.1 GB) plus two panel
buffers.  All O(n^2) / O(n^3) arithmetic is in ``libdgp_hip.so`` (``dgp_dist_*``, csrc/dgp_dist.hip); this module is the
communication schedule on top of ``torch.distributed`` (backend "nccl" = RCCL over xGMI; gloo in the tests):

* pass 1, per group: the owner factors its columns (panel chain + inverse of the diagonal block) and broadcasts them
  ONCE -- (N - c0) x GW + GW x GW elements; every rank applies that payload twice: to its block columns of K^ right of
  the group (trailing update) and to its columns of L^-1 (forward substitution), so the inverse costs no extra traffic.
  With ``lookahead`` the owner of group g + 1 updates that group first, factors it and starts its broadcast while every
  rank is still applying group g.
* O(n) vectors: z = L^-1 r, alpha = K^^-1 r and the log-determinant are sums of per-rank partials (three all-reduces).
* pass 2, per group: the owner broadcasts its columns of L^-1 -- (N - c0) x GW elements -- and every rank forms
  K^^-1 [that group's rows, its own later columns] = T_J^T T_I.
* gradient: each rank contracts its part of K^^-1 - alpha alpha^T with dK/dtheta; one all-reduce of ntheta numbers (and
  of the n-vector 1/2 (diag K^^-1 - alpha^2)).

Per rank: N^3 / world flops on MFMA and N^2 elements received over both passes (17 GB at n = 65536 fp32).

STATUS of the communication layer: the schedule has run over gloo (ranks as processes sharing one GPU, world 2..5) and
over ``ThreadComm`` (ranks as threads of one process, world up to 8 and beyond -- the GPU boxes of this project allow at
most six processes on a card).  The "nccl" (= RCCL) backend has only ever been entered with ONE rank
(``DGP_DIST_FORCE_COLLECTIVES=1``: a world-1 communicator really calls RCCL's broadcast / all_reduce / all_gather); a run
on several GPUs has not happened -- its timing is UNMEASURED and its stream-ordering assumptions (below) are UNPINNED.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading

import torch
import torch.distributed as dist

from . import _lib
from .backend import _DTYPES, _ptr, _stream, _theta_array, model_id


class TorchComm:
    """The ranks are the processes of a ``torch.distributed`` group ("nccl" = RCCL over xGMI on a node; gloo in tests).
    With one rank every collective is skipped unless ``force`` (or ``DGP_DIST_FORCE_COLLECTIVES=1``) asks for it: a
    world-1 RCCL communicator is legal, and that is how the RCCL entry points of this module are exercised on a
    one-GPU box."""

    def __init__(self, group=None, rank=None, world=None, force=None):
        initialised = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world = int(world) if world is not None else (dist.get_world_size(group) if initialised else 1)
        self.rank = int(rank) if rank is not None else (dist.get_rank(group) if initialised else 0)
        if self.world > 1 and not initialised:
            raise RuntimeError("world > 1 needs an initialised torch.distributed process group")
        if force is None:
            force = os.environ.get("DGP_DIST_FORCE_COLLECTIVES", "0") not in ("", "0")
        self.active = self.world > 1 or (bool(force) and initialised)
        self.calls = {"broadcast": 0, "all_reduce": 0, "all_gather": 0}  # collectives really issued (tests)

    def _global(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def broadcast(self, buf, owner):
        """Start the broadcast of ``buf`` from rank ``owner``; -> work handle (``.wait()``) or None."""
        if not self.active:
            return None
        self.calls["broadcast"] += 1
        return dist.broadcast(buf, src=self._global(owner), group=self.group, async_op=True)

    def all_reduce_sum(self, t):
        if self.active:
            self.calls["all_reduce"] += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t):
        """-> (world, *t.shape): every rank's ``t``, in rank order."""
        if not self.active:
            return t.unsqueeze(0)
        self.calls["all_gather"] += 1
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.stack(parts)


class ThreadComm:
    """The ranks are THREADS of one process that share one GPU and torch's default stream: a rehearsal communicator for
    world sizes a one-GPU box cannot host as processes (world = 8 is BASELINE config 5's; the boxes allow six processes
    per card).  ``ThreadComm.make(world)`` returns one communicator per rank; each rank's thread builds its own
    ``DistributedFit(..., comm=comms[r])`` and calls the same methods.  Collectives are barriers around device copies on
    the shared stream -- enqueue order is execution order, so a payload is complete before any rank reads it, and sums
    run in rank order (every rank gets bitwise the same result)."""

    class _Shared:
        def __init__(self, world):
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank, world):
        self._sh, self.rank, self.world = shared, rank, world
        self.active = world > 1
        self.calls = {"broadcast": 0, "all_reduce": 0, "all_gather": 0}

    @classmethod
    def make(cls, world):
        shared = cls._Shared(world)
        return [cls(shared, r, world) for r in range(world)]

    def _exchange(self, t):
        sh = self._sh
        sh.slots[self.rank] = t
        sh.barrier.wait()  # every rank's tensor is registered and its producers are enqueued
        return sh.slots

    def broadcast(self, buf, owner):
        if not self.active:
            return None
        self.calls["broadcast"] += 1
        slots = self._exchange(buf)
        if self.rank != owner:
            buf.copy_(slots[owner])
        self._sh.barrier.wait()  # every copy is enqueued before the owner may reuse its buffer
        return None

    def all_reduce_sum(self, t):
        if not self.active:
            return t
        self.calls["all_reduce"] += 1
        slots = self._exchange(t)
        total = slots[0].clone()
        for k in range(1, self.world):
            total += slots[k]
        self._sh.barrier.wait()  # every rank has read every contribution
        t.copy_(total)
        self._sh.barrier.wait()
        return t

    def all_gather(self, t):
        if not self.active:
            return t.unsqueeze(0)
        self.calls["all_gather"] += 1
        slots = self._exchange(t)
        out = torch.stack([slots[k] for k in range(self.world)])
        self._sh.barrier.wait()
        return out


def run_thread_ranks(world: int, fn, device=None):
    """``fn(comm)`` on ``world`` threads, one ``ThreadComm`` rank each -> the results in rank order.  A rank that raises
    breaks the barrier, so the others fail too instead of waiting for it; the first exception is re-raised."""
    comms = ThreadComm.make(world)
    results, errors = [None] * world, [None] * world

    def body(r):
        try:
            if device is not None:
                torch.cuda.set_device(device)
            results[r] = fn(comms[r])
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
            comms[r]._sh.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), name=f"dgp-rank-{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real or any(errors):
        raise (real[0] if real else next(e for e in errors if e is not None))
    return results


class DistributedFit:
    """This rank's share of ONE (model, n, d) exact-GP matrix.  Every rank constructs one with the same arguments and
    calls the same methods with the same (replicated) ``X``, ``theta``, ``r``, ``noise``."""

    def __init__(self, model: str, n: int, d: int, dtype=torch.float64, device="cuda", rank: int | None = None,
                 world: int | None = None, group_panels: int | None = None, group=None, lookahead: bool = True, comm=None,
                 force_collectives: bool | None = None, refine: bool = True):
        """``refine`` (fp32 only, the single plan's ``DGP_OPT_REFINE``): one step of iterative refinement of alpha and the
        quadratic form against an fp64 residual.  ``comm``: the communicator (default ``TorchComm`` on ``group`` / the default process group; ``ThreadComm`` for
        in-process ranks).  ``force_collectives`` (default: environment ``DGP_DIST_FORCE_COLLECTIVES``): issue every
        collective even with one rank."""
        mid = model_id(model)
        if dtype not in _DTYPES:
            raise ValueError("dtype must be torch.float64 or torch.float32")
        if group_panels is None:
            # K = 128 group_panels per read-modify-write pass of the slab GEMMs: measured on one rank at n = 65536 fp32,
            # 2 / 4 / 8 panels -> 2.73 / 2.50 / 2.36 s per fit step (gpurun_out/r3/c5_w.txt); small matrices keep 4 so that
            # every rank still owns several groups
            group_panels = 8 if n >= 32768 else 4
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("discontinuum_amd requires a ROCm GPU (MI355X); there is no CPU fallback")
        self.comm = comm if comm is not None else TorchComm(group, rank, world, force_collectives)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.model, self.n, self.d, self.dtype, self.device = model, int(n), int(d), dtype, torch.device(device)
        self.lookahead = bool(lookahead)
        self.refine = bool(refine) and dtype == torch.float32
        h = C.c_void_p()
        _lib.check(self.lib.dgp_dist_create(mid, _DTYPES[dtype], self.n, self.d, self.rank, self.world,
                                            int(group_panels), C.byref(h)), "dgp_dist_create")
        self._h = h
        self.ntheta = self.lib.dgp_model_ntheta(mid, self.d)
        self.N = int(self.lib.dgp_dist_padded_n(h))
        self.ngroups = int(self.lib.dgp_dist_groups(h))
        self.W = int(group_panels)
        nbytes = int(self.lib.dgp_dist_workspace_bytes(h))
        with torch.cuda.device(self.device):
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self._ws.data_ptr()) % 256
            _lib.check(self.lib.dgp_dist_set_workspace(h, C.c_void_p(self._ws.data_ptr() + off), nbytes), "dgp_dist_set_workspace")
            pmax = int(self.lib.dgp_dist_panel_elems(h, 0))
            self._panels = [torch.empty(pmax, dtype=dtype, device=self.device) for _ in range(2)]
        self.alpha = self.dnoise = None
        self._timing = None

    # ------------------------------------------------------------------ measurement (bench.py --config 5)
    STAGES = ("gram", "factor", "update", "invert", "pack", "product", "grad")
    _STAGE_OF = {"dgp_dist_gram": "gram", "dgp_dist_factor": "factor", "dgp_dist_update": "update", "dgp_dist_invert": "invert",
                 "dgp_dist_pack_inverse": "pack", "dgp_dist_product": "product", "dgp_dist_grad_partial": "grad"}

    def set_timing(self, enabled: bool):
        """HIP events (torch's, on the stream every ``dgp_dist_*`` kernel is launched on) around each library call of the
        following fit steps; ``get_timing()`` sums them per stage for the most recent step."""
        self._timing = {} if enabled else None

    def get_timing(self) -> dict:
        if self._timing is None:
            raise RuntimeError("enable timing and run fit_step first")
        torch.cuda.synchronize(self.device)
        return {st: sum(a.elapsed_time(b) for a, b in self._timing.get(st, [])) for st in self.STAGES}

    def stage_flops(self) -> dict:
        """Algorithmic flops THIS rank executes per fit step in each MFMA stage, counted tile by tile exactly as the
        kernels of csrc/dgp_dist.hip enumerate them (2 m n k per tile product; the owner's panel chain and the diagonal
        group inverse -- ``factor`` -- are not counted)."""
        W, nbk, world, rank, N = self.W, self.N // 128, self.world, self.rank, self.N
        ngl = -(-self.ngroups // world)
        gblock = lambda lb: ((lb // W) * world + rank) * W + lb % W  # noqa: E731
        owned_below = lambda g: 0 if g <= rank else (g - rank + world - 1) // world  # noqa: E731
        tile = 2.0 * 128 * 128
        fl = {"update": 0.0, "invert": 0.0, "product": 0.0}
        for g in range(self.ngroups):
            row0 = (g + 1) * W  # first block row below the group
            for lb in range(owned_below(g + 1) * W, ngl * W):  # update: own block columns right of g
                bj = gblock(lb)
                if bj < nbk:
                    fl["update"] += (nbk - max(bj, row0)) * tile * 128 * W
            nbelow = owned_below(g) * W
            fl["invert"] += nbelow * sum(tile * 128 * (i + 1) for i in range(W))  # T[G, H] = -Linv X[G, H]
            rows_below = nbk - row0
            if rows_below > 0:
                fl["invert"] += rows_below * nbelow * tile * 128 * W
                if g % world == rank:
                    fl["invert"] += rows_below * sum(tile * 128 * (W - h) for h in range(W))
            for lb in range(owned_below(g) * W, ngl * W):  # product: S[group rows, own columns >= the group]
                bi = gblock(lb)
                if bi >= nbk:
                    continue
                for jq in range(W):
                    if bi >= g * W + jq:  # tiles on or above the diagonal of S^T (kernel: col_i + BT > col_j)
                        fl["product"] += tile * (N - bi * 128)
        return fl

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                torch.cuda.synchronize(self.device)
            except Exception:  # noqa: BLE001
                pass
            self.lib.dgp_dist_destroy(h)
            self._h = None

    def hbm_bytes(self) -> int:
        return int(self._ws.numel() + sum(p.numel() * p.element_size() for p in self._panels))

    # ------------------------------------------------------------------ plumbing
    def _owner(self, g: int) -> int:
        return g % self.world

    def _payload(self, g: int, with_inverse_block: bool):
        elems = int(self.lib.dgp_dist_panel_elems(self._h, g))
        if not with_inverse_block:
            gw = 128 * self.W
            elems -= gw * gw
        return self._panels[g % 2][:elems]

    def _broadcast(self, buf, g: int):
        return self.comm.broadcast(buf, self._owner(g))

    def _sum(self, t):
        return self.comm.all_reduce_sum(t)

    def _call(self, name, *args):
        st = self._STAGE_OF.get(name) if self._timing is not None else None
        if st is None:
            _lib.check(getattr(self.lib, name)(self._h, *args, _stream()), name)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(getattr(self.lib, name)(self._h, *args, _stream()), name)
        b.record()
        self._timing.setdefault(st, []).append((a, b))

    def set_inputs(self, X: torch.Tensor):
        if not (torch.is_tensor(X) and X.is_cuda and X.dtype == self.dtype and X.is_contiguous() and X.numel() == self.n * self.d):
            raise ValueError(f"X must be a contiguous ({self.n}, {self.d}) {self.dtype} CUDA tensor")
        with torch.cuda.device(self.device):
            self._call("dgp_dist_set_inputs", _ptr(X))
        self._X = X

    def slab(self, which: int) -> torch.Tensor:
        """(N, Cl) view of one of the rank's slabs (tests)."""
        p = C.c_void_p()
        _lib.check(self.lib.dgp_dist_slab(self._h, which, C.byref(p)), "dgp_dist_slab")
        cl = int(self.lib.dgp_dist_slab_columns(self._h))
        esz = torch.empty((), dtype=self.dtype).element_size()
        off = p.value - self._ws.data_ptr()
        return self._ws[off:off + self.N * cl * esz].view(self.dtype).view(self.N, cl)

    # ------------------------------------------------------------------ the two passes
    def _pipeline(self, produce, consume):
        """Groups 0 .. ngroups-1 in order: the owner of g runs ``produce(g, payload)``, the payload is broadcast, every
        rank runs ``consume(g, payload, next_group_owned)``; with lookahead the next group is produced and its broadcast
        started BEFORE the bulk of ``consume(g)``, through ``consume``'s first phase.

        Buffer reuse: payload g lives in ``_panels[g % 2]``, so the broadcast of g + 2 overwrites what ``consume(g)``
        read.  That is safe only if every access is STREAM-ORDERED before that broadcast: ``produce`` and ``consume``
        launch on torch's current stream; under "nccl" ``work.wait()`` makes the current stream wait for the collective
        and the collective's own stream waits for the current stream at the time ``dist.broadcast`` is called
        (ProcessGroupNCCL semantics) -- which is after ``consume(g)`` was enqueued, because ``start(g + 2)`` is only
        reached in iteration g + 1.  gloo and ``ThreadComm`` are host-synchronous / same-stream, so the order is
        trivially kept there; the RCCL ordering has never run on more than one rank (module docstring: UNPINNED)."""
        ng = self.ngroups
        pending = {}

        def start(g, early=None):
            buf = self._cur_payload(g)
            if self.rank == self._owner(g):
                if early is not None:
                    early()
                produce(g, buf)
            pending[g] = self._broadcast(buf, g)

        start(0)
        for g in range(ng):
            work = pending.pop(g)
            if work is not None:
                work.wait()
            buf = self._cur_payload(g)
            nxt = g + 1
            mine_next = nxt < ng and self.rank == self._owner(nxt)
            if nxt < ng and self.lookahead:
                start(nxt, early=(lambda g=g, buf=buf, nxt=nxt: consume(g, buf, "next-only")) if mine_next else None)
                consume(g, buf, "rest" if mine_next else "all")
            else:
                consume(g, buf, "all")
                if nxt < ng:
                    start(nxt)

    def _factor_and_invert(self, theta_arr, noise):
        self._call("dgp_dist_gram", theta_arr, _ptr(noise))
        W, nbk = self.W, self.N // 128
        self._cur_payload = lambda g: self._payload(g, True)

        def produce(g, buf):
            self._call("dgp_dist_factor", g, _ptr(buf))

        def consume(g, buf, part):
            lo, hi = (g + 1) * W, min((g + 2) * W, nbk)  # block columns of the next group
            if part == "next-only":
                self._call("dgp_dist_update", g, _ptr(buf), lo, hi)
                return
            # "rest": this rank has already brought the next group's columns up to date ("next-only")
            self._call("dgp_dist_update", g, _ptr(buf), hi if part == "rest" else 0, 0)
            self._call("dgp_dist_invert", g, _ptr(buf))

        self._pipeline(produce, consume)

    def _inverse_products(self):
        self._cur_payload = lambda g: self._payload(g, False)

        def produce(g, buf):
            self._call("dgp_dist_pack_inverse", g, _ptr(buf))

        def consume(g, buf, part):
            if part != "next-only":  # nothing of pass 2 has to precede the next group's payload
                self._call("dgp_dist_product", g, _ptr(buf))

        self._pipeline(produce, consume)

    # ------------------------------------------------------------------ public
    def fit_step(self, theta, r: torch.Tensor, noise: torch.Tensor, with_grad: bool = True) -> torch.Tensor:
        """-> ``out[32]`` (layout DGP_OUT_*: NLL, quad, log-det, info, dNLL/dtheta, sum dr, -, -, sum dnoise), the same
        on every rank; ``self.alpha`` (= dNLL/dr) and ``self.dnoise`` (n,) are left on the device.  ``with_grad=False``
        stops after the value (no K^^-1, no second pass)."""
        for t, name in ((r, "r"), (noise, "noise")):
            if not (torch.is_tensor(t) and t.is_cuda and t.dtype == self.dtype and t.is_contiguous() and t.numel() == self.n):
                raise ValueError(f"{name} must be a contiguous {self.dtype} CUDA tensor with {self.n} elements")
        th = _theta_array(theta, self.ntheta)
        N, n, dt, dev = self.N, self.n, self.dtype, self.device
        if self._timing is not None:
            self._timing = {}
        with torch.cuda.device(dev):
            self._factor_and_invert(th, noise)
            stat = torch.empty(2, dtype=dt, device=dev)
            self._call("dgp_dist_status", _ptr(stat))
            # ONE gather of the (log-det part, info) pairs: the sum runs in rank order on every rank (bitwise the same
            # everywhere) and the reported pivot is the FIRST failing one among the ranks' columns
            allstat = self.comm.all_gather(stat.double())
            bad_piv = allstat[:, 1]
            first = torch.where(bad_piv > 0, bad_piv, torch.full_like(bad_piv, float("inf"))).min()
            stat64 = torch.stack([allstat[:, 0].sum(), torch.where(torch.isinf(first), torch.zeros_like(first), first)])
            z = torch.empty(N, dtype=dt, device=dev)
            self._call("dgp_dist_solve_partial", _ptr(r), _ptr(z))
            self._sum(z)
            alpha = torch.empty(N, dtype=dt, device=dev)
            self._call("dgp_dist_alpha_partial", _ptr(z), _ptr(alpha))
            self._sum(alpha)
            out = torch.zeros(_lib.OUT_LEN, dtype=dt, device=dev)
            quad = (z.double() * z.double()).sum()
            if self.refine:
                # rho = r - K^ alpha0 in double (K^ re-evaluated, every rank the whole vector); delta = K^^-1 rho from the
                # fp32 factor; quad = r^T alpha0 + rho^T (alpha0 + delta) is second-order accurate in delta's error
                rho64 = torch.empty(N, dtype=torch.float64, device=dev)
                rho32 = torch.empty(N, dtype=dt, device=dev)
                self._call("dgp_dist_residual", th, _ptr(noise), _ptr(r), _ptr(alpha), _ptr(rho64), _ptr(rho32))
                self._call("dgp_dist_solve_partial", _ptr(rho32), _ptr(z))
                self._sum(z)
                delta = torch.empty(N, dtype=dt, device=dev)
                self._call("dgp_dist_alpha_partial", _ptr(z), _ptr(delta))
                self._sum(delta)
                a0 = alpha.double()
                a1 = a0 + delta.double()
                quad = (r.double() * a0[:n]).sum() + (rho64 * a1).sum()
                alpha = a1.to(dt)
            bad = stat64[1] != 0
            nll = 0.5 * quad + 0.5 * stat64[0] + 0.5 * n * math.log(2.0 * math.pi)
            out[_lib.OUT_NLL] = torch.where(bad, torch.full_like(nll, float("nan")), nll).to(dt)
            out[_lib.OUT_QUAD], out[_lib.OUT_LOGDET], out[_lib.OUT_INFO] = quad.to(dt), stat64[0].to(dt), stat64[1].to(dt)
            self.alpha, self.dnoise = alpha[:n], None
            if with_grad:
                self._inverse_products()
                tail = torch.zeros(N + _lib.OUT_LEN, dtype=dt, device=dev)  # one all-reduce for both partial results
                dnoise, dtheta = tail[:N], tail[N:]
                self._call("dgp_dist_grad_partial", th, _ptr(alpha), _ptr(dtheta), _ptr(dnoise))
                self._sum(tail)
                out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + self.ntheta] = dtheta[: self.ntheta]
                out[_lib.OUT_SUM_DR] = alpha[:n].sum()
                out[_lib.OUT_SUM_DNOISE] = dnoise[:n].sum()
                self.dnoise = dnoise[:n]
        return out

    def nll(self, theta, r, noise) -> torch.Tensor:
        return self.fit_step(theta, r, noise, with_grad=False)
