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
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib
from .backend import _DTYPES, _ptr, _stream, _theta_array, model_id


class DistributedFit:
    """This rank's share of ONE (model, n, d) exact-GP matrix.  Every rank constructs one with the same arguments and
    calls the same methods with the same (replicated) ``X``, ``theta``, ``r``, ``noise``."""

    def __init__(self, model: str, n: int, d: int, dtype=torch.float64, device="cuda", rank: int | None = None,
                 world: int | None = None, group_panels: int = 4, group=None, lookahead: bool = True):
        mid = model_id(model)
        if dtype not in _DTYPES:
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("discontinuum_amd requires a ROCm GPU (MI355X); there is no CPU fallback")
        initialised = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world = int(world) if world is not None else (dist.get_world_size(group) if initialised else 1)
        self.rank = int(rank) if rank is not None else (dist.get_rank(group) if initialised else 0)
        if self.world > 1 and not initialised:
            raise RuntimeError("world > 1 needs an initialised torch.distributed process group")
        self.model, self.n, self.d, self.dtype, self.device = model, int(n), int(d), dtype, torch.device(device)
        self.lookahead = bool(lookahead)
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

    def _src(self, g: int) -> int:
        owner = self._owner(g)
        return dist.get_global_rank(self.group, owner) if self.group is not None else owner

    def _payload(self, g: int, with_inverse_block: bool):
        elems = int(self.lib.dgp_dist_panel_elems(self._h, g))
        if not with_inverse_block:
            gw = 128 * self.W
            elems -= gw * gw
        return self._panels[g % 2][:elems]

    def _broadcast(self, buf, g: int):
        if self.world == 1:
            return None
        return dist.broadcast(buf, src=self._src(g), group=self.group, async_op=True)

    def _sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _call(self, name, *args):
        _lib.check(getattr(self.lib, name)(self._h, *args, _stream()), name)

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
        started BEFORE the bulk of ``consume(g)``, through ``consume``'s first phase."""
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
        with torch.cuda.device(dev):
            self._factor_and_invert(th, noise)
            stat = torch.empty(2, dtype=dt, device=dev)
            self._call("dgp_dist_status", _ptr(stat))
            stat64 = stat.double()
            if self.world > 1:
                logdet, info = stat64[:1].clone(), stat64[1:].clone()
                dist.all_reduce(logdet, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(info, op=dist.ReduceOp.MAX, group=self.group)
                stat64 = torch.cat([logdet, info])
            z = torch.empty(N, dtype=dt, device=dev)
            self._call("dgp_dist_solve_partial", _ptr(r), _ptr(z))
            self._sum(z)
            alpha = torch.empty(N, dtype=dt, device=dev)
            self._call("dgp_dist_alpha_partial", _ptr(z), _ptr(alpha))
            self._sum(alpha)
            out = torch.zeros(_lib.OUT_LEN, dtype=dt, device=dev)
            quad = (z.double() * z.double()).sum()
            bad = stat64[1] != 0
            nll = 0.5 * quad + 0.5 * stat64[0] + 0.5 * n * math.log(2.0 * math.pi)
            out[_lib.OUT_NLL] = torch.where(bad, torch.full_like(nll, float("nan")), nll).to(dt)
            out[_lib.OUT_QUAD], out[_lib.OUT_LOGDET], out[_lib.OUT_INFO] = quad.to(dt), stat64[0].to(dt), stat64[1].to(dt)
            self.alpha, self.dnoise = alpha[:n], None
            if with_grad:
                self._inverse_products()
                dtheta = torch.zeros(_lib.OUT_LEN, dtype=dt, device=dev)
                dnoise = torch.empty(N, dtype=dt, device=dev)
                self._call("dgp_dist_grad_partial", th, _ptr(alpha), _ptr(dtheta), _ptr(dnoise))
                self._sum(dtheta)
                self._sum(dnoise)
                out[_lib.OUT_DTHETA:_lib.OUT_DTHETA + self.ntheta] = dtheta[: self.ntheta]
                out[_lib.OUT_SUM_DR] = alpha[:n].sum()
                out[_lib.OUT_SUM_DNOISE] = dnoise[:n].sum()
                self.dnoise = dnoise[:n]
        return out

    def nll(self, theta, r, noise) -> torch.Tensor:
        return self.fit_step(theta, r, noise, with_grad=False)
