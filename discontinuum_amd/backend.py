"""Device-side exact-GP plan: torch owns memory and streams, libdgp_hip.so does the arithmetic.

``GPPlan`` is the thin host object the engine (``discontinuum_amd.engines.hip``) drives; it is the
MI355X replacement for what gpytorch's ``ExactGP`` + ``ExactMarginalLogLikelihood`` +
``DefaultPredictionStrategy`` do underneath the reference loop
(``src/discontinuum/engines/gpytorch.py:318, 350-384, 599-626``).  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

MODELS = {"loadest": _lib.MODEL_LOADEST, "rating": _lib.MODEL_RATING}


def model_id(model: str) -> int:
    """``"loadest"`` / ``"rating"`` (the fused evaluators) or ``"composite:<id>"`` (a generic model registered through
    ``dgp_composite_define``; ``gp.lowering.lower`` returns such names)."""
    if model in MODELS:
        return MODELS[model]
    if isinstance(model, str) and model.startswith("composite:") and model[10:].isdigit():
        return int(model[10:])
    raise ValueError(f"unknown model {model!r}; expected one of {sorted(MODELS)} or 'composite:<id>'")
_DTYPES = {torch.float64: _lib.F64, torch.float32: _lib.F32}


def _theta_array(theta, ntheta):
    if torch.is_tensor(theta):
        theta = theta.detach().to("cpu", torch.float64).reshape(-1).tolist()
    vals = [float(v) for v in theta]
    if len(vals) != ntheta:
        raise ValueError(f"expected {ntheta} kernel hyperparameters, got {len(vals)}")
    return (C.c_double * ntheta)(*vals)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class GPPlan:
    """Fixed (model, dtype, n, d) exact-GP problem resident on one GPU."""

    def __init__(self, model: str, n: int, d: int, dtype=torch.float64, device="cuda", lookahead=True, batch: int = 1):
        """``lookahead``: False / 0 = one stream; 1 = bulk updates beside the panel chain (use this when several
        plans share one GPU); True / 2 = also the early inverse on a third stream (best for one plan per GPU).
        ``batch`` > 1: the plan carries that many independent sites in lockstep (one launch per kernel for all of
        them); ``set_inputs`` / ``fit_step`` / ``factorize`` then take batch-major arrays -- X (batch, n, d),
        theta (batch, ntheta), r / noise (batch, n) -- and return (batch, 32), (batch, n), (batch, n)."""
        mid = model_id(model)
        if dtype not in _DTYPES:
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("discontinuum_amd requires a ROCm GPU (MI355X); there is no CPU fallback")
        self.model, self.n, self.d, self.dtype = model, int(n), int(d), dtype
        self.device = torch.device(device)
        self.ntheta = self.lib.dgp_model_ntheta(mid, self.d)
        if self.ntheta < 0:
            raise ValueError(f"model {model!r} does not support d={d}")
        self.N = int(self.lib.dgp_padded_n(self.n))
        handle = C.c_void_p()
        _lib.check(self.lib.dgp_plan_create(mid, _DTYPES[dtype], self.n, self.d, C.byref(handle)), "dgp_plan_create")
        self._h = handle
        self.batch = int(batch)
        if self.batch != 1:
            _lib.check(self.lib.dgp_plan_set_batch(self._h, self.batch), "dgp_plan_set_batch")
        nbytes = int(self.lib.dgp_plan_workspace_bytes(self._h))
        with torch.cuda.device(self.device):
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self._ws.data_ptr()) % 256
            self._ws_off = off
            _lib.check(
                self.lib.dgp_plan_set_workspace(self._h, C.c_void_p(self._ws.data_ptr() + off), nbytes),
                "dgp_plan_set_workspace",
            )
            self._pred_ws = None
        self.set_lookahead(lookahead)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                torch.cuda.synchronize(self.device)
            except Exception:  # noqa: BLE001
                pass
            self.lib.dgp_plan_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ helpers
    def set_lookahead(self, level):
        level = 2 if level is True else int(level)
        _lib.check(self.lib.dgp_plan_set_lookahead(self._h, level), "dgp_plan_set_lookahead")

    def set_option(self, key: int, value: int):
        """Plan-level option (``_lib.OPT_*``; include/dgp_hip.h ``dgp_plan_set_option``): tile-shape selectors of the
        O(n^3) stages and the float32 refinement switch."""
        _lib.check(self.lib.dgp_plan_set_option(self._h, int(key), int(value)), "dgp_plan_set_option")

    def get_option(self, key: int) -> int:
        v = C.c_int64()
        _lib.check(self.lib.dgp_plan_get_option(self._h, int(key), C.byref(v)), "dgp_plan_get_option")
        return int(v.value)

    def _check_vec(self, t, name, length=None):
        length = (self.n if length is None else length) * self.batch
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == self.dtype and t.is_contiguous() and t.numel() == length):
            raise ValueError(f"{name} must be a contiguous {self.dtype} CUDA tensor with {length} elements")

    def buffer(self, which: int, site: int = 0) -> torch.Tensor:
        """Tensor view of a plan buffer (tests / profiling); ``site``: which site's copy of a batched plan."""
        p, ld = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.dgp_plan_buffer(self._h, which, C.byref(p), C.byref(ld)), "dgp_plan_buffer")
        if not 0 <= site < self.batch:
            raise ValueError(f"site must be in 0..{self.batch - 1}")
        esz = torch.empty((), dtype=self.dtype).element_size()
        off = p.value - self._ws.data_ptr() + site * int(self.lib.dgp_plan_site_stride_bytes(self._h))
        N = self.N
        count = {_lib.BUF_XT: self.d * N, _lib.BUF_Z: N, _lib.BUF_ALPHA: N}.get(which, N * N)
        flat = self._ws[off:off + count * esz].view(self.dtype)
        if which == _lib.BUF_XT:
            return flat.view(self.d, N)
        return flat if count == N else flat.view(N, N)

    def set_site_sizes(self, sizes):
        """Ragged batch: site b uses the first ``sizes[b]`` (<= n) rows of its slots; call before ``set_inputs``."""
        vals = [int(v) for v in sizes]
        if len(vals) != self.batch:
            raise ValueError(f"expected {self.batch} site sizes")
        arr = (C.c_int64 * self.batch)(*vals)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_plan_set_site_sizes(self._h, arr, _stream()), "dgp_plan_set_site_sizes")

    # ------------------------------------------------------------------ hot path
    def set_dr_weights(self, w):
        """Two device vectors (2, n) -- (batch, 2, n) for a batched plan -- for which every following fit step also
        returns sum_i dNLL/dr_i w_k[i] in ``out[..., OUT_DR_W0 + k]`` (None clears).  The plan keeps the tensor alive."""
        if w is not None:
            shape = (2, self.n) if self.batch == 1 else (self.batch, 2, self.n)
            if not (torch.is_tensor(w) and w.is_cuda and w.dtype == self.dtype and tuple(w.shape) == shape
                    and w.is_contiguous()):
                raise ValueError(f"dr weights must be a contiguous {shape} {self.dtype} CUDA tensor")
        self._dr_w = w
        _lib.check(self.lib.dgp_plan_set_dr_weights(self._h, _ptr(w) if w is not None else None), "dgp_plan_set_dr_weights")

    def set_inputs(self, X: torch.Tensor):
        self._check_vec(X, "X", self.n * self.d)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_set_inputs(self._h, _ptr(X), _stream()), "dgp_set_inputs")
        self._X = X  # keep alive until the async pack has certainly run

    def fit_step(self, theta, r: torch.Tensor, noise: torch.Tensor):
        """-> (out[32], alpha[n], dnoise[n]) device tensors; see include/dgp_hip.h DGP_OUT_*."""
        self._check_vec(r, "r")
        self._check_vec(noise, "noise")
        th = _theta_array(theta, self.ntheta * self.batch)
        shape = (lambda k: (k,)) if self.batch == 1 else (lambda k: (self.batch, k))
        with torch.cuda.device(self.device):
            out = torch.empty(shape(_lib.OUT_LEN), dtype=self.dtype, device=self.device)
            dr = torch.empty(shape(self.n), dtype=self.dtype, device=self.device)
            dnoise = torch.empty(shape(self.n), dtype=self.dtype, device=self.device)
            _lib.check(
                self.lib.dgp_fit_step(self._h, th, _ptr(r), _ptr(noise), _ptr(out), _ptr(dr), _ptr(dnoise), _stream()),
                "dgp_fit_step",
            )
        return out, dr, dnoise

    def factorize(self, theta, r: torch.Tensor, noise: torch.Tensor):
        self._check_vec(r, "r")
        self._check_vec(noise, "noise")
        th = _theta_array(theta, self.ntheta * self.batch)
        with torch.cuda.device(self.device):
            out = torch.empty(_lib.OUT_LEN if self.batch == 1 else (self.batch, _lib.OUT_LEN), dtype=self.dtype,
                              device=self.device)
            _lib.check(self.lib.dgp_factorize(self._h, th, _ptr(r), _ptr(noise), _ptr(out), _stream()), "dgp_factorize")
        return out

    def predict(self, theta, Xs: torch.Tensor, chunk: int | None = None):
        """Latent posterior (K*^T alpha, diag(K** - K*^T K^^-1 K*)) at Xs (m, d) from the held factorisation.
        Batched plans: Xs (batch, m, d), theta (batch, ntheta) -> mean, var (batch, m); every site predicts at its own
        points from the factorisation the last ``fit_step`` / ``factorize`` left in its slice of the workspace.
        ``chunk`` = prediction points per launch sequence.  The work area is batch x 2 N x chunk elements (cross Gram and
        V = L^-1 K* per site), so the default is 16384 // batch rounded down to a multiple of 128 (at least 128): a
        batched prediction then needs no more work memory than a single site's (n = 8192 fp64: 2.1 GB)."""
        if chunk is None:
            chunk = max(128, (16384 // self.batch) // 128 * 128)
        lead = () if self.batch == 1 else (self.batch,)
        if not (torch.is_tensor(Xs) and Xs.is_cuda and Xs.dtype == self.dtype and Xs.dim() == 2 + len(lead)
                and Xs.shape[-1] == self.d and tuple(Xs.shape[:-2]) == lead):
            raise ValueError(f"Xs must be a {lead + ('m', self.d)} {self.dtype} CUDA tensor")
        th = _theta_array(theta, self.ntheta * self.batch)
        m = Xs.shape[-2]
        mean = torch.empty(lead + (m,), dtype=self.dtype, device=self.device)
        var = torch.empty(lead + (m,), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            for lo in range(0, m, chunk):
                hi = min(lo + chunk, m)
                whole = lo == 0 and hi == m
                xs = Xs[..., lo:hi, :].contiguous()
                need = int(self.lib.dgp_predict_workspace_bytes(self._h, hi - lo))
                if self._pred_ws is None or self._pred_ws.numel() < need + 256:
                    self._pred_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
                base = self._pred_ws.data_ptr()
                base += (-base) % 256
                # a chunk of a batched prediction is not contiguous inside (batch, m): stage it
                mo = mean if (whole or not lead) else torch.empty(lead + (hi - lo,), dtype=self.dtype, device=self.device)
                vo = var if (whole or not lead) else torch.empty_like(mo)
                mp = mo if (whole or lead) else mean[lo:hi]
                vp = vo if (whole or lead) else var[lo:hi]
                _lib.check(
                    self.lib.dgp_predict(self._h, th, _ptr(xs), hi - lo, C.c_void_p(base), need, _ptr(mp), _ptr(vp), _stream()),
                    "dgp_predict",
                )
                if lead and not whole:
                    mean[:, lo:hi] = mo
                    var[:, lo:hi] = vo
        return mean, var

    # ------------------------------------------------------------------ sample(): posterior covariance, factor, draws
    def _jitter_ladder(self):
        """linear_operator's ``psd_safe_cholesky`` policy: first no jitter at all, then the
        dtype's default (1e-8 fp64 / 1e-6 fp32), then 10x and 100x that; after that it raises."""
        base = 1e-8 if self.dtype == torch.float64 else 1e-6
        return (0.0, base, 10 * base, 100 * base)

    def posterior_cov(self, theta, Xs: torch.Tensor):
        """(K*^T alpha, latent posterior covariance K** - V^T V) at Xs (m, d): the covariance as an (M, M) tensor,
        M = padded m, lower triangle valid (diagonal 128-blocks complete), identity pad -- ``dgp_posterior_cov``.
        Batched plans: Xs (batch, m, d), theta (batch, ntheta) -> mean (batch, m), cov (batch, M, M), one launch sequence."""
        lead = () if self.batch == 1 else (self.batch,)
        if not (torch.is_tensor(Xs) and Xs.is_cuda and Xs.dtype == self.dtype and Xs.dim() == 2 + len(lead)
                and Xs.shape[-1] == self.d and tuple(Xs.shape[:-2]) == lead):
            raise ValueError(f"Xs must be a {lead + ('m', self.d)} {self.dtype} CUDA tensor")
        th = _theta_array(theta, self.ntheta * self.batch)
        m = Xs.shape[-2]
        M = int(self.lib.dgp_padded_n(m))
        with torch.cuda.device(self.device):
            xs = Xs.contiguous()
            need = int(self.lib.dgp_predict_workspace_bytes(self._h, m))
            if self._pred_ws is None or self._pred_ws.numel() < need + 256:
                self._pred_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            base = self._pred_ws.data_ptr()
            base += (-base) % 256
            mean = torch.empty(lead + (m,), dtype=self.dtype, device=self.device)
            cov = torch.empty(lead + (M, M), dtype=self.dtype, device=self.device)
            _lib.check(
                self.lib.dgp_posterior_cov(self._h, th, _ptr(xs), m, C.c_void_p(base), need, _ptr(mean), _ptr(cov), _stream()),
                "dgp_posterior_cov",
            )
        return mean, cov

    def psd_safe_factor(self, cov: torch.Tensor, m: int):
        """Lower Cholesky factor of the (M, M) matrix ``cov`` (layout of ``posterior_cov``; left untouched) by the
        blocked HIP potrf of an order-m plan, with ``psd_safe_cholesky``'s jitter policy: the matrix itself first, then
        + 1e-8 I, 1e-7 I, 1e-6 I (fp32: 1e-6 .. 1e-4), each attempt restarting from the kept matrix; raises after the
        last.  -> (Lbuf, jitter): Lbuf is that plan's (M, M) buffer -- zeros above the diagonal inside the diagonal
        128-blocks, identity pad, blocks above the block diagonal undefined -- valid until the next call."""
        M = int(self.lib.dgp_padded_n(m))
        if tuple(cov.shape) != (M, M) or cov.dtype != self.dtype or not cov.is_cuda or not cov.is_contiguous():
            raise ValueError(f"cov must be a contiguous ({M}, {M}) {self.dtype} CUDA tensor")
        fac = getattr(self, "_fac", None)  # the order-m plan whose potrf factors the covariance; kept between calls
        if fac is None or fac.n != m:
            self._fac = None
            self._fac = fac = GPPlan(self.model, m, self.d, dtype=self.dtype, device=self.device)
        Lbuf = fac.buffer(_lib.BUF_A)
        info = -1
        with torch.cuda.device(self.device):
            for jitter in self._jitter_ladder():
                Lbuf.copy_(cov)
                if jitter:
                    Lbuf.diagonal()[:m].add_(jitter)
                fac.stage_potrf()
                info = fac.potrf_info()
                if info == 0:
                    return Lbuf, jitter
        raise RuntimeError(f"posterior covariance not positive definite after jitter {jitter:g} (pivot {info})")

    def posterior_factor(self, theta, Xs: torch.Tensor):
        """-> (K*^T alpha, Lbuf, jitter): ``posterior_cov`` followed by ``psd_safe_factor``."""
        mean, cov = self.posterior_cov(theta, Xs)
        Lbuf, jitter = self.psd_safe_factor(cov, Xs.shape[0])
        return mean, Lbuf, jitter

    def sample_draws(self, Lbuf: torch.Tensor, m: int, mean, ndraw: int, generator=None):
        """(ndraw, m) draws mean + L z, z ~ N(0, I), through ``dgp_sample_draws`` (one MFMA launch on the factor as
        ``psd_safe_factor`` leaves it).  The normals come from torch's generator (plumbing); ``self._last_z`` keeps
        them for the tests."""
        M = int(self.lib.dgp_padded_n(m))
        Q = int(self.lib.dgp_padded_n(ndraw))
        if tuple(Lbuf.shape) != (M, M) or Lbuf.dtype != self.dtype or not Lbuf.is_contiguous():
            raise ValueError(f"Lbuf must be the contiguous ({M}, {M}) factor buffer")
        with torch.cuda.device(self.device):
            z = torch.randn(M, Q, dtype=self.dtype, device=self.device, generator=generator)
            out = torch.empty(ndraw, m, dtype=self.dtype, device=self.device)
            mp = _ptr(mean.contiguous()) if mean is not None else None
            _lib.check(self.lib.dgp_sample_draws(_DTYPES[self.dtype], _ptr(Lbuf), m, _ptr(z), ndraw, mp, _ptr(out), _stream()),
                       "dgp_sample_draws")
        self._last_z = z
        return out

    def _vjp_workspace(self, m):
        need = int(self.lib.dgp_mean_vjp_workspace_bytes(self._h, m))
        ws = getattr(self, "_vjp_ws", None)
        if ws is None or ws.numel() < need + 256:
            self._vjp_ws = ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = ws.data_ptr()
        return C.c_void_p(base + (-base) % 256), need

    def predict_mean(self, theta, Xs: torch.Tensor):
        """K(X*, X) alpha from the held factorisation (no variance work).  Batched plans: Xs (batch, m, d) -> (batch, m)."""
        th = _theta_array(theta, self.ntheta * self.batch)
        lead = () if self.batch == 1 else (self.batch,)
        if Xs.dim() != 2 + len(lead) or tuple(Xs.shape[:-2]) != lead or Xs.shape[-1] != self.d:
            raise ValueError(f"Xs must have shape {lead + ('m', self.d)}")
        m = Xs.shape[-2]
        with torch.cuda.device(self.device):
            xs = Xs.contiguous()
            work, need = self._vjp_workspace(m)
            mean = torch.empty(lead + (m,), dtype=self.dtype, device=self.device)
            _lib.check(self.lib.dgp_predict_mean(self._h, th, _ptr(xs), m, work, need, _ptr(mean), _stream()), "dgp_predict_mean")
        return mean

    def mean_vjp(self, theta, Xs: torch.Tensor, w: torch.Tensor):
        """Vector-Jacobian product of ``predict_mean``: (dtheta[P], dr[n], dnoise[n]) for upstream w[m].  Batched plans:
        Xs (batch, m, d), w (batch, m) -> dtheta (batch, P), dr (batch, n), dnoise (batch, n), one launch sequence for all sites."""
        th = _theta_array(theta, self.ntheta * self.batch)
        lead = () if self.batch == 1 else (self.batch,)
        if Xs.dim() != 2 + len(lead) or tuple(Xs.shape[:-2]) != lead or Xs.shape[-1] != self.d:
            raise ValueError(f"Xs must have shape {lead + ('m', self.d)}")
        m = Xs.shape[-2]
        self._check_vec(w, "w", m)
        with torch.cuda.device(self.device):
            xs = Xs.contiguous()
            work, need = self._vjp_workspace(m)
            width = _lib.OUT_LEN if self.batch == 1 else self.ntheta
            dtheta = torch.zeros(lead + (width,), dtype=self.dtype, device=self.device)
            dr = torch.empty(lead + (self.n,), dtype=self.dtype, device=self.device)
            dnoise = torch.empty(lead + (self.n,), dtype=self.dtype, device=self.device)
            _lib.check(
                self.lib.dgp_mean_vjp(self._h, th, _ptr(xs), m, _ptr(w), work, need, _ptr(dtheta), _ptr(dr), _ptr(dnoise), _stream()),
                "dgp_mean_vjp",
            )
        return dtheta[..., : self.ntheta], dr, dnoise

    def potrf_info(self) -> int:
        """info of the last factorisation (0 = ok, k = first non-positive pivot), synchronising."""
        off = self._info_offset()
        return int(self._ws[off:off + 4].view(torch.int32)[0].item())

    def _info_offset(self):
        # the int info slot sits right after the 16-element scalar block that follows the partials;
        # recover it from the ALPHA buffer pointer is fragile, so the library exposes it as buffer 6
        p, ld = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.dgp_plan_buffer(self._h, 6, C.byref(p), C.byref(ld)), "dgp_plan_buffer")
        return p.value - self._ws.data_ptr()

    def set_timing(self, enabled: bool):
        _lib.check(self.lib.dgp_plan_set_timing(self._h, int(bool(enabled))), "dgp_plan_set_timing")

    def get_timing(self):
        """Per-stage HIP-event milliseconds of the most recent fit step (include/dgp_hip.h DGP_TIME_*)."""
        ms = (C.c_double * _lib.TIME_COUNT)()
        _lib.check(self.lib.dgp_plan_get_timing(self._h, ms), "dgp_plan_get_timing")
        return list(ms)

    # ------------------------------------------------------------------ single stages (tests, profiling)
    def stage_gram(self, theta, noise):
        self._check_vec(noise, "noise")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_stage_gram(self._h, _theta_array(theta, self.ntheta * self.batch), _ptr(noise), _stream()), "dgp_stage_gram")

    def stage_potrf(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_stage_potrf(self._h, _stream()), "dgp_stage_potrf")

    def stage_trtri(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_stage_trtri(self._h, _stream()), "dgp_stage_trtri")

    def stage_lauum(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_stage_lauum(self._h, _stream()), "dgp_stage_lauum")

    def stage_solve(self, r):
        self._check_vec(r, "r")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dgp_stage_solve(self._h, _ptr(r), _stream()), "dgp_stage_solve")

    def stage_grad(self, theta):
        with torch.cuda.device(self.device):
            out = torch.zeros(_lib.OUT_LEN, dtype=self.dtype, device=self.device)
            _lib.check(self.lib.dgp_stage_grad(self._h, _theta_array(theta, self.ntheta), _ptr(out), _stream()), "dgp_stage_grad")
        return out[: self.ntheta]

    def cross_gram(self, theta, Xs):
        m = Xs.shape[0]
        M = int(self.lib.dgp_padded_n(m))
        with torch.cuda.device(self.device):
            work = torch.empty(self.d * M, dtype=self.dtype, device=self.device)
            Ks = torch.empty(self.N, M, dtype=self.dtype, device=self.device)
            xs = Xs.contiguous()
            _lib.check(
                self.lib.dgp_cross_gram(self._h, _theta_array(theta, self.ntheta), _ptr(xs), m, _ptr(work), _ptr(Ks), _stream()),
                "dgp_cross_gram",
            )
        return Ks[: self.n, :m]
