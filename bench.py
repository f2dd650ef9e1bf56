#!/usr/bin/env python3
"""Headline benchmark: GP fits/sec (NLL + gradient step) for the loadest-gp kernel, n=8192 d=3 fp64.

    python bench.py --gpus N --steps K --warmup W

One "fit" = one `dgp_fit_step` through the C ABI: Gram build, blocked Cholesky, L^-1, K^^-1, NLL and all
hyperparameter / residual / noise gradients for one site, inputs resident in HBM (BASELINE.json
configs[1]).  One "step" = one fit of each of `--sites-per-gpu` (default 32) INDEPENDENT
sites per GPU, carried in lockstep by ONE batched plan (`dgp_plan_set_batch`): every kernel of the step is
launched once for all of them (gridDim.z = sites) -- the north star's "independent sites / hyperparameter-sample
batches".  A single fit is bound by the sequential panel chain of its factorisation for half of its time; over
a batch that chain and the launch rate are amortised and the step becomes GEMM-bound.  `value` = fits/s over all
sites and GPUs; the latency of one site alone on the GPU (a single fit loop, `--sites-per-gpu 1` is the same
thing as the timed region) is reported next to it (`single_site`).  With N > 1 every rank owns its own sites
(global site g = rank + j N lives on rank g mod N -- sites.site_partition -- and is the synthetic site with seed g, which is
also its row in the gathered table; weak scaling, no data-path collective); the only RCCL traffic is the gather of the
(NLL, gradient) rows at the end of the timed region.  Rank 0 prints ONE JSON line.

`roofline`     dominant kernel of the step, timed live with HIP events recorded inside the library on the
               stream each kernel is launched on (last step of the timed region).  Algorithmic flops per DESIGN.md.
               `traffic` comes from the committed rocprofv3 --pmc passes and is emitted only when the sources of
               libdgp_hip.so are the ones those passes were taken on (`source_hash` in the profile JSON).
`cpu_baseline` the CPU oracle (a dense torch restatement of the reference's gpytorch math -- gpytorch itself is not
               installable here) on the host cores, rank 0 / N=1 only: the torch thread count is picked by a measured,
               bracketed two-stage sweep (8 ... all of the affinity mask at n/2, then the best and its neighbours at n;
               printed), then the median of `--cpu-steps` (3) NLL+gradient steps at n = 8192 fp64,
               plus the fp32 figure.
`configs`      (default single-GPU run only) the other BASELINE.json configurations that fit one GPU, each timed here
               with its own rooflines: C1 n~300 engine iterations (both models; warm and cold predict), the engine-level
               `model.fit` iteration at n=8192, C3 rating-gp n=16384 d=2 fp32,
               C4's per-GPU share (64 sites of n=4096 in one batched plan), C5's matrix on one GPU (n=65536 fp32), and
               inference from an n=8192 factorisation at the reference workflow's sizes (predict at m=11323, predict_grid,
               sample(n=1000): posterior covariance + order-11392 Cholesky + draws), each with its MFMA fraction.

`--model rating` runs the headline loop on the rating-gp kernel instead (d = 2); `--config 5` is the torchrun entry
point of the distributed factorisation + gradient of ONE matrix over all ranks (discontinuum_amd/dist_chol.py).
"""
from __future__ import annotations

import argparse
import contextlib
import hashlib
import io
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # MI355X dense vector/matrix peaks (MI355X_MICROARCH.md; fp64 spec)
PEAK_HBM_GBS = 8000.0
LN2 = 0.6931471805599453  # softplus(0): gpytorch's default for every Positive-constrained hyperparameter
METRIC = "GP fits/sec (NLL+grad step) at n=8192 d=3, 1/2/4/8 MI355X"


# ------------------------------------------------------------------------------------------ synthetic sites
def synth_loadest(n, d, seed):
    """Sorted centred decimal years, N(0,1) covariates, standardised target."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16.0, 16.0, n))
    cov = rng.standard_normal((n, d - 1))
    y = 0.8 * np.sin(2 * np.pi * t) + 0.5 * cov[:, 0] + 0.1 * t / 16.0 + 0.3 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std()
    return np.concatenate([t[:, None], cov], axis=1), y


def synth_rating(n, seed):
    """Time as above, stage = 1 + Beta(2,5), standardised log-discharge-like target, per-observation variances."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16.0, 16.0, n))
    s = 1.0 + rng.beta(2.0, 5.0, n)
    y = 1.6 * np.log(s - 0.5) + 0.2 * np.sin(2 * np.pi * t) * (s < 1.3) + 0.05 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std()
    return np.stack([t, s], axis=1), y, rng.uniform(1e-3, 4e-3, n)


def site(model, n, d, seed):
    """-> (X (n,d), r (n,), noise (n,), theta list) of one synthetic site at gpytorch's initial hyperparameters."""
    if model == "loadest":
        X, y = synth_loadest(n, d, seed)
        return X, y, np.full(n, 0.01), [LN2] * (2 * d + 5)
    X, y, yu = synth_rating(n, seed)
    # gate location = median stage (inside its Interval constraint), learned noise at its initial value softplus(0)+1e-4
    return X, y, yu + LN2 + 1e-4, [float(np.median(X[:, 1]))] + [LN2] * 15


def source_hash():
    """sha256 over the sources libdgp_hip.so is built from: profiles are only valid for the tree they were taken on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "discontinuum_amd", "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for f in names + [os.path.join("..", "..", "include", "dgp_hip.h")]:
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(n, d, steps):
    """The oracle (test infrastructure) as the CPU baseline.  torch's intra-op pool is set by MEASUREMENT, in two stages so
    that the minimum is BRACKETED without spending minutes on it: (1) one NLL+gradient step at n/2 (an eighth of the work)
    for 8 / 16 / 24 / 32 / 48 / 64 / 96 / 128 / all threads of the affinity mask; (2) at the full n, the best count of (1)
    and its two neighbours in that list, one step each.  The fastest is kept and the reported value is the median of
    `steps` steps at it (its sweep step counts as one of them)."""
    from oracle import gp_oracle as orc

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1

    def problem(nn, dt):
        X, y = synth_loadest(nn, d, 0)
        return (torch.tensor(X, dtype=dt), torch.tensor(y, dtype=dt), torch.full((nn,), 0.01, dtype=dt),
                torch.full((orc.loadest_ntheta(d),), LN2, dtype=dt))

    def step(prob):
        X, y, noise, theta = prob
        t0 = time.perf_counter()
        orc.nll_data_and_grads("loadest", X, y, noise, theta)
        return time.perf_counter() - t0

    ns = min(n, 8192)
    candidates = sorted({c for c in (8, 16, 24, 32, 48, 64, 96, 128, avail) if 1 <= c <= avail} or {avail})
    half = problem(max(256, ns // 2), torch.float64)
    torch.set_num_threads(candidates[min(3, len(candidates) - 1)])
    step(half)  # warm-up (allocator, MKL / OpenBLAS thread pools)
    coarse = {}
    for c in candidates:
        torch.set_num_threads(c)
        coarse[c] = step(half)
    k = candidates.index(min(coarse, key=coarse.get))
    finalists = candidates[max(0, k - 1):k + 2]
    p64 = problem(ns, torch.float64)
    sweep = {}
    for c in finalists:
        torch.set_num_threads(c)
        sweep[c] = step(p64)
    cores = min(sweep, key=sweep.get)
    interior = 0 < candidates.index(cores) < len(candidates) - 1 or len(candidates) == 1
    torch.set_num_threads(cores)
    t64 = [sweep[cores]] + [step(p64) for _ in range(max(0, steps - 1))]
    p32 = problem(ns, torch.float32)
    t32 = [step(p32) for _ in range(2)]
    med = statistics.median(t64)
    scale = (ns / n) ** 3
    sample = (f"median of {len(t64)} NLL+grad steps of oracle/gp_oracle.py (torch CPU fp64 dense, autograd) at n={ns} d={d}: "
              f"{med:.2f} s (min {min(t64):.2f}, max {max(t64):.2f}); {cores} torch threads of {avail} in the affinity mask "
              f"(os.cpu_count()={os.cpu_count()}), chosen by a two-stage sweep {{threads: s per step}}: at n={half[0].shape[0]} "
              f"{({c: round(v, 2) for c, v in coarse.items()})}, then at n={ns} {({c: round(v, 2) for c, v in sweep.items()})}; "
              f"the minimum is {'interior to' if interior else 'at the EDGE of'} the swept counts")
    if ns != n:
        sample += f"; value scaled to n={n} by (n_s/n)^3"
    return {"value": scale / med, "unit": "fits/s", "cores": cores, "kind": "port", "sample": sample,
            "affinity_cores": avail, "steps_s": [round(t, 3) for t in t64],
            "thread_sweep_s": {str(c): round(v, 3) for c, v in sweep.items()},
            "thread_sweep_half_n_s": {str(c): round(v, 3) for c, v in coarse.items()}, "sweep_minimum_interior": interior,
            "fp32": {"value": scale / statistics.median(t32), "unit": "fits/s", "steps_s": [round(t, 3) for t in t32],
                     "note": "same restatement in float32 (the reference's dtype, engines/gpytorch.py:221-222); first step "
                             "includes the float32 warm-up"}}


# ------------------------------------------------------------------------------------------ device-side measurement
def stage_report(plan, S, dtype_name, level, lib):
    """Per-stage HIP-event times of the plan's most recent fit step -> stages, dominant kernel, roofline numbers."""
    N = plan.N
    ms = plan.get_timing()
    peak = PEAK_TFLOPS[dtype_name]
    stages = {
        "syrk_kernel": {"flops": ms[lib.TIME_SYRK_FLOP], "ms": ms[lib.TIME_SYRK_SUM], "launches": int(ms[lib.TIME_SYRK_N])},
        "trtri_level_kernel": {"flops": S * N ** 3 / 3.0, "ms": ms[lib.TIME_TRTRI], "launches": None},
        "lauum_kernel": {"flops": S * N ** 3 / 3.0, "ms": ms[lib.TIME_LAUUM], "launches": 1},
    }
    if level == 2 and S == 1 and 16 <= N // 128 <= 80:  # the early inverse applies (dgp_api.hip::early_applies): most of
        stages["trtri_level_kernel"]["flops"] = None     # it ran under the factorisation, the stage time is the remainder
    for v in stages.values():
        v["tflops"] = v["flops"] / (v["ms"] * 1e-3) / 1e12 if (v["flops"] and v["ms"] > 0) else None
    # the dominant KERNEL: trtri is a stage of a dozen launches of several kernel instantiations, none of which
    # outweighs the single lauum launch or the bulk syrk launches (profiles/*_kernel_stats.csv)
    dom = max(("syrk_kernel", "lauum_kernel"), key=lambda k: stages[k]["ms"])
    esz = 8 if dtype_name == "f64" else 4
    tri_bytes = N * (N + 64) / 2 * esz * S  # lower-triangle 64 x 64 tiles of all sites of the launch
    gram_bytes = tri_bytes + plan.n * plan.d * esz * S
    hbm = lambda b, t: b / (t * 1e-3) / 1e9 if t > 0 else None  # noqa: E731
    entries = N * (N + 64) / 2 * S
    slots = VALU_SLOTS.get((plan.model, plan.d, dtype_name), (None, None))
    return {
        "dominant": dom, "achieved": stages[dom]["tflops"], "peak": peak,
        "frac": stages[dom]["tflops"] / peak if stages[dom]["tflops"] else None,
        "launches": stages[dom]["launches"], "ms": stages[dom]["ms"],
        "stages_tflops": {k: v["tflops"] for k, v in stages.items()},
        "stages_ms": {"gram": ms[lib.TIME_GRAM], "potrf_wall": ms[lib.TIME_POTRF], "syrk_sum": ms[lib.TIME_SYRK_SUM],
                      "trtri": ms[lib.TIME_TRTRI], "lauum": ms[lib.TIME_LAUUM], "solve": ms[lib.TIME_SOLVE],
                      "grad": ms[lib.TIME_GRAD]},
        "potrf_stage_tflops": S * N ** 3 / 3.0 / (ms[lib.TIME_POTRF] * 1e-3) / 1e12 if ms[lib.TIME_POTRF] > 0 else None,
        "gram_hbm": {"bound": "hbm", "achieved": hbm(gram_bytes, ms[lib.TIME_GRAM]), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "bytes": gram_bytes, "kernel": "gram_sym_kernel (writes the lower-triangle tiles)",
                     "valu_frac": valu_bound(entries, slots[0], ms[lib.TIME_GRAM]), "valu_slots_per_entry": slots[0]},
        "gram_grad_hbm": {"bound": "hbm", "achieved": hbm(tri_bytes, ms[lib.TIME_GRAD]), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "bytes": tri_bytes, "kernel": "gram_grad_kernel + reduction (reads K^^-1 once)",
                          "valu_frac": valu_bound(entries, slots[1], ms[lib.TIME_GRAD]), "valu_slots_per_entry": slots[1]},
    }


def clock_probe(lib, dev, load, load_s, seconds=0.4, nwg=16):
    """Shader clock (MHz) the chip holds while `load()` -- one asynchronous enqueue of about `load_s` seconds of GPU work
    on torch's current stream -- runs back to back: `dgp_debug_clock_probe` keeps `nwg` one-wave workgroups resident on
    one of the library's internal streams for `seconds` and stamps s_memtime / s_memrealtime at both ends (MI355X_MICROARCH.md "DVFS give-back",
    item 6: clock = d s_memtime / d s_memrealtime x 100 MHz, median over workgroups).  No product kernel carries a stamp and
    the probe never runs inside a timed region.  -> {"mhz": median, "min", "max", "workgroups"} or None."""
    import ctypes as C
    import math

    try:
        out = torch.zeros(2 * nwg, dtype=torch.int64, device=dev)
        reps = max(2, int(math.ceil(1.3 * seconds / max(load_s, 1e-4))))
        load()  # the chip is already warm from the timed region; one more enqueue so that the probe starts under load
        # (the probe goes to one of the library's internal streams of the current stream: a stream of its own would take a
        # hardware queue away from the plans measured after it)
        rc = lib.dgp_debug_clock_probe(C.c_void_p(out.data_ptr()), nwg, float(seconds), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            return None
        for _ in range(reps):
            load()
        torch.cuda.synchronize(dev)
        v = out.cpu().reshape(nwg, 2).double()
        mhz = sorted((v[:, 0] / v[:, 1] * 100.0).tolist())
        if not mhz or not all(math.isfinite(x) and x > 0 for x in mhz):
            return None
        return {"mhz": round(mhz[len(mhz) // 2], 1), "min": round(mhz[0], 1), "max": round(mhz[-1], 1), "workgroups": nwg,
                "window_s": seconds, "source": "dgp_debug_clock_probe: d s_memtime / d s_memrealtime x 100 MHz while the "
                                               "load runs back to back (median over workgroups)"}
    except Exception:  # noqa: BLE001
        return None


NOMINAL_MHZ = 2400.0  # the clock the datasheet peaks are quoted at (MI355X_MICROARCH.md)
# VALU issue slots per matrix entry of the Gram kernels (ISA counts of the shipped code objects):
# (model, dtype) -> (gram_sym, gram_grad).  One wave-instruction occupies a SIMD's VALU for 4 cycles (64 lanes / 16 per cycle).
VALU_SLOTS = {("loadest", 3, "f64"): (109, 142)}


def valu_bound(entries, slots, ms, clock_mhz=NOMINAL_MHZ):
    """Fraction of the VALU ISSUE bound a Gram kernel reaches: entries x slots / 64 lanes wave-instructions, 4 cycles each,
    over 1024 SIMDs at `clock_mhz` -- the roofline that binds these kernels in fp64 (they sit at 0.2-0.3 of HBM)."""
    if not slots or not ms or ms <= 0:
        return None
    bound_s = entries * slots / 64.0 * 4.0 / (1024.0 * clock_mhz * 1e6)
    return bound_s / (ms * 1e-3)


def make_plan(model, n, d, dt, dev, S, level, seed0=0, seed_stride=1):
    """A (batched) plan with S synthetic sites resident in HBM -> (plan, theta, r, noise) ready for fit_step.  Local site j
    is the synthetic site with seed seed0 + j seed_stride (multi-GPU runs: rank + j world, the global index of the site in
    sites.site_partition's round-robin order, which is also the row it gets in the gathered table)."""
    from discontinuum_amd.backend import GPPlan

    Xs, rs, nz, th = [], [], [], []
    for sidx in range(S):
        X, r, noise, theta = site(model, n, d, seed0 + sidx * seed_stride)
        Xs.append(torch.tensor(X, dtype=dt))
        rs.append(torch.tensor(r, dtype=dt))
        nz.append(torch.tensor(noise, dtype=dt))
        th += theta
    plan = GPPlan(model, n, d, dtype=dt, device=dev, lookahead=level, batch=S)
    if S > 1:
        plan.set_inputs(torch.stack(Xs).to(dev).contiguous())
        return plan, th, torch.stack(rs).to(dev).contiguous(), torch.stack(nz).to(dev).contiguous()
    plan.set_inputs(Xs[0].to(dev).contiguous())
    return plan, th, rs[0].to(dev).contiguous(), nz[0].to(dev).contiguous()


def time_config(name, model, n, d, dtype_name, S, steps, warmup, dev, lib, level=None, probe=True):
    """One BASELINE configuration on this GPU: `steps` timed fit steps of S sites in one plan."""
    dt = torch.float64 if dtype_name == "f64" else torch.float32
    level = (1 if S > 1 else 2) if level is None else level
    plan, th, r, noise = make_plan(model, n, d, dt, dev, S, level)
    plan.set_timing(True)
    for _ in range(warmup):
        out = plan.fit_step(th, r, noise)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = plan.fit_step(th, r, noise)[0]
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / steps
    host = out.reshape(S, -1).cpu().double()
    ok = bool((host[:, lib.OUT_INFO] == 0).all()) and bool(torch.isfinite(host[:, lib.OUT_NLL]).all())
    rep = stage_report(plan, S, dtype_name, level, lib)
    N = plan.N
    parity = None
    if S > 1:  # site 0 of the batch against a single-site plan on the same inputs (batch_parity)
        one, th1, r1, nz1 = make_plan(model, n, d, dt, dev, 1, 2)
        parity = batch_parity(plan.fit_step(th, r, noise), one.fit_step(th1, r1, nz1), plan.ntheta, dtype_name, lib)
        ok = ok and parity["ok"]
        del one
    clk = None
    if probe:  # shader clock while the dominant kernel's stage runs back to back (after the timed steps)
        if rep["dominant"] == "lauum_kernel":
            clk = clock_probe(lib.load(), dev, plan.stage_lauum, max(1e-4, rep["stages_ms"]["lauum"] * 1e-3))
        else:
            clk = clock_probe(lib.load(), dev, lambda: plan.fit_step(th, r, noise), dtm)
    res = {"workload": name, "kernel": f"{model}-gp", "n": n, "d": d, "dtype": dtype_name, "sites_in_plan": S, "steps": steps,
           "ms_per_step": dtm * 1e3, "fits_per_s": S / dtm, "tflops": S * float(N) ** 3 / dtm / 1e12,
           "frac_of_peak": S * float(N) ** 3 / dtm / 1e12 / PEAK_TFLOPS[dtype_name], "ok": ok,
           "nll_site0": float(host[0, lib.OUT_NLL]), "hbm_gib": plan._ws.numel() / 2 ** 30, "lookahead": level, "parity": parity,
           "roofline": {"bound": "mfma", "kernel": rep["dominant"], "achieved": rep["achieved"], "peak": rep["peak"],
                        "unit": "TFLOP/s", "frac": rep["frac"], "ms_per_step": rep["ms"],
                        "clock_mhz": clk["mhz"] if clk else None,
                        "frac_at_clock": (rep["achieved"] / (rep["peak"] * clk["mhz"] / NOMINAL_MHZ)) if (clk and rep["achieved"]) else None,
                        "clock_probed_under": ("the dominant kernel's stage (lauum) back to back" if rep["dominant"] == "lauum_kernel"
                                               else "whole fit steps back to back") if clk else None},
           "stages_ms": rep["stages_ms"], "stages_tflops": rep["stages_tflops"],
           "gram_hbm_gbs": rep["gram_hbm"]["achieved"], "gram_grad_hbm_gbs": rep["gram_grad_hbm"]["achieved"]}
    del plan
    torch.cuda.empty_cache()
    return res


def engine_site(family, n, seed=0):
    """One synthetic sampling record as the ENGINE takes it (labelled arrays in data space, like the reference's fixtures
    tests/test_loadest_gp.py:12-28) -> (Model class, (covariates, target), fit keyword arguments)."""
    from discontinuum_amd.xr_compat import DataArray, Dataset

    rng = np.random.default_rng(seed)
    if family == "loadest":
        from discontinuum_amd.loadest_gp import LoadestGP as Model

        t = (np.datetime64("1990-01-01") + np.sort(rng.choice(365 * 30, n, replace=False)).astype("timedelta64[D]")).astype("datetime64[ns]")
        flow = np.exp(rng.standard_normal(n)) * 10
        conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
        args = (Dataset({"flow": ("time", flow)}, coords={"time": t}), DataArray(conc, dims=("time",), coords={"time": t}, name="c"))
        kw = {}
    else:
        from discontinuum_amd.rating_gp import RatingGP as Model

        t = (np.datetime64("2005-01-01") + np.sort(rng.choice(365 * 15, n, replace=False)).astype("timedelta64[D]")).astype("datetime64[ns]")
        stage = 1.0 + 3.0 * rng.beta(2, 5, n)
        q = np.exp(1.6 * np.log(stage) + 0.05 * rng.standard_normal(n))
        args = (Dataset({"stage": ("time", stage)}, coords={"time": t}), DataArray(q, dims=("time",), coords={"time": t}, name="q"))
        kw = {"target_unc": DataArray(np.full(n, 1.05), dims=("time",), coords={"time": t}, name="q_unc")}
    return Model, args, kw


def train_many(family, n, sites, iters, barrier=None, distributed=False):
    """The product-level many-site path (discontinuum_amd/multisite_fit.py): TRAIN `sites` independent sites of n observations for
    `iters` iterations -- one batched fit step + the vectorised host algebra + the reference's per-site optimiser semantics
    (engines/gpytorch.py:346-451) per iteration -- through `fit_many` (one process) or `fit_many_distributed` (site i on rank
    i mod world, ONE gather of the fitted parameters at the end of the fit).  `fit_many` itself reports how long its loop took;
    the rest of the call is set-up (model objects, pipelines, plan creation, upload) and hand-back.
    -> dict with sites x iterations / s (whole call) and ms per iteration (the training loop alone: `multisite_fit.LAST_TIMING`)."""
    from discontinuum_amd import multisite_fit

    def run(k):
        models, data = [], []
        for i in range(sites):
            Model, args, kw = engine_site(family, n, seed=i)
            models.append(Model())
            data.append(args + ((kw["target_unc"],) if kw else ()))
        if barrier:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stderr(io.StringIO()):
            if distributed:
                objs, _ = multisite_fit.fit_many_distributed(models, data, iterations=k, load="rank0")
            else:
                objs = multisite_fit.fit_many(models, data, iterations=k)
        torch.cuda.synchronize()
        if barrier:
            barrier()
        return time.perf_counter() - t0, objs

    run(min(2, iters))  # warm-up: code objects, allocator
    t_full, objs = run(iters)
    tm = dict(multisite_fit.LAST_TIMING)  # this rank's fit_many: seconds before the first iteration / inside the loop
    per_it = tm["loop_s"] / max(1, tm["iterations"])
    local_sites = tm["sites"]
    return {"workload": f"{sites} {family}-gp sites of n={n} (d=2, fp64) trained for {iters} iterations by "
                        f"{'fit_many_distributed' if distributed else 'fit_many'} (labelled arrays in, fitted models out)",
            "sites": sites, "n": n, "iterations": iters, "seconds": t_full, "site_iterations_per_s": sites * iters / t_full,
            "ms_per_iteration": per_it * 1e3, "site_iterations_per_s_steady": local_sites / per_it if per_it > 0 else None,
            "setup_s": t_full - tm["loop_s"], "closed_form_host_algebra": tm.get("closed_form"),
            "timing_note": "ms_per_iteration = this rank's training loop / iterations (its share of the sites); seconds = the whole call "
                           "incl. model objects, pipelines, plan creation, upload and -- distributed -- the final gather",
            "ok": bool(torch.isfinite(objs).all()), "objective_mean": float(objs.mean())}


def engine_iteration(family, n, iters):
    """BASELINE config 1 through the engine surface: ms per training iteration of `model.fit` (host loop + device step)."""
    Model, args, kw = engine_site(family, n)
    m = Model()
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):  # the progress bar
        m.fit(*args, iterations=5, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.fit(*args, iterations=iters, **kw)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / iters
        t1 = time.perf_counter()
        m.predict(args[0])
        torch.cuda.synchronize()
        tp = time.perf_counter() - t1  # the FIRST predict of the process: plan creation, code-object loading
        reps = 5 if n <= 1000 else 2
        t2 = time.perf_counter()
        for _ in range(reps):
            m.predict(args[0])
        torch.cuda.synchronize()
        tw = (time.perf_counter() - t2) / reps
    return {"workload": f"{family}-gp site, n={n} d=2 fp64, `model.fit` through the engine surface", "n": n,
            "iterations": iters, "ms_per_iteration": dtm * 1e3, "fits_per_s": 1.0 / dtm, "predict_ms_cold": tp * 1e3,
            "predict_ms": tw * 1e3, "predict_note": "predict at the n training points; cold = first call of the process "
                                                    "(plan creation + code objects), predict_ms = warm mean"}


def split_batch_config(model, n, d, dtype_name, halves, steps, warmup, dev, lib):
    """The same sites as several batched plans on separate streams (one plan's sequential panel chain runs under the
    others' bulk updates): ms per sweep of ALL sites and fits/s.  Measured LAST in a run: every extra stream takes one of
    the process's few hardware queues from whatever is measured after it."""
    dt = torch.float64 if dtype_name == "f64" else torch.float32
    plans, seed = [], 0
    for S in halves:
        plans.append(make_plan(model, n, d, dt, dev, S, 1, seed0=seed))
        seed += S
    streams = [torch.cuda.Stream(device=dev) for _ in halves]

    def sweep():
        for (plan, th, r, noise), st in zip(plans, streams):
            with torch.cuda.stream(st):
                out = plan.fit_step(th, r, noise)[0]
        return out

    for _ in range(warmup):
        sweep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = sweep()
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / steps
    ok = bool(torch.isfinite(out.reshape(-1, lib.OUT_LEN)[:, lib.OUT_NLL]).all())
    del plans
    torch.cuda.empty_cache()
    return {"plans": list(halves), "ms_per_step": dtm * 1e3, "fits_per_s": sum(halves) / dtm, "ok": ok,
            "note": "several batched plans driven from separate streams, each with its own internal stream set"}


def inference_configs(dev, lib, quick):
    """Inference from ONE n = 8192 loadest factorisation (fp64) at the reference workflow's sizes, through the plan's
    C-ABI entry points, each piece with its MFMA fraction (flops counted as in DESIGN.md section 4):
      predict        m = 11323 (a daily grid over a 31-year record: engines/gpytorch.py:460-501, 599-626): K* build,
                     V = L^-1 K* (n^2 m flop on MFMA), column reductions
      predict_grid   m = 18 x 384 = 6912 (engines/gpytorch.py:503-549: 18 covariate values x 12 steps a year over the
                     synthetic record's 32 years)
      sample         n_draw = 1000 at m = 11323 (engines/gpytorch.py:551-593): posterior covariance K** - V^T V
                     (n^2 m + m^2 n flop), its Cholesky factor (the same blocked potrf, order 11392: m^3 / 3) and the
                     draws mean + L z (m^2 n_draw flop)."""
    from discontinuum_amd.backend import GPPlan

    n, d, dt = 8192, 3, torch.float64
    peak = PEAK_TFLOPS["f64"]
    X, r, noise, theta = site("loadest", n, d, 0)
    plan = GPPlan("loadest", n, d, dtype=dt, device=dev)
    plan.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
    plan.factorize(theta, torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev))
    rng = np.random.default_rng(1)

    def points(m, grid=None):
        if grid:
            t = np.repeat(np.linspace(-16.0, 16.0, grid[0]), grid[1])
            c1 = np.tile(np.linspace(-2.5, 2.5, grid[1]), grid[0])
            return torch.tensor(np.stack([t, c1, np.zeros_like(t)], axis=1), dtype=dt, device=dev).contiguous()
        return torch.tensor(np.concatenate([np.linspace(-16.0, 16.0, m)[:, None], rng.standard_normal((m, d - 1))], axis=1),
                            dtype=dt, device=dev).contiguous()

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, res

    reps = 2 if quick else 5
    N = plan.N
    out = {"factorisation": f"loadest n={n} d={d} fp64, one site (dgp_factorize)", "peak_tflops": peak}
    for key, m, grid in (("predict_m11323", 11323, None), ("predict_grid_18x384", 6912, (384, 18))):
        Xs = points(m, grid)
        ms, (mean, var) = timed(lambda: plan.predict(theta, Xs), reps)
        fl = float(N) ** 2 * m
        out[key] = {"m": m, "ms": ms, "flops": fl, "tflops": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / peak,
                    "ok": bool(torch.isfinite(mean).all() and torch.isfinite(var).all()), "entry": "dgp_predict"}
    m, ndraw = 11323, 1000
    Xs = points(m)
    M = (m + 127) // 128 * 128
    ms_cov, (mean, cov) = timed(lambda: plan.posterior_cov(theta, Xs), max(1, reps // 2))
    ms_fac, (Lbuf, jitter) = timed(lambda: plan.psd_safe_factor(cov, m), max(1, reps // 2))
    ms_draw, draws = timed(lambda: plan.sample_draws(Lbuf, m, mean, ndraw), reps)
    f_cov, f_fac, f_draw = float(N) ** 2 * M + float(M) ** 2 * N, float(M) ** 3 / 3.0, float(M) ** 2 * ndraw
    tf = lambda f, t: f / (t * 1e-3) / 1e12  # noqa: E731
    out["sample_1000_m11323"] = {
        "m": m, "draws": ndraw, "ms": ms_cov + ms_fac + ms_draw, "jitter": jitter, "ok": bool(torch.isfinite(draws).all()),
        "posterior_cov": {"ms": ms_cov, "flops": f_cov, "tflops": tf(f_cov, ms_cov), "frac": tf(f_cov, ms_cov) / peak,
                          "entry": "dgp_posterior_cov"},
        "factor_order_11392": {"ms": ms_fac, "flops": f_fac, "tflops": tf(f_fac, ms_fac), "frac": tf(f_fac, ms_fac) / peak,
                               "entry": "dgp_stage_potrf (incl. the copy of the covariance and the info read-back)"},
        "draws": {"ms": ms_draw, "flops": f_draw, "tflops": tf(f_draw, ms_draw), "frac": tf(f_draw, ms_draw) / peak,
                  "entry": "dgp_sample_draws (incl. torch.randn of the M x 1024 normals)"}}
    del plan, cov, Lbuf, draws
    torch.cuda.empty_cache()
    return out


def run_configs(dev, lib, quick):
    """The BASELINE.json configurations other than the headline that fit one GPU, timed in this run."""
    out = {}
    out["C1_loadest_n300_engine"] = engine_iteration("loadest", 300, 50 if quick else 200)
    out["C1_rating_n300_engine"] = engine_iteration("rating", 300, 50 if quick else 200)
    out["C2_loadest_n8192_engine"] = engine_iteration("loadest", 8192, 4 if quick else 20)
    out["C3_rating_n16384_f32"] = time_config("rating-gp kernel, n=16384 d=2 fp32 exact GP, one site", "rating", 16384, 2,
                                              "f32", 1, 3 if quick else 8, 2, dev, lib)
    # the headline batch in the reference's ONLY dtype (engines/gpytorch.py:221-222): not a BASELINE config (those fix fp64 for
    # config 2), reported because a user of the reference would run exactly this
    out["C2_batch_32x8192_f32"] = time_config("32 independent loadest sites of n=8192 d=3 fp32 in one batched plan (the headline batch in the "
                                              "reference's dtype; with the fp64 refinement of alpha)", "loadest", 8192, 3, "f32", 32,
                                              3 if quick else 6, 2, dev, lib, probe=False)
    out["C4_share_64x4096_f64"] = time_config("64 independent loadest sites of n=4096 d=3 fp64 in one batched plan "
                                              "(BASELINE config 4's per-GPU share of 512 sites / 8 GPUs)", "loadest",
                                              4096, 3, "f64", 64, 3 if quick else 8, 2, dev, lib)
    out["C5_matrix_n65536_f32_one_gpu"] = time_config("single n=65536 d=3 fp32 matrix, whole fit step (factor + inverse + "
                                                      "gradient) on ONE GPU", "loadest", 65536, 3, "f32", 1, 2, 1, dev, lib,
                                                      probe=False)
    out["inference_from_n8192"] = inference_configs(dev, lib, quick)
    # the product-level many-site path: whole training runs through fit_many (host algebra vmapped over sites + per-site
    # Adam / plateau / clipping around ONE batched fit step per iteration)
    out["fit_many_64x4096"] = train_many("loadest", 4096, 64, 6 if quick else 20)
    out["fit_many_256x300"] = train_many("loadest", 300, 256, 20 if quick else 100)
    # last (it creates streams): config 4's share as two plans of 32 -- at this size a second plan hides part of the
    # first one's panel chain (scripts/c4_experiments.py: 82.5 -> 79.9 ms)
    out["C4_share_64x4096_f64"]["as_two_plans_of_32"] = split_batch_config("loadest", 4096, 3, "f64", (32, 32), 3 if quick else 8, 2,
                                                                           dev, lib)
    return out


# ------------------------------------------------------------------------------------------ main
_SAMPLER_SRC = r"""
import json, subprocess, sys, time
while True:
    t = time.time()
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        c = next(iter(json.loads(r[r.index("{"):]).values()))
        sclk = int("".join(ch for ch in c.get("sclk clock speed:", "") if ch.isdigit()) or 0)
        pw = [float(v) for k, v in c.items() if "Power" in k]
        sys.stdout.write(json.dumps({"t": t, "sclk": sclk, "power": pw[0] if pw else None}) + "\n")
        sys.stdout.flush()
    except Exception:
        pass
    time.sleep(0.15)
"""


class GpuStateSampler:
    """Shader clock and package power of GPU 0 (rocm-smi) while the timed region runs: boxes of the pool differ by up to 9 % in
    every kernel's rate, and this is the only way the line can say which kind it ran on.  The sampling CHILD is started before
    this process touches the GPU (nothing is exec'ed from a process that has initialised it); failures yield null."""

    def __init__(self):
        import subprocess

        import atexit

        try:
            self.proc = subprocess.Popen([sys.executable, "-c", _SAMPLER_SRC], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            atexit.register(lambda p=self.proc: p.poll() is None and p.kill())  # never outlives the bench, whatever ends it
        except Exception:  # noqa: BLE001
            self.proc = None

    def window(self, t_start, t_end):
        if self.proc is None:
            return None
        try:
            self.proc.terminate()
            lines = self.proc.communicate(timeout=10)[0].splitlines()
        except Exception:  # noqa: BLE001
            return None
        rows = [json.loads(l) for l in lines if l.startswith("{")]
        rows = [r for r in rows if t_start <= r["t"] <= t_end and r["sclk"] > 0]
        if not rows:
            return None
        clk = [r["sclk"] for r in rows]
        pw = [r["power"] for r in rows if r["power"] is not None]
        return {"samples": len(rows), "sclk_mhz": {"min": min(clk), "mean": round(sum(clk) / len(clk), 1), "max": max(clk)},
                "power_w": ({"min": min(pw), "mean": round(sum(pw) / len(pw), 1), "max": max(pw)} if pw else None),
                "source": "rocm-smi --showclocks --showpower, every ~0.25 s during the timed region"}


# |batched - single| / scale bounds of `config.parity`.  fp64: both plans run the same k-ordered fma chains per element, the
# observed difference is 0 or a few ulp; 1e-11 / 1e-9 are SURVEY 8d's oracle tolerances divided by 10.  fp32: both plans
# refine alpha against the fp64-evaluated matrix, what differs is the rounding of two differently grouped factorisations:
# tests/test_gpu_fp32.py's batched-vs-single bounds.
PARITY_TOL = {"f64": {"nll_rel": 1e-11, "grad_rel": 1e-9, "alpha_rel": 1e-9, "dnoise_rel": 1e-9},
              "f32": {"nll_rel": 2e-5, "grad_rel": 5e-4, "alpha_rel": 5e-5, "dnoise_rel": 5e-4}}


def batch_parity(batched, single, ntheta, dtype_name, lib):
    """Site 0 of a batched plan's fit step against a single-site plan's on the same inputs -> relative differences of the
    NLL, the hyperparameter gradient, alpha = dNLL/dr and dNLL/dnoise (max-norm over max-norm), with the bounds and `ok`."""
    (ob, ab, nb), (o1, a1, n1) = batched, single
    ob, o1 = ob[0].cpu().double(), o1.cpu().double()
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-300))  # noqa: E731
    g = slice(lib.OUT_DTHETA, lib.OUT_DTHETA + ntheta)
    res = {"nll_rel": abs(float(ob[lib.OUT_NLL] - o1[lib.OUT_NLL])) / abs(float(o1[lib.OUT_NLL])),
           "grad_rel": rel(ob[g], o1[g]), "alpha_rel": rel(ab[0].cpu().double(), a1.cpu().double()),
           "dnoise_rel": rel(nb[0].cpu().double(), n1.cpu().double()),
           "info": [int(ob[lib.OUT_INFO]), int(o1[lib.OUT_INFO])]}
    tol = PARITY_TOL[dtype_name]
    res["ok"] = bool(res["info"] == [0, 0] and all(res[k] <= tol[k] for k in tol))
    res["tol"] = tol
    res["against"] = "a single-site plan (lookahead 2) on site 0's inputs, same seed; after the timed region"
    return res


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)
    as a CHILD process, hand its stdout through unchanged (rank 0's one JSON line) and return its exit code.  Called before
    anything in this process has touched the GPU (no torch.cuda call, no plan, no sampler); nothing is exec'ed.  A child
    that dies leaves its exit code and the tail of its stderr on this process's stderr."""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    with tempfile.TemporaryFile(mode="w+") as err:
        run = subprocess.run(cmd, env=env, stderr=err)
        err.seek(0)
        text = err.read()
    if run.returncode != 0:
        sys.stderr.write(text[-6000:])
        sys.stderr.write(f"\nbench.py: the {n_ranks}-rank child ({' '.join(cmd[1:4])} ...) exited with code {run.returncode}\n")
    else:
        sys.stderr.write(text[-2000:])
    return run.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--size", dest="n", type=int, default=8192,
                    help="observations per site (--size: torch.distributed.run rejects --n as ambiguous with its own options)")
    ap.add_argument("--d", type=int, default=None, help="design-matrix columns (default 3 for loadest, 2 for rating)")
    ap.add_argument("--model", choices=["loadest", "rating"], default="loadest")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps of the CPU baseline at the chosen thread count")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (C1, C3, C4 share, C5 on one GPU)")
    ap.add_argument("--quick-configs", action="store_true", help="fewer steps per config (tests)")
    ap.add_argument("--no-lookahead", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the shader-clock probe runs after the timed region")
    ap.add_argument("--sites-per-gpu", type=int, default=32,
                    help="independent sites carried in lockstep by one batched plan per GPU (1.5 GiB of HBM each at "
                         "n = 8192 fp64; 8 -> 104.5, 16 -> 106.7, 32 -> 107.6 fits/s)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 5],
                    help="2: the headline loop (default); 5: ONE n x n matrix factored and differentiated across all "
                         "ranks (dist_chol.DistributedFit.fit_step; --size 65536 --dtype f32 is BASELINE config 5)")
    ap.add_argument("--group-panels", type=int, default=0,
                    help="--config 5: 128-wide panels per column group of the block-cyclic layout (0: DistributedFit's default)")
    ap.add_argument("--train", type=int, default=0, metavar="K",
                    help="after the step metric: TRAIN world x sites-per-gpu engine-level sites of --size observations for K "
                         "iterations through multisite_fit.fit_many_distributed (each rank its share, one gather at the end); "
                         "reported as `train` (sites x iterations / s) next to the step metric")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the per-kernel timing loop of the roofline object (the command profiled with "
                         "rocprofv3 --kernel-trace --stats for profiles/)")
    args = ap.parse_args()
    model = args.model
    d = args.d if args.d is not None else (3 if model == "loadest" else 2)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks and never touches the GPU
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sampler = GpuStateSampler() if (rank == 0 and args.config == 2 and not args.roofline_only) else None
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launched by something else with a different rank count")
    dist = None
    # DGP_DIST_FORCE_COLLECTIVES=1: a ONE-rank run still creates its communicator and issues every collective (a
    # world-1 RCCL communicator is legal) -- the only way the RCCL entry points can execute on a one-GPU box
    force = os.environ.get("DGP_DIST_FORCE_COLLECTIVES", "0") not in ("", "0")
    if world > 1 or force:
        import torch.distributed as dist  # noqa: PLW0621

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
    # one rank per GPU; DGP_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path on a
    # single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("DGP_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    if dist is not None:
        # stdout carries ONE line, rank 0's JSON: communication libraries that print to file descriptor 1 while they connect
        # (gloo: "[Gloo] Rank 0 is connected to ...") go to stderr -- for good on the other ranks
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            dist.barrier()  # lazily created transports connect (and print) here, not under the timed region
        finally:
            sys.stdout.flush()
            if rank == 0:
                os.dup2(keep, 1)
            os.close(keep)

    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan
    from discontinuum_amd.sites import gather_site_results

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == 5:
        return run_config5(args, model, d, dev, world, rank, dist, barrier, _lib)

    dt = torch.float64 if args.dtype == "f64" else torch.float32
    n, S = args.n, max(1, args.sites_per_gpu)
    # S independent sites per rank carried by ONE batched plan: every kernel of the fit step is launched once for
    # all S sites (gridDim.z = S), so the sequential panel chain and the launch rate are amortised over them.
    # The single-site plan (latency) uses the default level 2 (early inverse on a third stream).
    level = 0 if args.no_lookahead else (1 if S > 1 else 2)
    bplan, btheta, ball, bnoise = make_plan(model, n, d, dt, dev, S, level, seed0=rank, seed_stride=world)
    if S > 1:
        plan, theta, y0, noise0 = make_plan(model, n, d, dt, dev, 1, 0 if args.no_lookahead else 2, seed0=rank)
    else:
        plan, theta, y0, noise0 = bplan, btheta, ball, bnoise

    def batch_step():
        out = bplan.fit_step(btheta, ball, bnoise)[0]
        return list(out) if S > 1 else [out]

    bplan.set_timing(True)  # HIP events around every bulk launch, on the stream it runs on (the roofline object)
    if args.roofline_only:
        # the command the profiler passes run (scripts/collect_profiles.sh): the SAME process prints what its own HIP events
        # say about every step's lauum launch, so the tracer's durations can be compared with them launch by launch, and the
        # shader clock the chip held under that kernel in THIS (profiled) process
        lauum_ms = []
        for _ in range(max(3, args.steps)):
            batch_step()
            lauum_ms.append(bplan.get_timing()[_lib.TIME_LAUUM])  # (synchronises on the step's events)
        torch.cuda.synchronize()
        ms = bplan.get_timing()
        clk = None if args.no_clock_probe else clock_probe(_lib.load(), dev, bplan.stage_lauum, max(1e-4, ms[_lib.TIME_LAUUM] * 1e-3))
        Nn = bplan.N
        print(json.dumps({"roofline_only": True, "kernel": "syrk_kernel", "launches": int(ms[_lib.TIME_SYRK_N]),
                          "ms_per_step": ms[_lib.TIME_SYRK_SUM],
                          "avg_launch_us": 1e3 * ms[_lib.TIME_SYRK_SUM] / max(1, int(ms[_lib.TIME_SYRK_N])),
                          "achieved_tflops": ms[_lib.TIME_SYRK_FLOP] / (ms[_lib.TIME_SYRK_SUM] * 1e-3) / 1e12,
                          "lauum_ms": ms[_lib.TIME_LAUUM], "lauum_ms_per_step": lauum_ms,
                          "lauum_flops": S * float(Nn) ** 3 / 3.0, "lauum_clock": clk, "sites": S, "N": Nn,
                          "stages_ms": {"gram": ms[_lib.TIME_GRAM], "potrf_wall": ms[_lib.TIME_POTRF], "trtri": ms[_lib.TIME_TRTRI],
                                        "lauum": ms[_lib.TIME_LAUUM], "solve": ms[_lib.TIME_SOLVE], "grad": ms[_lib.TIME_GRAD]},
                          "source_hash": source_hash()}))
        return
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        outs = batch_step()
    barrier()
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = batch_step()
    torch.cuda.synchronize()
    wall1 = time.time()
    # the batch gather: (NLL, info, gradient) row of every site -> every rank (RCCL all_gather, 256 B per site)
    local = torch.stack(outs)
    table = gather_site_results(local, world * S) if dist is not None else local
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    host = local.cpu().double()
    assert bool((host[:, _lib.OUT_INFO] == 0).all()) and bool(torch.isfinite(host[:, _lib.OUT_NLL]).all()), "fit step failed"
    assert table.shape[0] == world * S and bool(torch.isfinite(table[:, _lib.OUT_NLL]).all()), "a site failed"

    # ---- shader clock under load, rank 0, AFTER the timed region (the probe shares a few CUs with the load): once while
    # whole fit steps run back to back, once while the dominant kernel's stage (K^^-1 = L^-T L^-1, lauum_kernel) does
    clk_step = clk_lauum = clk_potrf = clk_trtri = None
    rep = stage_report(bplan, S, args.dtype, level, _lib) if rank == 0 else None  # last step of the TIMED REGION (before any probe)
    if rank == 0 and not args.no_clock_probe:
        lauum_s = max(1e-4, rep["stages_ms"]["lauum"] * 1e-3)
        clk_step = clock_probe(_lib.load(), dev, batch_step, elapsed / args.steps)
        clk_lauum = clock_probe(_lib.load(), dev, bplan.stage_lauum, lauum_s)
        # the other two O(n^3) stages in loops of their own (the factorisation needs a fresh K^ each time: Gram + potrf)
        def potrf_loop():
            bplan.stage_gram(btheta, bnoise)
            bplan.stage_potrf()
        clk_potrf = clock_probe(_lib.load(), dev, potrf_loop, max(1e-4, (rep["stages_ms"]["gram"] + rep["stages_ms"]["potrf_wall"]) * 1e-3))
        clk_trtri = clock_probe(_lib.load(), dev, bplan.stage_trtri, max(1e-4, rep["stages_ms"]["trtri"] * 1e-3))
        torch.cuda.synchronize()

    # ---- the headline CONFIGURATION under parity (rank 0): site 0 of the batched plan (hyperparameters of > 8 sites through
    # device scratch, default tile selectors) against the single-site plan built from the same seed -- a different schedule
    # (pairs + split chain + early inverse against groups of 4), different launches, the same matrix.  Asserted: the line is
    # only printed if the batch the value was measured on reproduces the single-site result.
    parity = None
    if rank == 0 and S > 1:
        parity = batch_parity(bplan.fit_step(btheta, ball, bnoise), plan.fit_step(theta, y0, noise0), plan.ntheta, args.dtype, _lib)
        assert parity["ok"], f"site 0 of the {S}-site batch differs from the single-site plan: {parity}"

    # ---- single-site loop on rank 0: latency of one fit alone on the GPU
    single_ms = None
    if rank == 0:
        ksingle = max(3, args.steps // 2)
        plan.fit_step(theta, y0, noise0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ksingle):
            plan.fit_step(theta, y0, noise0)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / ksingle * 1e3

    N = bplan.N
    train = None
    if args.train > 0:  # every rank takes part (fit_many_distributed); the step metric's plans are released first
        del bplan, plan
        bplan = plan = None
        torch.cuda.empty_cache()
        train = train_many(model, n, world * S, args.train, barrier=barrier, distributed=True)
        train["ranks"] = world
    if rank == 0:
        dom = rep["dominant"]
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs, gfx950 half-count correction applied: scripts/pmc_summary.py); only valid
        # for the shape AND the source tree those passes were taken on
        traffic, traffic_from = None, None
        shash = source_hash()
        for cand in ("r05_pmc_hbm_n8192_f64.json", "r04_pmc_hbm_n8192_f64.json", "r03_pmc_hbm_n8192_f64.json", "r02_pmc_hbm_n8192_f64.json"):
            pmc_path = os.path.join(ROOT, "profiles", cand)
            if not os.path.exists(pmc_path) or (model, n, d, args.dtype) != ("loadest", 8192, 3, "f64"):
                continue
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("sites_per_launch") == S and pmc.get("source_hash") == shash:
                    key = next(k for k in pmc["kernels"] if k.startswith(dom))
                    traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
                    traffic_from = f"profiles/{cand} (commit {pmc.get('commit')}, source_hash {shash})"
            except Exception:  # noqa: BLE001
                traffic = None
        dom_clk = clk_lauum if dom == "lauum_kernel" else clk_step
        roofline = {
            "bound": "mfma", "kernel": dom, "achieved": rep["achieved"], "peak": rep["peak"], "unit": "TFLOP/s",
            "frac": rep["frac"], "traffic": traffic, "traffic_from": traffic_from,
            # the clock the chip held under this kernel (in-kernel stamps of a separate probe launch) and the fraction of
            # the peak AT THAT CLOCK: `frac` mixes pipe utilisation with DVFS, `frac_at_clock` is the utilisation alone
            "clock_mhz": dom_clk["mhz"] if dom_clk else None,
            "frac_at_clock": (rep["achieved"] / (rep["peak"] * dom_clk["mhz"] / NOMINAL_MHZ)) if (dom_clk and rep["achieved"]) else None,
            "clock_probe": {"nominal_mhz": NOMINAL_MHZ, "dominant_kernel_loop": clk_lauum if dom == "lauum_kernel" else None,
                            "lauum_loop": clk_lauum, "potrf_loop": clk_potrf, "trtri_loop": clk_trtri, "whole_steps": clk_step,
                            "stages_frac_at_clock": {k: (rep["stages_tflops"][kk] / (rep["peak"] * c["mhz"] / NOMINAL_MHZ)
                                                         if (c and rep["stages_tflops"].get(kk)) else None)
                                                     for k, kk, c in (("syrk_kernel (bulk updates, in situ)", "syrk_kernel", clk_potrf),
                                                                      ("trtri_level_kernel", "trtri_level_kernel", clk_trtri),
                                                                      ("lauum_kernel", "lauum_kernel", clk_lauum))}},
            "launches_per_step": rep["launches"], "ms_per_step": rep["ms"],
            "stages_tflops": rep["stages_tflops"], "stages_ms": rep["stages_ms"],
            "potrf_stage_tflops": rep["potrf_stage_tflops"],
            "fit_flops": float(N) ** 3, "job_tflops": float(N) ** 3 * world * S * args.steps / elapsed / 1e12,
            "measured_in": "the timed region itself (last step): every launch carries %d site(s), lookahead level %d" % (S, level),
            "gram_hbm": rep["gram_hbm"], "gram_grad_hbm": rep["gram_grad_hbm"], "source_hash": shash,
        }
        result = {
            "metric": METRIC,
            "value": world * S * args.steps / elapsed,
            "unit": "fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"synthetic {model}-gp kernel, n={n} d={d} {args.dtype} exact GP, "
                                   f"{S} independent site(s) per GPU in one batched plan (one step = one fit of each)",
                       "n": n, "d": d, "sites_per_gpu": S, "fits_per_step": world * S,
                       "lookahead": not args.no_lookahead, "nll_site0": float(host[0, _lib.OUT_NLL]), "parity": parity},
            "gpu_state": sampler.window(wall0, wall1) if sampler is not None else None,
            "single_site": {"fits_per_s": 1e3 / single_ms, "ms_per_fit": single_ms,
                            "tflops": float(N) ** 3 / (single_ms * 1e-3) / 1e12,
                            "note": "one site alone on one GPU, steps strictly sequential (a single fit loop)"},
            "roofline": roofline,
        }
        if train is not None:
            result["train"] = train
        default_run = (world == 1 and (model, n, d, args.dtype) == ("loadest", 8192, 3, "f64"))
        del bplan, plan
        torch.cuda.empty_cache()
        if default_run and not args.no_configs:
            result["configs"] = run_configs(dev, _lib, args.quick_configs)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(n, d, max(1, args.cpu_steps))
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_config5(args, model, d, dev, world, rank, dist, barrier, lib):
    """BASELINE config 5: ONE matrix over all ranks -- distributed factorisation, inverse and gradient
    (discontinuum_amd/dist_chol.py::DistributedFit.fit_step).  One step = one NLL + gradient evaluation.  The
    `roofline` object is rank 0's dominant MFMA stage (trailing update / inverse / K^^-1 products): flops counted tile by
    tile (DistributedFit.stage_flops), time = HIP events around that stage's launches on the stream they run on, in one
    extra step after the timed region (the events serialise nothing, but they are kept out of `value`)."""
    from discontinuum_amd import dist_chol

    dt = torch.float64 if args.dtype == "f64" else torch.float32
    n = args.n
    X, r, noise, theta = site(model, n, d, 0)
    kw = {"group_panels": args.group_panels} if args.group_panels else {}
    ctx = dist_chol.DistributedFit(model, n, d, dtype=dt, device=dev, rank=rank, world=world, **kw)
    ctx.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
    rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
    for _ in range(args.warmup):
        out = ctx.fit_step(theta, rd, nd)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ctx.fit_step(theta, rd, nd)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ctx.set_timing(True)
    for k in ctx.comm.calls:
        ctx.comm.calls[k] = 0  # `collectives_issued` = what ONE fit step issues
    ctx.fit_step(theta, rd, nd)  # every rank takes part; rank 0 reports its own stages
    stage_ms = ctx.get_timing()
    if rank == 0:
        N = ctx.N
        host = out.cpu().double()
        flops = ctx.stage_flops()
        dom = max(flops, key=lambda k: stage_ms[k])
        peak = PEAK_TFLOPS[args.dtype]
        ach = flops[dom] / (stage_ms[dom] * 1e-3) / 1e12 if stage_ms[dom] > 0 else None
        names = {"update": "slab_syrk_kernel", "invert": "slab_tconv_kernel + slab_xacc_kernel", "product": "slab_ttt_kernel"}
        roofline = {"bound": "mfma", "kernel": names[dom], "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": ach / peak if ach else None, "traffic": None,
                    "ms_per_step": stage_ms[dom], "flops_per_step_rank0": flops[dom],
                    "stages_ms": stage_ms,
                    "stages_tflops": {k: (flops[k] / (stage_ms[k] * 1e-3) / 1e12 if stage_ms[k] > 0 else None) for k in flops},
                    "measured_in": "one extra fit step after the timed region, rank 0's launches",
                    "collectives_issued": dict(ctx.comm.calls)}
        print(json.dumps({
            "metric": METRIC, "value": args.steps / elapsed, "unit": "fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ONE {model}-gp matrix n={n} d={d} {args.dtype}: block-cyclic distributed Cholesky, "
                                   f"inverse and gradient over {world} rank(s)", "n": n, "d": d,
                       "nll": float(host[lib.OUT_NLL]), "info": int(host[lib.OUT_INFO]), "group_panels": ctx.W,
                       "backend": (dist.get_backend() if dist is not None else None)},
            "job_tflops": float(N) ** 3 * args.steps / elapsed / 1e12,
            "per_rank_hbm_gib": ctx.hbm_bytes() / 2 ** 30, "cpu_baseline": None, "roofline": roofline}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
