#!/usr/bin/env python3
"""fp32 plans with and without the fp64 refinement step (DGP_OPT_REFINE): fit-step time and errors against the fp64 plan
on the same inputs.  python scripts/refine_ab.py [n=16384] [seeds=3]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discontinuum_amd import _lib  # noqa: E402
from discontinuum_amd.backend import GPPlan  # noqa: E402
from tests.test_gpu_stages import make_case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
rows = []
for model, d in (("rating", 2), ("loadest", 3)):
    for seed in range(7, 7 + seeds):
        X, r, noise, theta = make_case(model, d, n, seed=seed, perturb=0.1)
        P = theta.numel()
        p64 = GPPlan(model, n, d, dtype=torch.float64, device=dev)
        p64.set_inputs(X.to(dev).contiguous())
        o64, a64, dn64 = p64.fit_step(theta, r.to(dev), noise.to(dev))
        o64 = o64.cpu()
        del p64
        p = GPPlan(model, n, d, dtype=torch.float32, device=dev)
        p.set_inputs(X.float().to(dev).contiguous())
        rr, nn = r.float().to(dev), noise.float().to(dev)
        for refine in (1, 0):
            p.set_option(_lib.OPT_REFINE, refine)
            for _ in range(2):
                o, a, dn = p.fit_step(theta, rr, nn)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                o, a, dn = p.fit_step(theta, rr, nn)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            o = o.cpu().double()
            rows.append({"model": model, "n": n, "seed": seed, "refine": refine, "ms": round(ms, 3),
                         "nll64": o64[0].item(), "nll_rel": (abs(o[0] - o64[0]) / abs(o64[0])).item(),
                         "nll_bound": 1e-4 * max(1.0, n / 1024), "quad_abs": (o[1] - o64[1]).item(), "logdet_abs": (o[2] - o64[2]).item(),
                         "grad_rel": ((o[4:4 + P] - o64[4:4 + P]).abs().max() / o64[4:4 + P].abs().max()).item(),
                         "alpha_rel": (torch.linalg.norm(a.double() - a64) / torch.linalg.norm(a64)).item(),
                         "dnoise_rel": ((dn.double() - dn64).abs().max() / dn64.abs().max()).item()})
            print(json.dumps(rows[-1]), flush=True)
        del p
        torch.cuda.empty_cache()
