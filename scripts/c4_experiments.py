#!/usr/bin/env python3
"""BASELINE config 4's per-GPU share (64 loadest sites of n = 4096, fp64) under different schedules:
  * ONE batched plan of 64 against 2 x 32 / 4 x 16 plans on separate streams (one half's panel chain under the other's
    bulk updates);
  * panels per group of the batched factorisation (DGP_GROUP = 2 / 4 / 8; read per call).
Per configuration: ms per sweep of all 64 sites, fits/s, and the stage times of the first plan (HIP events).
usage: python scripts/c4_experiments.py [n=4096] [sites=64]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd import _lib  # noqa: E402
from discontinuum_amd.backend import GPPlan  # noqa: E402
from oracle.gp_oracle import synth_loadest  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev, dt, d = torch.device("cuda:0"), torch.float64, 3


def mk(B, seed0):
    Xs, ys = zip(*[synth_loadest(n, d, seed0 + b) for b in range(B)])
    p = GPPlan("loadest", n, d, dtype=dt, device=dev, lookahead=1, batch=B)
    p.set_inputs(torch.tensor(np.stack(Xs), device=dev).contiguous())
    p.set_timing(True)
    return p, torch.tensor(np.stack(ys), device=dev).contiguous(), torch.full((B, n), 0.01, dtype=dt, device=dev), [0.6931471805599453] * (11 * B)


def run(cfg, label, reps=10):
    plans = [mk(B, 10 * i) for i, B in enumerate(cfg)]
    streams = [torch.cuda.Stream(device=dev) for _ in cfg] if len(cfg) > 1 else [torch.cuda.current_stream()]

    def step():
        for (p, y, nz, th), st in zip(plans, streams):
            with torch.cuda.stream(st):
                p.fit_step(th, y, nz)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):  # three rounds, the best mean (rule 24: distributions, one process)
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    ms = plans[0][0].get_timing()
    print(f"{label:34s} plans {str(cfg):18s} {best * 1e3:8.2f} ms per sweep  {sum(cfg) / best:7.1f} fits/s   first plan: potrf {ms[_lib.TIME_POTRF]:6.2f} "
          f"(bulk {ms[_lib.TIME_SYRK_SUM]:6.2f}) trtri {ms[_lib.TIME_TRTRI]:6.2f} lauum {ms[_lib.TIME_LAUUM]:6.2f}", flush=True)
    del plans


for g in ("", "2", "8", "6"):
    if g:
        os.environ["DGP_GROUP"] = g
    else:
        os.environ.pop("DGP_GROUP", None)
    run((S,), f"DGP_GROUP={g or 'default(4)'}")
os.environ.pop("DGP_GROUP", None)
run((S // 2, S // 2), "two half-batches, two streams")
run((S // 4,) * 4, "four quarter-batches")
run((S,), "one plan again (drift check)")
