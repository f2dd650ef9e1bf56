#!/usr/bin/env python3
"""The headline batch (32 loadest sites of n = 8192, fp64) as ONE batched plan against several smaller plans on separate
streams: does one part's chain-bound / HBM-bound phase hide under another's MFMA-bound phase?  Uneven splits keep the
parts' stage boundaries apart without an explicit offset.
usage: python scripts/phase_experiments.py [n=8192] [sites=32]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan  # noqa: E402
from oracle.gp_oracle import synth_loadest  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev, dt, d = torch.device("cuda:0"), torch.float64, 3
sites = [synth_loadest(n, d, b) for b in range(S)]


def mk(lo, B):
    Xs, ys = zip(*sites[lo:lo + B])
    p = GPPlan("loadest", n, d, dtype=dt, device=dev, lookahead=1, batch=B)
    p.set_inputs(torch.tensor(np.stack(Xs), device=dev).contiguous())
    return p, torch.tensor(np.stack(ys), device=dev).contiguous(), torch.full((B, n), 0.01, dtype=dt, device=dev), [0.6931471805599453] * (11 * B)


def run(cfg, reps=8):
    plans, lo = [], 0
    for B in cfg:
        plans.append(mk(lo, B))
        lo += B
    streams = [torch.cuda.Stream(device=dev) for _ in cfg] if len(cfg) > 1 else [torch.cuda.current_stream()]

    def step():
        outs = []
        for (p, y, nz, th), st in zip(plans, streams):
            with torch.cuda.stream(st):
                outs.append(p.fit_step(th, y, nz)[0])
        return outs

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            outs = step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    nll = torch.cat([o.reshape(-1, o.shape[-1])[:, 0] for o in outs]).cpu().numpy()
    print(f"plans {str(cfg):22s} {best * 1e3:8.2f} ms per sweep  {sum(cfg) / best:7.2f} fits/s   nll[0] {nll[0]:.9f} nll[-1] {nll[-1]:.9f}", flush=True)
    del plans
    torch.cuda.empty_cache()


for cfg in ((S,), (S // 2, S // 2), (S * 5 // 8, S * 3 // 8), (S * 3 // 4, S // 4), (S,)):
    run(cfg)
