#!/usr/bin/env python3
"""A/B of two builds of libdgp_hip.so on ONE box: stage times of the headline batched plan (and of config 4's share).
usage: python scripts/ab_lib.py <lib.so> [n=8192] [sites=32] [reps=6]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
S = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
plan, th, r, noise = bench.make_plan("loadest", n, 3, torch.float64, dev, S, 1)
plan.set_timing(True)
for _ in range(4):
    plan.fit_step(th, r, noise)
torch.cuda.synchronize()
best = None
for _ in range(reps):
    plan.fit_step(th, r, noise)
    torch.cuda.synchronize()
    ms = plan.get_timing()
    tot = sum(ms[k] for k in (_lib.TIME_GRAM, _lib.TIME_POTRF, _lib.TIME_TRTRI, _lib.TIME_LAUUM, _lib.TIME_SOLVE, _lib.TIME_GRAD))
    if best is None or tot < best[0]:
        best = (tot, list(ms))
tot, ms = best
print(f"{os.path.basename(sys.argv[1]):24s} n={n} S={S}: step {tot:8.2f} ms  potrf {ms[_lib.TIME_POTRF]:7.2f}  trtri {ms[_lib.TIME_TRTRI]:7.2f}  lauum {ms[_lib.TIME_LAUUM]:7.2f}", flush=True)
