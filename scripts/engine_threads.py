"""Engine-level iteration times (bench.engine_iteration) under the process's torch thread setting:
OMP_NUM_THREADS=1 python scripts/engine_threads.py"""
import os
import sys

sys.path.insert(0, ".")
import torch

import bench

print("torch threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)))
for fam, n, it in (("loadest", 300, 200), ("rating", 300, 200), ("loadest", 8192, 10)):
    r = bench.engine_iteration(fam, n, it)
    print(fam, n, round(r["ms_per_iteration"], 3), "ms/iter; predict warm", round(r["predict_ms"], 3), "ms")
