"""fit_many's training loop per iteration (multisite_fit.LAST_TIMING), closed-form host algebra against the autograd path:
CF=0 python scripts/fm_time.py   (OMP_NUM_THREADS=... to see what the loop's one-thread guard protects against)"""
import os
import sys

sys.path.insert(0, ".")
import torch

import bench
from discontinuum_amd import multisite_fit

cf = os.environ.get("CF", "1") == "1"
orig = multisite_fit.fit_many


def patched(*a, **k):
    k.setdefault("closed_form", cf)
    return orig(*a, **k)


multisite_fit.fit_many = patched
print("torch threads", torch.get_num_threads(), "closed_form", cf)
for cfg in (("loadest", 300, 256, 100), ("rating", 300, 256, 100), ("loadest", 1000, 128, 50), ("loadest", 4096, 64, 20)):
    r = bench.train_many(*cfg)
    print(cfg, round(r["ms_per_iteration"], 3), "ms/iter; whole call", round(r["seconds"], 3), "s; set-up + hand-back", round(r["setup_s"], 3), "s; closed form:", r["closed_form_host_algebra"])
