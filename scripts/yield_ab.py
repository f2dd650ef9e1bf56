#!/usr/bin/env python3
"""A/B of the cooperative yield (DGP_CHAIN_YIELD = 0 / 1; dgp_common.h: yield_if_asked) on single-site plans, alternating
processes on one box: fit-step time, potrf wall, bulk sum, and the NLL (the yield is a hint: results must not change).
usage: python scripts/yield_ab.py            (driver)      python scripts/yield_ab.py worker   (one setting)"""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
SHAPES = (("rating", 16384, 2, "f32"), ("loadest", 16384, 3, "f32"), ("loadest", 8192, 3, "f64"), ("loadest", 4096, 3, "f64"),
          ("loadest", 2048, 3, "f64"), ("loadest", 12288, 3, "f64"))


def worker():
    import torch
    import bench
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan

    dev = torch.device("cuda:0")
    for model, n, d, dtn in SHAPES:
        dt = torch.float64 if dtn == "f64" else torch.float32
        X, r, noise, theta = bench.site(model, n, d, 0)
        p = GPPlan(model, n, d, dtype=dt, device=dev)
        p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
        p.set_timing(True)
        rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
        for _ in range(3):
            out = p.fit_step(theta, rd, nd)[0]
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(8):
                out = p.fit_step(theta, rd, nd)[0]
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 8)
        ms = p.get_timing()
        print(f"  {model:8s} n={n:6d} {dtn}  step {best * 1e3:7.3f} ms  potrf {ms[_lib.TIME_POTRF]:7.3f} (bulk {ms[_lib.TIME_SYRK_SUM]:7.3f}) "
              f"trtri {ms[_lib.TIME_TRTRI]:7.3f} lauum {ms[_lib.TIME_LAUUM]:7.3f}  nll {float(out[0])!r} grad0 {float(out[4])!r}", flush=True)
        del p
        torch.cuda.empty_cache()


if len(sys.argv) > 1 and sys.argv[1] == "worker":
    worker()
else:
    for y in ("0", "1", "0", "1"):
        print(f"DGP_CHAIN_YIELD={y}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, DGP_CHAIN_YIELD=y), check=False, timeout=250)
