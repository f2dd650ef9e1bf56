#!/usr/bin/env python3
"""A/B of the bulk update's occupancy on single-site plans (DGP_BULK_LDS_PAD = 0: three workgroups per CU; 14336: two, the
default inside the size window of dgp_chol.hip::launch_bulk), alternating processes on one box.
usage: python scripts/bulk_pad_ab.py"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch, bench
    from discontinuum_amd import _lib
    from discontinuum_amd.backend import GPPlan
    dev = torch.device("cuda:0")
    for model, n, d, dtn in (("loadest", 32768, 3, "f32"), ("loadest", 24576, 3, "f32"), ("loadest", 16384, 3, "f64"), ("loadest", 6144, 3, "f64"), ("loadest", 2048, 3, "f64"), ("loadest", 1024, 3, "f64")):
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
            for _ in range(6):
                out = p.fit_step(theta, rd, nd)[0]
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 6)
        ms = p.get_timing()
        print(f"  {model} n={n} {dtn}: step {best*1e3:7.3f} potrf {ms[_lib.TIME_POTRF]:7.3f} (bulk {ms[_lib.TIME_SYRK_SUM]:7.3f}) nll {float(out[0])!r}", flush=True)
        del p; torch.cuda.empty_cache()
else:
    for pad in ("0", "14336", "0", "14336"):
        print("DGP_BULK_LDS_PAD", pad, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "w"], env=dict(os.environ, DGP_BULK_LDS_PAD=pad), timeout=200)
