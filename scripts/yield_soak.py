#!/usr/bin/env python3
"""Soak of the cooperative yield (DGP_OPT_CHAIN_YIELD): (1) many sizes, hint on against off, whole result rows bitwise equal;
(2) 1500 back-to-back single-site steps at n = 8192 and 400 at n = 16384 fp32 with the hint on: every row bitwise the first,
info = 0, and the step time stays flat (a stuck wait would show as the 0.2 ms bound times the number of checks).
usage: python scripts/yield_soak.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from discontinuum_amd import _lib
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0")
bad = 0
for dtn in ("f64", "f32"):
    dt = torch.float64 if dtn == "f64" else torch.float32
    for n in (130, 257, 640, 1100, 1536, 2600, 3333, 5000, 6144, 7000, 9000):
        X, r, noise, theta = bench.site("loadest", n, 3, n)
        rows = []
        for hint in (1, 0):
            p = GPPlan("loadest", n, 3, dtype=dt, device=dev)
            p.set_option(_lib.OPT_CHAIN_YIELD, hint)
            p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
            rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
            out = [p.fit_step(theta, rd, nd)[0].clone() for _ in range(3)]
            rows.append(out)
            del p
        same = all(torch.equal(a, rows[0][0]) for a in rows[0] + rows[1])
        ok = bool(rows[0][0][_lib.OUT_INFO] == 0)
        bad += (not same) + (not ok)
        print(f"{dtn} n={n:5d}: bitwise {same} info ok {ok} nll {float(rows[0][0][0])!r}", flush=True)
for model, n, d, dtn, reps in (("loadest", 8192, 3, "f64", 1500), ("rating", 16384, 2, "f32", 400)):
    dt = torch.float64 if dtn == "f64" else torch.float32
    X, r, noise, theta = bench.site(model, n, d, 0)
    p = GPPlan(model, n, d, dtype=dt, device=dev)
    p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
    rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
    first = p.fit_step(theta, rd, nd)[0].clone()
    torch.cuda.synchronize()
    diff, times = 0, []
    for blk in range(reps // 100):
        t0 = time.perf_counter()
        outs = [p.fit_step(theta, rd, nd)[0].clone() for _ in range(100)]
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 10)
        diff += sum(not torch.equal(o, first) for o in outs)
    bad += diff
    print(f"{model} n={n} {dtn}: {reps} steps, rows that differ from the first: {diff}; ms per step per block of 100: min {min(times):.3f} max {max(times):.3f}", flush=True)
    del p
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
