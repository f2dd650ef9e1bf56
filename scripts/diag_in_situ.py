#!/usr/bin/env python3
"""Where does the diagonal-block kernel's time go UNDER a bulk update?  A second build of the library (-DDGP_DIAG_LOG,
discontinuum_amd/libdgp_hip_log.so: see the end of this docstring) logs, per 128-column block, the 100 MHz wall clock at
the workgroup's first and last instruction and the CU it ran on.  Run under `rocprofv3 --kernel-trace`: the profiler's
duration of the same launch minus the in-kernel span is the time the launch waited to be placed.
  run:      rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/diag_in_situ.py run MODEL N DTYPE LOG.json [SITES]
  combine:  python3 scripts/diag_in_situ.py combine OUT/.../*_kernel_trace.csv LOG.json
  build:    for f in dgp_gram dgp_chol dgp_api dgp_dist dgp_selftest: hipcc <Makefile flags> -DDGP_DIAG_LOG -c f.hip; link -> libdgp_hip_log.so"""
import csv
import ctypes as C
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import numpy as np
    import torch
    from discontinuum_amd import _lib

    _lib.LIB_PATH = os.path.join(ROOT, "discontinuum_amd", "libdgp_hip_log.so")
    import bench
    from discontinuum_amd.backend import GPPlan

    model, n, dtn, out = sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    S = int(sys.argv[6]) if len(sys.argv) > 6 else 1  # sites in the plan (the log is site 0's workgroup)
    dt = torch.float64 if dtn == "f64" else torch.float32
    d = 3 if model == "loadest" else 2
    dev = torch.device("cuda:0")
    if S > 1:
        p, theta, rd, nd = bench.make_plan(model, n, d, dt, dev, S, 1)
    else:
        X, r, noise, theta = bench.site(model, n, d, 0)
        p = GPPlan(model, n, d, dtype=dt, device=dev)
        p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
        rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
    for _ in range(4):
        o = p.fit_step(theta, rd, nd)[0].reshape(-1)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.dgp_debug_diag_log.restype = C.c_int
    lib.dgp_debug_diag_log.argtypes = [C.c_void_p]
    buf = np.zeros(4096, dtype=np.uint64)
    assert lib.dgp_debug_diag_log(buf.ctypes.data) == 0
    nbk = p.N // 128
    json.dump({"nbk": nbk, "begin": buf[0::4][:nbk].tolist(), "where": buf[1::4][:nbk].tolist(), "end": buf[2::4][:nbk].tolist(),
               "nll": float(o[0])}, open(out, "w"))
    print("logged", nbk, "blocks; nll", float(o[0]))
else:
    rows = sorted(({"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])}
                   for r in csv.DictReader(open(sys.argv[2]))), key=lambda r: r["start"])
    log = json.load(open(sys.argv[3]))
    last = rows[max(i for i, r in enumerate(rows) if "gram_sym" in r["name"]):]
    diags = [r for r in last if "potrf_diag" in r["name"]]
    bulk = [r for r in last if "syrk_kernel" in r["name"]]
    assert len(diags) == log["nbk"], (len(diags), log["nbk"])
    print(" k   profiler us  in-kernel us   waiting us   under bulk   xcc  cu")
    tot = [0.0, 0.0, 0, 0.0, 0.0, 0]
    for k, dgn in enumerate(diags):
        prof = (dgn["end"] - dgn["start"]) / 1e3
        inner = (log["end"][k] - log["begin"][k]) / 100.0
        under = any(b["start"] < dgn["end"] and b["end"] > dgn["start"] for b in bulk)
        w = log["where"][k]
        print(f"{k:3d} {prof:12.1f} {inner:13.1f} {prof - inner:12.1f}   {'yes' if under else 'no ':10s} {w >> 16:4d} {w & 255:3x}")
        i = 0 if under else 3
        tot[i] += prof
        tot[i + 1] += inner
        tot[i + 2] += 1
    for lab, i in (("under a bulk launch", 0), ("alone", 3)):
        if tot[i + 2]:
            print(f"{lab}: {tot[i + 2]} launches, mean profiler duration {tot[i] / tot[i + 2]:.1f} us, mean in-kernel span {tot[i + 1] / tot[i + 2]:.1f} us")
