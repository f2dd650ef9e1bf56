"""Split panel chain (DGP_SPLIT_CHAIN, default on) against the single-stream chain: bitwise comparison of L, L^-1 and
the result row, then timings.  usage: python scripts/split_chain_check.py [f64|f32] [n ...]"""
import os, sys, time
import torch
sys.path.insert(0, ".")
import bench
from discontinuum_amd import _lib
from discontinuum_amd.backend import GPPlan

dtn = sys.argv[1] if len(sys.argv) > 1 else "f64"
ns = [int(a) for a in sys.argv[2:]] or [512, 640, 1000, 2048, 3300, 4096, 8192]
dt = torch.float64 if dtn == "f64" else torch.float32
dev = torch.device("cuda:0")
level = int(os.environ.get("LEVEL", "2"))
for n in ns:
    X, r, noise, theta = bench.site("loadest", n, 3, 0)
    Xd = torch.tensor(X, dtype=dt, device=dev).contiguous(); rd = torch.tensor(r, dtype=dt, device=dev); nd = torch.tensor(noise, dtype=dt, device=dev)
    res = {}
    for mode in ("0", "1"):
        os.environ["DGP_SPLIT_CHAIN"] = mode
        p = GPPlan("loadest", n, 3, dtype=dt, device=dev, lookahead=level)
        p.set_inputs(Xd)
        out, a, dn = p.fit_step(theta, rd, nd)
        torch.cuda.synchronize()
        res[mode] = (out.cpu().clone(), torch.tril(p.buffer(_lib.BUF_A)).cpu().clone(), torch.tril(p.buffer(_lib.BUF_T)).cpu().clone())
        reps = 20 if n <= 8192 else 3
        for _ in range(5):
            p.fit_step(theta, rd, nd)
        best = float("inf")
        for _batch in range(4):  # best of four batches: the first steps after a plan is created / freed are erratic
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out, a, dn = p.fit_step(theta, rd, nd)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        ms = best
        p.set_timing(True); p.fit_step(theta, rd, nd); torch.cuda.synchronize(); tm = p.get_timing()
        res[mode] += (ms, tm[_lib.TIME_POTRF])
        del p
    o0, L0, T0, ms0, pf0 = res["0"]; o1, L1, T1, ms1, pf1 = res["1"]
    print(f"n={n} {dtn} level {level}: info {int(o0[3])}/{int(o1[3])}  NLL {o0[0].item():.12g} / {o1[0].item():.12g}  "
          f"L equal {torch.equal(L0, L1)} (max diff {(L0 - L1).abs().max().item():.2e})  T equal {torch.equal(T0, T1)}  row equal {torch.equal(o0, o1)}  "
          f"fit step {ms0:.3f} -> {ms1:.3f} ms  potrf {pf0:.3f} -> {pf1:.3f} ms", flush=True)
