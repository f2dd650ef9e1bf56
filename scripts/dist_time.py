"""Timing of the distributed-factorisation driver with ONE rank (no communication): what the building blocks cost
against the single-GPU factorisation of the same matrix."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
from discontinuum_amd.dist_chol import distributed_nll
from oracle.gp_oracle import synth_loadest
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
dev = torch.device("cuda:0")
X, y = synth_loadest(n, 3, 0)
p = GPPlan("loadest", n, 3, dtype=dt, device=dev); p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
yd = torch.tensor(y, dtype=dt, device=dev); noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931471805599453] * 11
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def single(): p.stage_gram(theta, noise); p.stage_potrf()
print(f"n={n} {dt}: gram+potrf (single-GPU schedule) {t(single):.1f} ms")
for W in (2, 4, 8):
    ms = t(lambda: distributed_nll(p, theta, yd, noise, group_panels=W))
    print(f"  distributed driver, 1 rank, groups of {W}: gram + factor + forward solve {ms:.1f} ms")
ref = p.fit_step(theta, yd, noise)[0].cpu(); out = distributed_nll(p, theta, yd, noise, group_panels=4).cpu()
print("  NLL", float(out[0]), "vs fit_step", float(ref[0]))
# pieces
from discontinuum_amd import _lib
def factor_only(W=4):
    p.stage_gram(theta, noise); p.dist_begin(); nbk = p.N // 128
    for g in range((nbk + W - 1) // W):
        p.dist_factor_group(g * W, min(W, nbk - g * W)); p.dist_update(g * W, W, 0, 1)
print(f"  factor only (W=4): {t(factor_only):.1f} ms;  forward solve only: {t(lambda: p.dist_finish(yd, 0.0, 0)):.1f} ms")
def chain_only(W=4):
    p.stage_gram(theta, noise); p.dist_begin(); nbk = p.N // 128
    for g in range((nbk + W - 1) // W): p.dist_factor_group(g * W, min(W, nbk - g * W))
def update_only(W=4):
    nbk = p.N // 128
    for g in range((nbk + W - 1) // W): p.dist_update(g * W, W, 0, 1)
print(f"  chain launches only: {t(chain_only):.1f} ms; update launches only: {t(update_only):.1f} ms")
