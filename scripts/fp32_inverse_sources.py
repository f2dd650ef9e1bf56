#!/usr/bin/env python3
"""Which stage of an fp32 plan puts the error into K^^-1 (S = T^T T): the factor L (L L^T = K^ + E1), the inverse
T (T L = I + F) or the product itself (S - T^T T).  Everything recomputed in fp64 by torch on the GPU from the fp32
plan's buffers.  The figure of merit is the outputscale-like trace tr(S K^) - n (exactly 0 for the true inverse)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discontinuum_amd import _lib  # noqa: E402
from discontinuum_amd.backend import GPPlan  # noqa: E402
from tests.test_gpu_stages import make_case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
model = sys.argv[2] if len(sys.argv) > 2 else "rating"
d = 2 if model == "rating" else 3
dev = torch.device("cuda:0")
X, r, noise, theta = make_case(model, d, n, seed=7, perturb=0.1)
p64 = GPPlan(model, n, d, dtype=torch.float64, device=dev)
p64.set_inputs(X.to(dev).contiguous())
p64.stage_gram(theta, noise.to(dev))
K = p64.buffer(_lib.BUF_A).clone()
K = torch.tril(K) + torch.tril(K, -1).T
del p64
p = GPPlan(model, n, d, dtype=torch.float32, device=dev)
p.set_inputs(X.float().to(dev).contiguous())
p.stage_gram(theta, noise.float().to(dev))
K32 = p.buffer(_lib.BUF_A).double()
K32 = torch.tril(K32) + torch.tril(K32, -1).T
p.fit_step(theta, r.float().to(dev), noise.float().to(dev))
L = torch.tril(p.buffer(_lib.BUF_A)).double()
T = torch.tril(p.buffer(_lib.BUF_T)).double()
S = p.buffer(_lib.BUF_S).double()
S = torch.tril(S) + torch.tril(S, -1).T
eye = torch.eye(n, dtype=torch.float64, device=dev)
print("tr(S32 K) - n                      ", (torch.sum(S * K) - n).item())
E0 = K32 - K
print("Gram in fp32: ||K32 - K||_F/||K||_F", (E0.norm() / K.norm()).item())
E1 = L @ L.T - K32
print("factor: ||L L^T - K32||_F / ||K||_F", (E1.norm() / K.norm()).item())
F = T @ L - eye
print("inverse: ||T L - I||_F", F.norm().item(), " tr(F)", torch.trace(F).item(), " max|F|", F.abs().max().item())
TtT = T.T @ T
G = S - TtT
print("product: ||S - T^T T||_F / ||S||_F", (G.norm() / S.norm()).item(), " tr((S - T^T T) K)", torch.sum(G * K).item())
print("tr(T^T T K) - n (exact product of the fp32 T)", (torch.sum(TtT * K) - n).item())
Linv = torch.linalg.solve_triangular(L, eye, upper=False)
print("tr(L^-T L^-1 K) - n (exact inverse of the fp32 L)", (torch.sum((Linv.T @ Linv) * K) - n).item())
print("||T - L^-1||_F / ||L^-1||_F", ((T - Linv).norm() / Linv.norm()).item())
# per block column of T: relative error (does it grow with the level recursion / distance from the diagonal?)
dT = (T - Linv)
for lo in range(0, n, n // 8):
    blk = slice(lo, lo + n // 8)
    print("  rows", lo, " ||dT||/||Linv|| of the row block", (dT[blk].norm() / Linv[blk].norm()).item(),
          " diag block only", (dT[blk, blk].norm() / Linv[blk, blk].norm()).item())
