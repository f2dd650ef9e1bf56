#!/usr/bin/env python3
"""Where the fp32 plan's gradient error comes from (config 3's size): its K^^-1 (S = T^T T from the fp32 factor), its alpha,
or the fp32 contraction 1/2 sum (S - alpha alpha^T) dK itself.  The fp64 plan's contraction kernel is used as the exact
contraction: its S / alpha buffers are overwritten with the fp32 plan's.  python scripts/fp32_grad_sources.py [n] [model]"""
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
P = theta.numel()
p64 = GPPlan(model, n, d, dtype=torch.float64, device=dev)
p64.set_inputs(X.to(dev).contiguous())
o64, a64, _ = p64.fit_step(theta, r.to(dev), noise.to(dev))
g64 = o64[4:4 + P].cpu()
S64 = p64.buffer(_lib.BUF_S).clone()
A64 = p64.buffer(_lib.BUF_ALPHA).clone()
p = GPPlan(model, n, d, dtype=torch.float32, device=dev)
p.set_inputs(X.float().to(dev).contiguous())
o32, a32, _ = p.fit_step(theta, r.float().to(dev), noise.float().to(dev))
g32 = o32[4:4 + P].cpu().double()
S32 = p.buffer(_lib.BUF_S).double()
A32 = p.buffer(_lib.BUF_ALPHA).double()
rel = lambda g: ((g - g64).abs().max() / g64.abs().max()).item()  # noqa: E731
print("fp32 plan as shipped            ", rel(g32))


def contract(S, A):
    p64.buffer(_lib.BUF_S).copy_(S)
    p64.buffer(_lib.BUF_ALPHA).copy_(A)
    return p64.stage_grad(theta).cpu()


print("exact contraction, S32, alpha32 ", rel(contract(S32, A32)))
print("exact contraction, S64, alpha32 ", rel(contract(S64, A32)))
print("exact contraction, S32, alpha64 ", rel(contract(S32, A64)))
print("exact contraction, S64, alpha64 ", rel(contract(S64, A64)))
# the two halves of the gradient and how much they cancel
zero = torch.zeros_like(A64)
gS = contract(S64, zero)
print("1/2 tr(S dK)      ", gS.tolist())
print("gradient          ", g64.tolist())
print("cancellation max |1/2 tr(S dK)| / max |g|:", (gS.abs().max() / g64.abs().max()).item())
print("per-parameter error of the shipped fp32 gradient / |1/2 tr(S dK)_p|:", ((g32 - g64).abs() / gS.abs()).tolist())
dS = torch.tril(S32 - S64)
print("||S32 - S64||_F / ||S64||_F (lower):", (dS.norm() / torch.tril(S64).norm()).item(), " max abs", dS.abs().max().item(), " max |S64|", S64.abs().max().item())
