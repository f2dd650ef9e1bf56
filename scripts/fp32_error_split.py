"""Where does the fp32 path's NLL error come from?  For one matrix:
   NLL32      the fp32 plan's result row
   NLL(G32)   fp64 Cholesky (torch, on the GPU -- measurement script only) of the fp32 plan's OWN Gram matrix, cast up
   NLL64      the fp64 plan on the fp64 inputs (= oracle to 1e-10, tests/test_gpu_stages.py)
so  NLL32 - NLL(G32) = factorisation + solve error,   NLL(G32) - NLL64 = Gram assembly in fp32 + input rounding.
usage: python scripts/fp32_error_split.py [model n] ...   (default: rating 4096, loadest 4096, rating 16384)"""
import json, math, sys
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd import _lib
from discontinuum_amd.backend import GPPlan
from tests.test_gpu_stages import make_case

dev = torch.device("cuda:0")
args = sys.argv[1:]
cases = [(args[i], int(args[i + 1])) for i in range(0, len(args), 2)] or [("rating", 4096), ("loadest", 4096), ("rating", 16384)]
import os
SEEDS = [int(v) for v in os.environ.get("SEEDS", "7").split(",")]
for model, n, seed in [(m, k, sd) for (m, k) in cases for sd in SEEDS]:
    d = 2 if model == "rating" else 3
    X, r, noise, theta = make_case(model, d, n, seed=seed, perturb=0.1)
    p64 = GPPlan(model, n, d, dtype=torch.float64, device=dev)
    p64.set_inputs(X.to(dev).contiguous())
    o64 = p64.fit_step(theta, r.to(dev), noise.to(dev))[0].cpu()
    del p64
    p = GPPlan(model, n, d, dtype=torch.float32, device=dev)
    p.set_inputs(X.float().to(dev).contiguous())
    r32, nz32 = r.float().to(dev).contiguous(), noise.float().to(dev).contiguous()
    p.stage_gram(theta, nz32)
    G = p.buffer(_lib.BUF_A)[:n, :n].double()
    G = torch.tril(G) + torch.tril(G, -1).T
    L = torch.linalg.cholesky(G)
    z = torch.linalg.solve_triangular(L, r32.double()[:, None], upper=False)[:, 0]
    quad_g, logdet_g = float(z @ z), float(2 * torch.log(torch.diagonal(L)).sum())
    nll_g = 0.5 * quad_g + 0.5 * logdet_g + 0.5 * n * math.log(2 * math.pi)
    ev = torch.linalg.eigvalsh(G)
    o32 = p.fit_step(theta, r32, nz32)[0].cpu().double()
    row = {"model": model, "n": n, "seed": seed, "cond": float(ev[-1] / ev[0]), "nll64": o64[0].item(), "nll_g32": nll_g, "nll32": o32[0].item(),
           "quad64": o64[1].item(), "quad_g32": quad_g, "quad32": o32[1].item(),
           "logdet64": o64[2].item(), "logdet_g32": logdet_g, "logdet32": o32[2].item()}
    row["err_total_rel_nll"] = abs(row["nll32"] - row["nll64"]) / abs(row["nll64"])
    row["err_factor_abs"] = row["nll32"] - nll_g
    row["err_gram_abs"] = nll_g - row["nll64"]
    row["err_factor_quad"] = row["quad32"] - quad_g
    row["err_factor_logdet"] = row["logdet32"] - logdet_g
    print(json.dumps(row), flush=True)
    del p, G, L
    torch.cuda.empty_cache()
