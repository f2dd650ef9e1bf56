"""Kernel sequence of ONE fit step at small n (for rocprofv3 --kernel-trace): python small_n_trace.py [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
X = torch.as_tensor(np.concatenate([np.sort(rng.uniform(-16, 16, n))[:, None], rng.standard_normal((n, 2))], 1), device="cuda")
y = torch.randn(n, dtype=torch.float64, device="cuda"); noise = torch.full((n,), 0.01, dtype=torch.float64, device="cuda")
plan = GPPlan("loadest", n, 3); plan.set_inputs(X)
for _ in range(5):
    out = plan.fit_step([0.6931] * 11, y, noise)[0]
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200):
    out = plan.fit_step([0.6931] * 11, y, noise)[0]
    out.cpu()
print("ms per fit step (with result readback):", (time.perf_counter() - t0) * 5)
