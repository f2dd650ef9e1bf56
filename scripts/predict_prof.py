"""One predict at n = 8192, m = 11323 (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
n, m, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 11323, 3
dtype = torch.float32 if "f32" in sys.argv else torch.float64
rng = np.random.default_rng(0)
def pts(k):
    return torch.as_tensor(np.concatenate([np.sort(rng.uniform(-16, 16, k))[:, None], rng.standard_normal((k, d - 1))], 1), dtype=dtype, device="cuda")
X, Xs = pts(n), pts(m)
y = torch.randn(n, dtype=dtype, device="cuda"); noise = torch.full((n,), 0.01, dtype=dtype, device="cuda")
plan = GPPlan("loadest", n, d, dtype=dtype); plan.set_inputs(X)
th = [0.6931] * 11
plan.factorize(th, y, noise)
for _ in range(4):
    plan.predict(th, Xs)
torch.cuda.synchronize()
