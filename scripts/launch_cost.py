import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from discontinuum_amd.backend import GPPlan, _theta_array, _ptr, _stream
from discontinuum_amd import _lib
import ctypes as C
n = 300
rng = np.random.default_rng(0)
X = torch.as_tensor(np.concatenate([np.sort(rng.uniform(-16, 16, n))[:, None], rng.standard_normal((n, 2))], 1), device="cuda")
y = torch.randn(n, dtype=torch.float64, device="cuda"); noise = torch.full((n,), 0.01, dtype=torch.float64, device="cuda")
plan = GPPlan("loadest", n, 3); plan.set_inputs(X)
th = [0.6931] * 11
for _ in range(10): plan.fit_step(th, y, noise)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    plan.fit_step(th, y, noise)
    torch.cuda.synchronize()
t1 = time.perf_counter()
print("fit_step + sync:", (t1 - t0) / N * 1e6, "us")
# host-side cost only: enqueue N steps without sync (queue depth permitting)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): plan.fit_step(th, y, noise)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("enqueue only (python wrapper):", (t1 - t0) / 50 * 1e6, "us per call")
tha = _theta_array(th, plan.ntheta)
out = torch.empty(32, dtype=torch.float64, device="cuda"); dr = torch.empty(n, dtype=torch.float64, device="cuda"); dn = torch.empty(n, dtype=torch.float64, device="cuda")
args = (plan._h, tha, _ptr(y), _ptr(noise), _ptr(out), _ptr(dr), _ptr(dn), _stream())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): plan.lib.dgp_fit_step(*args)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("enqueue only (C call):", (t1 - t0) / 50 * 1e6, "us per call")
