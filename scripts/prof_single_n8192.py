"""A few fit steps of ONE loadest site, n = 8192, d = 3, fp64 (the bench line's `single_site`) for rocprofv3."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0"); dt = torch.float64
X, r, noise, theta = bench.site("loadest", 8192, 3, 0)
p = GPPlan("loadest", 8192, 3, dtype=dt, device=dev); p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
for _ in range(4):
    out, _, _ = p.fit_step(theta, rd, nd)
torch.cuda.synchronize(); print(out[:4].cpu())
