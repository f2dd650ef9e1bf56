"""A few fit steps of BASELINE config 3 (rating-gp kernel, n = 16384, d = 2, fp32, one site) for rocprofv3."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0"); dt = torch.float32
X, r, noise, theta = bench.site("rating", 16384, 2, 0)
p = GPPlan("rating", 16384, 2, dtype=dt, device=dev); p.set_inputs(torch.tensor(X, dtype=dt, device=dev).contiguous())
rd, nd = torch.tensor(r, dtype=dt, device=dev), torch.tensor(noise, dtype=dt, device=dev)
for _ in range(4):
    out, _, _ = p.fit_step(theta, rd, nd)
torch.cuda.synchronize(); print(out[:4].cpu())
