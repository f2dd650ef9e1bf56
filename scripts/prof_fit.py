"""A few fit steps at the bench shape, for rocprofv3."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
la = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # lookahead level 0 / 1 / 2
d = 3; dev = torch.device("cuda:0"); dt = torch.float64
rng = np.random.default_rng(0)
t = np.sort(rng.uniform(-16, 16, n)); cov = rng.standard_normal((n, d-1))
X = torch.tensor(np.concatenate([t[:, None], cov], 1), dtype=dt, device=dev)
y = torch.tensor(rng.standard_normal(n), dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev)
p = GPPlan("loadest", n, d, dtype=dt, device=dev, lookahead=la); p.set_inputs(X)
for _ in range(3):
    out, _, _ = p.fit_step([0.6931]*11, y, noise)
torch.cuda.synchronize(); print(out[:4].cpu())
