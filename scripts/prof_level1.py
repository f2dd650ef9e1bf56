"""A few single-site fit steps at lookahead level 1 (no early inverse) for kernel traces of the plain schedule."""
import sys
import numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
from oracle.gp_oracle import synth_loadest
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0"); dt = torch.float64
X, y = synth_loadest(n, 3, 0)
X = torch.tensor(X, dtype=dt, device=dev).contiguous(); y = torch.tensor(y, dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931471805599453] * 11
p = GPPlan("loadest", n, 3, dtype=dt, device=dev, lookahead=level); p.set_inputs(X)
for _ in range(4): p.fit_step(theta, y, noise)
torch.cuda.synchronize()
