"""A few fit steps of a batched loadest plan for rocprofv3: python3 scripts/prof_batch.py [n=4096] [sites=64] [steps=3] [dtype=f64]"""
import sys

import torch

sys.path.insert(0, ".")
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
dev = torch.device("cuda:0")
plan, th, r, noise = bench.make_plan("loadest", n, 3, dt, dev, S, 1 if S > 1 else 2)
for _ in range(steps):
    out = plan.fit_step(th, r, noise)[0]
torch.cuda.synchronize()
print(out.reshape(S, -1)[0, :4].cpu())
