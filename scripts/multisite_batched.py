"""Throughput of independent sites on ONE GPU with one BATCHED plan (B sites per launch), n from argv."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0"); dt = torch.float64
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 3
def site(seed):
    r = np.random.default_rng(seed); t = np.sort(r.uniform(-16, 16, n))
    X = np.concatenate([t[:, None], r.standard_normal((n, d - 1))], 1)
    return torch.tensor(X, dtype=dt, device=dev), torch.tensor(r.standard_normal(n), dtype=dt, device=dev)
nsites = (64 if n <= 4096 else 16) if n > 1024 else 256
sites = [site(i) for i in range(nsites)]
theta1 = [0.6931] * 11
for B in (((1, 8, 16, 32, 64) if n <= 4096 else (1, 2, 4, 8)) if n > 1024 else (1, 8, 32, 128, 256)):
    p = GPPlan("loadest", n, d, dtype=dt, device=dev, lookahead=2 if B == 1 else 1, batch=B)
    noise = torch.full((B, n) if B > 1 else (n,), 0.01, dtype=dt, device=dev)
    groups = []
    for g in range(0, nsites, B):
        X = torch.stack([sites[g + b][0] for b in range(B)]).contiguous() if B > 1 else sites[g][0]
        y = torch.stack([sites[g + b][1] for b in range(B)]).contiguous() if B > 1 else sites[g][1]
        groups.append((X, y))
    def sweep():
        for X, y in groups:
            p.set_inputs(X); p.fit_step(theta1 * B, y, noise)
    sweep(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): sweep()
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / (reps * nsites)
    print(f"n={n} batch={B}: {dtm*1e3:.3f} ms/site  {1/dtm:.1f} sites/s  {n**3/dtm/1e12:.1f} TFLOP/s", flush=True)
    del p
