"""Throughput of independent n=4096 sites on ONE GPU with 1..4 plans driven on separate streams."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0"); dt = torch.float64
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 3
rng = np.random.default_rng(0)
def site(seed):
    r = np.random.default_rng(seed); t = np.sort(r.uniform(-16, 16, n))
    X = np.concatenate([t[:, None], r.standard_normal((n, d - 1))], 1)
    return torch.tensor(X, dtype=dt, device=dev), torch.tensor(r.standard_normal(n), dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931] * 11
sites = [site(i) for i in range(16)]
for conc in (1, 2, 3, 4, 6, 8):
    plans = [GPPlan("loadest", n, d, dtype=dt, device=dev) for _ in range(conc)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(conc)]
    def sweep():
        for i, (X, y) in enumerate(sites):
            p, s = plans[i % conc], streams[i % conc]
            with torch.cuda.stream(s):
                p.set_inputs(X); p.fit_step(theta, y, noise)
    sweep(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): sweep()
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / (reps * len(sites))
    print(f"n={n} concurrency={conc}: {dtm*1e3:.3f} ms/site  {1/dtm:.1f} sites/s  {n**3/dtm/1e12:.1f} TFLOP/s", flush=True)
    del plans
