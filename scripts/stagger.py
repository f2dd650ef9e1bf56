"""Does a deliberate half-period phase offset between two sites on one GPU improve throughput?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0"); dt = torch.float64
n, d = 8192, 3
def site(seed):
    r = np.random.default_rng(seed); t = np.sort(r.uniform(-16, 16, n))
    X = np.concatenate([t[:, None], r.standard_normal((n, d - 1))], 1)
    return torch.tensor(X, dtype=dt, device=dev), torch.tensor(r.standard_normal(n), dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931] * 11
plans = [GPPlan("loadest", n, d, dtype=dt, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
data = [site(i) for i in range(2)]
for p, (X, y) in zip(plans, data): p.set_inputs(X)
def run(offset_ms, steps=12):
    torch.cuda.synchronize()
    # warm
    for i in range(2):
        with torch.cuda.stream(streams[i]): plans[i].fit_step(theta, data[i][1], noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset_ms > 0:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(offset_ms * 1e-3 * 2.4e9))  # delay stream 1 by ~offset
    for _ in range(steps):
        for i in range(2):
            with torch.cuda.stream(streams[i]): plans[i].fit_step(theta, data[i][1], noise)
    torch.cuda.synchronize(); dtm = time.perf_counter() - t0
    print(f"offset {offset_ms:5.1f} ms: {2*steps/dtm:.1f} fits/s  ({dtm/steps*1e3:.2f} ms per pair)", flush=True)
for off in (0.0, 3.0, 7.0, 10.0, 0.0, 7.0):
    run(off)
