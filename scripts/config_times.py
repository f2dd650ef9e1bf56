"""Fit-step timings for the BASELINE.json configs that fit one GPU (diagnostic, not the bench contract)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
from discontinuum_amd import _lib
dev = torch.device("cuda:0")

def data(model, n, d, dt, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16, 16, n))
    if model == "loadest":
        X = np.concatenate([t[:, None], rng.standard_normal((n, d - 1))], 1); noise = np.full(n, 0.01)
        theta = [0.6931] * (2 * d + 5)
    else:
        s = 1 + rng.beta(2, 5, n); X = np.stack([t, s], 1); noise = rng.uniform(1e-3, 4e-3, n) + 0.01
        theta = [float(np.median(s))] + [0.6931] * 15
    y = rng.standard_normal(n)
    return (torch.tensor(X, dtype=dt, device=dev), torch.tensor(y, dtype=dt, device=dev),
            torch.tensor(noise, dtype=dt, device=dev), theta)

def run(model, n, d, dt, reps=5, sites=1):
    X, y, noise, theta = data(model, n, d, dt)
    p = GPPlan(model, n, d, dtype=dt, device=dev); p.set_inputs(X)
    out, _, _ = p.fit_step(theta, y, noise); torch.cuda.synchronize()
    assert out[_lib.OUT_INFO].item() == 0, out[:4]
    t0 = time.perf_counter()
    for _ in range(reps):
        for _ in range(sites):
            if sites > 1: p.set_inputs(X)
            out, _, _ = p.fit_step(theta, y, noise)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / (reps * sites)
    N = p.N
    print(f"{model:8s} n={n:6d} d={d} {str(dt)[6:]:8s}: {dtm*1e3:8.3f} ms/fit  {1/dtm:8.1f} fits/s  {N**3/dtm/1e12:6.1f} TFLOP/s  nll={out[0].item():.6g}", flush=True)
    del p; torch.cuda.empty_cache()

run("loadest", 300, 2, torch.float64, reps=50)
run("loadest", 300, 2, torch.float32, reps=50)
run("loadest", 1024, 3, torch.float64, reps=30)
run("loadest", 4096, 3, torch.float64, reps=10, sites=8)
run("loadest", 8192, 3, torch.float64)
run("loadest", 8192, 3, torch.float32)
run("rating", 8192, 2, torch.float64)
run("rating", 16384, 2, torch.float32, reps=3)
run("loadest", 16384, 3, torch.float64, reps=2)
