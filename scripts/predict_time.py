"""Inference at the sizes the reference's workflow uses (a daily grid over a 31-year record: m = 11323 points):
plan-level predict / posterior factor at n = 8192, and engine-level predict / sample at n = 300."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.backend import GPPlan
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.xr_compat import Dataset

def loadest_inputs(n, d, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16, 16, n)); cov = rng.standard_normal((n, d - 1))
    y = 0.8 * np.sin(2 * np.pi * t) + 0.5 * cov[:, 0] + 0.1 * t / 16 + 0.3 * rng.standard_normal(n)
    return np.concatenate([t[:, None], cov], 1), (y - y.mean()) / y.std(), np.full(n, 0.01)

def loadest_theta(d):
    return [0.6931] * 11

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

m = 11323
for dtype in (torch.float64, torch.float32):
    for n in (8192, 2048, 300):
        X, y, noise = loadest_inputs(n, 3, seed=0)
        Xs, _, _ = loadest_inputs(m, 3, seed=1)
        dev = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda")
        plan = GPPlan("loadest", n, 3, dtype=dtype)
        plan.set_inputs(dev(X))
        th = loadest_theta(3)
        plan.factorize(th, dev(y), dev(noise))
        xs = dev(Xs)
        tp = timed(lambda: plan.predict(th, xs))
        line = f"{str(dtype)[6:]} n={n:5d} m={m}: predict {tp:8.2f} ms ({n * n * m / tp / 1e9:6.1f} TFLOP/s on n^2 m)"
        if n <= 2048:
            tf = timed(lambda: plan.posterior_factor(th, xs), reps=1)
            line += f" | posterior cov + factor {tf:8.2f} ms"
        print(line, flush=True)
        del plan

def site(n, seed):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.choice(np.arange("1990-01-01", "2021-01-01", dtype="datetime64[D]"), n, replace=False)).astype("datetime64[ns]")
    flow = np.exp(rng.standard_normal(n)) * 10
    conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
    return Dataset({"flow": ("time", flow)}, coords={"time": t}), Dataset({"c": ("time", conc)}, coords={"time": t})["c"]

cov, tgt = site(300, 3)
model = LoadestGP()
model.fit(cov, tgt, iterations=20)
rng = np.random.default_rng(5)
days = np.arange("1990-01-01", "2021-01-01", dtype="datetime64[D]").astype("datetime64[ns]")
daily = Dataset({"flow": ("time", np.exp(rng.standard_normal(len(days))) * 10)}, coords={"time": days})
tp = timed(lambda: model.predict(daily))
ts = timed(lambda: model.sample(daily, n=1000), reps=1)
print(f"engine n=300, m={len(days)}: predict {tp:.1f} ms | sample(n=1000) {ts:.1f} ms")
