"""Per-iteration wall time of the full engine loop (host logic + device step) at several n."""
import sys, time, io, contextlib
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.rating_gp import RatingGP
from discontinuum_amd.xr_compat import Dataset, DataArray
import os; os.environ["TQDM_DISABLE"] = "1"

def loadest_data(n, seed=0):
    rng = np.random.default_rng(seed)
    t = (np.datetime64("1990-01-01") + np.sort(rng.choice(365 * 30, n, replace=False)).astype("timedelta64[D]")).astype("datetime64[ns]")
    flow = np.exp(rng.standard_normal(n)) * 10
    conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
    return Dataset({"flow": ("time", flow)}, coords={"time": t}), DataArray(conc, dims=("time",), coords={"time": t}, name="c")

for n, iters in ((300, 100), (2048, 50), (8192, 20)):
    cov, tgt = loadest_data(n)
    m = LoadestGP()
    m.fit(cov, tgt, iterations=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit(cov, tgt, iterations=iters)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    t1 = time.perf_counter(); mu, se = m.predict(cov); torch.cuda.synchronize(); tp = time.perf_counter() - t1
    print(f"loadest n={n}: {dt*1e3:.2f} ms per training iteration ({iters} its), predict(m=n) {tp*1e3:.1f} ms", flush=True)

def rating_data(n, seed=0):
    rng = np.random.default_rng(seed)
    t = (np.datetime64("2005-01-01") + np.sort(rng.choice(365 * 15, n, replace=False)).astype("timedelta64[D]")).astype("datetime64[ns]")
    stage = 1.0 + 3.0 * rng.beta(2, 5, n)
    q = np.exp(1.6 * np.log(stage) + 0.05 * rng.standard_normal(n))
    cov = Dataset({"stage": ("time", stage)}, coords={"time": t})
    return cov, DataArray(q, dims=("time",), coords={"time": t}, name="q"), DataArray(np.full(n, 1.05), dims=("time",), coords={"time": t}, name="q_unc")

for n, iters in ((300, 100), (2048, 50)):
    cov, tgt, unc = rating_data(n)
    m = RatingGP()
    m.fit(cov, tgt, target_unc=unc, iterations=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit(cov, tgt, target_unc=unc, iterations=iters)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    m.fit(cov, tgt, target_unc=unc, iterations=iters, monotonic_penalty_weight=1.0)
    torch.cuda.synchronize(); dp = (time.perf_counter() - t0) / iters
    print(f"rating n={n}: {dt*1e3:.2f} ms per training iteration, {dp*1e3:.2f} ms with the monotonicity penalty (64 grid points)", flush=True)
