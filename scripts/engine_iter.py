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
