import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.multisite_fit import fit_many
from discontinuum_amd.xr_compat import Dataset
def site(n, seed):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.choice(np.arange("1990-01-01", "2020-01-01", dtype="datetime64[D]"), n, replace=False)).astype("datetime64[ns]")
    flow = np.exp(rng.standard_normal(n)) * 10
    conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
    return Dataset({"flow": ("time", flow)}, coords={"time": t}), Dataset({"c": ("time", conc)}, coords={"time": t})["c"]
rng = np.random.default_rng(0)
for nsites in (64, 256):
    data = [site(int(rng.integers(200, 400)), 10 + i) for i in range(nsites)]
    fit_many([LoadestGP() for _ in data[:4]], data[:4], iterations=3)
    res = {}
    for iters in (1, 51):
        models = [LoadestGP() for _ in data]
        torch.cuda.synchronize(); t0 = time.perf_counter(); fit_many(models, data, iterations=iters); torch.cuda.synchronize()
        res[iters] = time.perf_counter() - t0
    print(f"{nsites} sites: setup+1 iteration {res[1]*1e3:.0f} ms; per iteration {(res[51]-res[1])/50*1e3:.2f} ms")
