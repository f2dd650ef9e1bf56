"""cProfile of fit_many (host share of a training iteration over many sites)."""
import os, sys, time, cProfile, pstats
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
nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(0)
data = [site(int(rng.integers(200, 400)), 10 + i) for i in range(nsites)]
models = [LoadestGP() for _ in data]
fit_many(models[:4], data[:4], iterations=3)
t0 = time.perf_counter(); fit_many(models, data, iterations=iters); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{nsites} sites, {iters} iterations: {dt:.2f} s = {dt / iters * 1e3:.1f} ms per iteration")
models = [LoadestGP() for _ in data]
pr = cProfile.Profile(); pr.enable(); fit_many(models, data, iterations=iters); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
