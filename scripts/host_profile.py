"""cProfile of the engine's training loop at small n (host overhead per iteration)."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, ".")
os.environ["TQDM_DISABLE"] = "1"
import numpy as np, torch
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.xr_compat import Dataset, DataArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
t = (np.datetime64("1990-01-01") + np.sort(rng.choice(365 * 30, n, replace=False)).astype("timedelta64[D]")).astype("datetime64[ns]")
flow = np.exp(rng.standard_normal(n)) * 10
conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
cov, tgt = Dataset({"flow": ("time", flow)}, coords={"time": t}), DataArray(conc, dims=("time",), coords={"time": t}, name="c")
m = LoadestGP()
m.fit(cov, tgt, iterations=5)
torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(cov, tgt, iterations=200); torch.cuda.synchronize()
print("ms per iteration:", (time.perf_counter() - t0) * 5)
pr = cProfile.Profile(); pr.enable(); m.fit(cov, tgt, iterations=200); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
