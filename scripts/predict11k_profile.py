import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.xr_compat import Dataset
from tests.helpers import loadest_dataset
cov, tgt = loadest_dataset(300, seed=1)
m = LoadestGP(); m.fit(cov, tgt, iterations=10)
rng = np.random.default_rng(5)
days = np.arange("1990-01-01", "2021-01-01", dtype="datetime64[D]").astype("datetime64[ns]")
daily = Dataset({"flow": ("time", np.exp(rng.standard_normal(len(days))) * 10)}, coords={"time": days})
m.predict(daily); m.predict(daily)
t0 = time.perf_counter()
for _ in range(10): m.predict(daily)
torch.cuda.synchronize(); print("predict ms:", (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): m.predict(daily)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
