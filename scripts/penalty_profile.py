import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.rating_gp import RatingGP
from discontinuum_amd.xr_compat import Dataset, DataArray
def rating_data(n, seed=0):
    rng = np.random.default_rng(seed)
    t = (np.datetime64("2005-01-01") + np.sort(rng.choice(365 * 15, n, replace=False)).astype("timedelta64[D]")).astype("datetime64[ns]")
    stage = 1.0 + 3.0 * rng.beta(2, 5, n)
    q = np.exp(1.6 * np.log(stage) + 0.05 * rng.standard_normal(n))
    cov = Dataset({"stage": ("time", stage)}, coords={"time": t})
    return cov, DataArray(q, dims=("time",), coords={"time": t}, name="q"), DataArray(np.full(n, 1.05), dims=("time",), coords={"time": t}, name="q_unc")
cov, tgt, unc = rating_data(300)
m = RatingGP(); m.fit(cov, tgt, target_unc=unc, iterations=3, monotonic_penalty_weight=1.0)
pr = cProfile.Profile(); pr.enable(); m.fit(cov, tgt, target_unc=unc, iterations=100, monotonic_penalty_weight=1.0); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
