import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.loadest_gp import LoadestGP
from tests.helpers import loadest_dataset
cov, tgt = loadest_dataset(300, seed=1)
m = LoadestGP(); m.fit(cov, tgt, iterations=10)
m.predict(cov); m.predict(cov)
t0 = time.perf_counter()
for _ in range(20): m.predict(cov)
torch.cuda.synchronize(); print("predict ms:", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m.predict(cov)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
