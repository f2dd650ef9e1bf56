"""Wall time of the first fit / predict of a fresh process against the following ones (n = 300, 100 iterations)."""
import os, sys, time
t_import0 = time.perf_counter()
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.loadest_gp import LoadestGP
from tests.helpers import loadest_dataset
t_import = time.perf_counter() - t_import0
cov, tgt = loadest_dataset(300, seed=1)
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter(); m = LoadestGP(); m.fit(cov, tgt, iterations=100); torch.cuda.synchronize(); t1 = time.perf_counter()
    mu, se = m.predict(cov); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"run {k}: fit(100 iterations) {1e3 * (t1 - t0):.1f} ms, first predict {1e3 * (t2 - t1):.1f} ms")
print(f"imports {t_import:.2f} s")
