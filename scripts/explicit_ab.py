"""Same-box A/B: closed-form host algebra (gp/explicit.py) against the autograd path, per training iteration."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.engines.hip import MarginalHIP
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.rating_gp import RatingGP
from tests.helpers import loadest_dataset, rating_dataset

def run(make, data, iters=150, **kw):
    m = make(); m.fit(*data, iterations=3, **kw)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(*data, iterations=iters, **kw); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best

for n in (300, 1500):
    ld, rd = loadest_dataset(n, seed=1), rating_dataset(n, seed=2)
    res = {}
    for label in ("explicit", "autograd", "explicit", "autograd"):
        MarginalHIP.explicit_host_algebra = label == "explicit"
        res.setdefault(label, []).append((run(LoadestGP, ld), run(RatingGP, (rd[0], rd[1]), target_unc=rd[2])))
    for label, v in res.items():
        print(f"n={n} {label:8s}: loadest {min(x[0] for x in v):.3f} ms/iter, rating {min(x[1] for x in v):.3f} ms/iter", flush=True)
