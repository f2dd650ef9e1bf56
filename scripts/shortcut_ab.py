"""Same-box A/B of the host-side mean / noise shortcut (gradients from the result row) against device-side autograd."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.engines.hip import MarginalHIP
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.rating_gp import RatingGP
from discontinuum_amd.rating_gp.models import RatingGPMarginalHIP
from tests.helpers import loadest_dataset, rating_dataset

def run(make, data, iters=150, **kw):
    m = make(); m.fit(*data, iterations=3, **kw)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.fit(*data, iterations=iters, **kw); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best

orig_base, orig_rating = MarginalHIP._mean_shortcut, RatingGPMarginalHIP._mean_shortcut
for n in (300, 1500):
    ld, rd = loadest_dataset(n, seed=1), rating_dataset(n, seed=2)
    res = {}
    for label in ("shortcut", "autograd", "shortcut", "autograd"):
        if label == "autograd":
            MarginalHIP._mean_shortcut = lambda self: None
            RatingGPMarginalHIP._mean_shortcut = lambda self: None
        else:
            MarginalHIP._mean_shortcut, RatingGPMarginalHIP._mean_shortcut = orig_base, orig_rating
        res.setdefault(label, []).append((run(LoadestGP, ld), run(RatingGP, (rd[0], rd[1]), target_unc=rd[2])))
    for label, v in res.items():
        print(f"n={n} {label:9s}: loadest {min(x[0] for x in v):.3f} ms/iter, rating {min(x[1] for x in v):.3f} ms/iter", flush=True)
