"""Same-box A/B: the loop's lean Adam step against torch.optim.Adam.step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.engines import hip
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

lean = hip._lean_step
ld, rd = loadest_dataset(300, seed=1), rating_dataset(300, seed=2)
res = {}
for label in ("lean", "torch", "lean", "torch"):
    hip._lean_step = lean if label == "lean" else (lambda opt, decoupled: False)
    res.setdefault(label, []).append((run(LoadestGP, ld), run(RatingGP, (rd[0], rd[1]), target_unc=rd[2])))
for label, v in res.items():
    print(f"n=300 optimiser step {label:5s}: loadest {min(x[0] for x in v):.3f} ms/iter, rating {min(x[1] for x in v):.3f} ms/iter")
