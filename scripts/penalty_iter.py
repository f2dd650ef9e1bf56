"""Cost of one training iteration WITH the rating-gp monotonicity penalty: RatingGP.fit against fit_many (1 and 16 sites)."""
import contextlib, io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.multisite_fit import fit_many
from discontinuum_amd.rating_gp import RatingGP
from tests.helpers import rating_dataset

def bent(rec):  # a rating that bends down over the upper half of the stages: the penalty is ACTIVE (non-zero gradient)
    cov, tgt, unc = rec
    stage = cov["stage"].values
    v = tgt.values * np.exp(-1.5 * np.maximum(stage - np.median(stage), 0.0) ** 2)
    return cov, type(tgt)(v, dims=tgt.dims, coords={"time": tgt.coords["time"]}, name=tgt.name, attrs=dict(tgt.attrs)), unc


for n in (300, 2048):
    data = [bent(rating_dataset(n, seed=i)) for i in range(16)]
    for label, fn in (("RatingGP.fit, no penalty", lambda it: RatingGP().fit(*data[0][:2], target_unc=data[0][2], iterations=it)),
                      ("RatingGP.fit, penalty", lambda it: RatingGP().fit(*data[0][:2], target_unc=data[0][2], iterations=it, monotonic_penalty_weight=1.0)),
                      ("fit_many 1 site, no penalty", lambda it: fit_many([RatingGP()], data[:1], iterations=it)),
                      ("fit_many 1 site, penalty", lambda it: fit_many([RatingGP()], data[:1], iterations=it, monotonic_penalty_weight=1.0)),
                      ("fit_many 16 sites, penalty", lambda it: fit_many([RatingGP() for _ in range(16)], data, iterations=it, monotonic_penalty_weight=1.0))):
        with contextlib.redirect_stderr(io.StringIO()):
            fn(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(50)
            torch.cuda.synchronize()
            t50 = time.perf_counter() - t0
            t0 = time.perf_counter()
            fn(10)
            torch.cuda.synchronize()
            t10 = time.perf_counter() - t0
        print(f"n={n} {label:30s}: {(t50 - t10) / 40 * 1e3:7.2f} ms per iteration", flush=True)
