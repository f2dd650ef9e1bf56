import os, sys, io, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.rating_gp import RatingGP
from discontinuum_amd.engines.base import ModelConfig
from tests.helpers import loadest_dataset, rating_dataset
cov, tgt = loadest_dataset(120, seed=1)
for cfg in (None, ModelConfig(transform="standard")):
    m = LoadestGP(model_config=cfg) if cfg else LoadestGP()
    m.fit(cov, tgt, iterations=15)
    mu, se = m.predict(cov); mu2, se2 = m.predict(cov, pred_noise=True)
    assert np.all(np.isfinite(mu.values)) and np.all(se2.values >= se.values - 1e-12)
    g = m.predict_grid("flow"); assert np.all(np.isfinite(g.values)), g.shape
    s = m.sample(cov, n=20); assert s.shape[0] == 20 and np.all(np.isfinite(s.values))
    m.fit(cov, tgt, iterations=5, resume=True)
    m.fit(cov, tgt, iterations=5, optimizer="adamw", learning_rate=0.01, early_stopping=True, patience=3, scheduler=False)
    buf = io.BytesIO(); m.save(buf); buf.seek(0)
    m2 = type(m).load(buf, cov, tgt)
    a, _ = m.predict(cov); b, _ = m2.predict(cov)
    assert np.allclose(a.values, b.values, rtol=1e-9), "save/load"
    print("loadest", cfg, "ok", float(mu.values.mean()))
rc, rt, ru = rating_dataset(90, seed=2)
r = RatingGP()
r.fit(rc, rt, target_unc=ru, iterations=25, early_stopping=True, patience=5)
r.fit(rc, rt, target_unc=ru, iterations=10, monotonic_penalty_weight=1.0, grid_size=16, monotonic_penalty_interval=2, resume=True)
mu, se = r.predict(rc); mu2, se2 = r.predict(rc, pred_noise=True)
assert np.all(np.isfinite(mu.values)) and np.all(np.isfinite(se2.values))
g = r.predict_grid("stage"); assert np.all(np.isfinite(g.values))
s = r.sample(rc, n=10); assert np.all(np.isfinite(s.values))
r2 = RatingGP(); r2.fit(rc, rt, iterations=5)  # no uncertainty supplied
buf = io.BytesIO(); r.save(buf); buf.seek(0); r3 = RatingGP.load(buf, rc, rt, ru)
assert np.allclose(r.predict(rc)[0].values, r3.predict(rc)[0].values, rtol=1e-9)
print("rating ok")
try:
    LoadestGP().predict(cov)
except RuntimeError as e:
    print("not fitted:", e)
try:
    LoadestGP().fit(cov, tgt, iterations=2, optimizer="sgd")
except ValueError as e:
    print("bad optimizer:", e)
