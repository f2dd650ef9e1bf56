"""Where sample(m = 11323, 1000 draws) spends its time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["TQDM_DISABLE"] = "1"
from discontinuum_amd.loadest_gp import LoadestGP
from discontinuum_amd.xr_compat import Dataset
def site(n, seed):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.choice(np.arange("1990-01-01", "2021-01-01", dtype="datetime64[D]"), n, replace=False)).astype("datetime64[ns]")
    flow = np.exp(rng.standard_normal(n)) * 10
    conc = np.exp(0.3 * np.log(flow) + 0.2 * rng.standard_normal(n))
    return Dataset({"flow": ("time", flow)}, coords={"time": t}), Dataset({"c": ("time", conc)}, coords={"time": t})["c"]
cov, tgt = site(300, 3)
model = LoadestGP(); model.fit(cov, tgt, iterations=20)
rng = np.random.default_rng(5)
days = np.arange("1990-01-01", "2021-01-01", dtype="datetime64[D]").astype("datetime64[ns]")
daily = Dataset({"flow": ("time", np.exp(rng.standard_normal(len(days))) * 10)}, coords={"time": days})
model.sample(daily, n=1000)
def tick(label, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); print(f"  {label:28s} {(t - t0) * 1e3:8.2f} ms"); return t
for rep in range(2):
    t = time.perf_counter()
    Xnew = torch.tensor(model.dm.Xnew(daily), dtype=model.dtype).to(model.device).contiguous(); t = tick("Xnew (host transform+upload)", t)
    model._ensure_factor(); t = tick("ensure_factor", t)
    mean, cov = model._plan.posterior_cov(model._factor_theta, Xnew); t = tick("posterior_cov (n^2 m + m^2 n)", t)
    Lbuf, jitter = model._plan.psd_safe_factor(cov, Xnew.shape[0]); t = tick(f"psd_safe_factor (jitter {jitter:g})", t)
    sim = model._plan.sample_draws(Lbuf, Xnew.shape[0], mean, 1000); t = tick("sample_draws (randn + MFMA)", t)
    host = sim.reshape(-1).cpu().numpy(); t = tick("to host", t)
    temp = model.dm.y_t(host); t = tick("y_t (numpy)", t)
    t0 = time.perf_counter(); model.sample(daily, n=1000); torch.cuda.synchronize()
    print(f"  sample(n=1000) end to end     {(time.perf_counter() - t0) * 1e3:8.2f} ms")
    print()
