"""Per-stage timing of the fit step (HIP events on torch's current stream)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
from discontinuum_amd import _lib

def synth(n, d, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(-16, 16, n)); cov = rng.standard_normal((n, d - 1))
    y = 0.8*np.sin(2*np.pi*t) + 0.5*cov[:, 0] + 0.1*t/16 + 0.3*rng.standard_normal(n)
    y = (y - y.mean())/y.std()
    return np.concatenate([t[:, None], cov], 1), y

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = 3
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
dev = torch.device("cuda:0")
X, y = synth(n, d)
X = torch.tensor(X, dtype=dt, device=dev); y = torch.tensor(y, dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev)
theta = [0.6931]*11
for la in (True, False):
    p = GPPlan("loadest", n, d, dtype=dt, device=dev, lookahead=la)
    p.set_inputs(X)
    N = p.N; fl = N**3/3
    def potrf(): p.stage_gram(theta, noise); p.stage_potrf()
    tg = timeit(lambda: p.stage_gram(theta, noise))
    tp = timeit(potrf) - tg
    tt = timeit(p.stage_trtri)
    tl = timeit(p.stage_lauum)
    ts = timeit(lambda: p.stage_solve(y))
    tgr = timeit(lambda: p.stage_grad(theta))
    tf = timeit(lambda: p.fit_step(theta, y, noise))
    print(f"n={n} {dt} lookahead={la}: gram {tg:.3f} ms | potrf {tp:.3f} ms ({fl/tp/1e9:.1f} TF) | trtri {tt:.3f} ({fl/tt/1e9:.1f} TF) | "
          f"lauum {tl:.3f} ({fl/tl/1e9:.1f} TF) | solve {ts:.3f} | grad {tgr:.3f} | fit_step {tf:.3f} ms = {1e3/tf:.1f} fits/s ({N**3/tf/1e9:.1f} TF)")
    del p
