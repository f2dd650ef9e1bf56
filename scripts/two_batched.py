"""n = 8192: ONE batched plan of 32 sites against several smaller batched plans on separate streams (fits/s)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from discontinuum_amd.backend import GPPlan
from oracle.gp_oracle import synth_loadest
dev = torch.device("cuda:0"); dt = torch.float64; n, d = 8192, 3
def mk(B, seed0):
    Xs, ys = zip(*[synth_loadest(n, d, seed0 + b) for b in range(B)])
    p = GPPlan("loadest", n, d, dtype=dt, device=dev, lookahead=1, batch=B)
    p.set_inputs(torch.tensor(np.stack(Xs), device=dev).contiguous())
    return p, torch.tensor(np.stack(ys), device=dev).contiguous(), torch.full((B, n), 0.01, dtype=dt, device=dev), [0.6931471805599453] * (11 * B)
for cfg in ((32,), (16, 16), (8, 8, 8, 8), (24, 24)):
    plans = [mk(B, 10 * i) for i, B in enumerate(cfg)]
    streams = [torch.cuda.Stream(device=dev) for _ in cfg]
    def step():
        for (p, y, nz, th), st in zip(plans, streams):
            with torch.cuda.stream(st):
                p.fit_step(th, y, nz)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 8
    for _ in range(reps): step()
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / reps
    print(f"plans {cfg}: {dtm*1e3:.2f} ms per step, {sum(cfg)/dtm:.1f} fits/s", flush=True)
    del plans
