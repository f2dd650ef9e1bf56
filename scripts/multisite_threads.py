"""Independent sites on ONE GPU: one host thread per plan/stream (ctypes drops the GIL inside the C ABI call),
to see whether the sweep is bound by the host's launch rate rather than by the GPU."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
dev = torch.device("cuda:0"); dt = torch.float64
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 3
def site(seed):
    r = np.random.default_rng(seed); t = np.sort(r.uniform(-16, 16, n))
    X = np.concatenate([t[:, None], r.standard_normal((n, d - 1))], 1)
    return torch.tensor(X, dtype=dt, device=dev), torch.tensor(r.standard_normal(n), dtype=dt, device=dev)
noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931] * 11
sites = [site(i) for i in range(24)]
for conc in (1, 2, 3, 4, 6):
    plans = [GPPlan("loadest", n, d, dtype=dt, device=dev) for _ in range(conc)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(conc)]
    def worker(w, reps):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[w]):
            for _ in range(reps):
                for i in range(w, len(sites), conc):
                    X, y = sites[i]
                    plans[w].set_inputs(X); plans[w].fit_step(theta, y, noise)
    def sweep(reps):
        th = [threading.Thread(target=worker, args=(w, reps)) for w in range(conc)]
        [t.start() for t in th]; [t.join() for t in th]
    sweep(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    sweep(reps); torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / (reps * len(sites))
    print(f"n={n} threads={conc}: {dtm*1e3:.3f} ms/site  {1/dtm:.1f} sites/s  {n**3/dtm/1e12:.1f} TFLOP/s", flush=True)
    del plans
