"""Single-GPU runs at the sizes of BASELINE configs 3 and 5 (fp32): n = 16384, 32768, 65536."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from discontinuum_amd.backend import GPPlan
from discontinuum_amd import _lib
dev = torch.device("cuda:0"); dt = torch.float32
for n in [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536]:
    d = 3; rng = np.random.default_rng(0)
    t = np.sort(rng.uniform(-16, 16, n)); X = np.concatenate([t[:, None], rng.standard_normal((n, d - 1))], 1)
    y = 0.8 * np.sin(2 * np.pi * t) + 0.5 * X[:, 1] + 0.3 * rng.standard_normal(n); y = (y - y.mean()) / y.std()
    Xd = torch.tensor(X, dtype=dt, device=dev); yd = torch.tensor(y, dtype=dt, device=dev)
    noise = torch.full((n,), 0.01, dtype=dt, device=dev); theta = [0.6931] * 11
    p = GPPlan("loadest", n, d, dtype=dt, device=dev); p.set_inputs(Xd)
    print(f"n={n}: workspace {p._ws.numel()/2**30:.1f} GiB", flush=True)
    out, a, _ = p.fit_step(theta, yd, noise); torch.cuda.synchronize()
    t0 = time.perf_counter(); out, a, _ = p.fit_step(theta, yd, noise); torch.cuda.synchronize(); dtm = time.perf_counter() - t0
    o = out.cpu()
    # size-independent check: K alpha = r through a fresh Gram build
    p.stage_gram(theta, noise); K = p.buffer(_lib.BUF_A)
    v = a.double()
    Ka = torch.zeros(n, dtype=torch.float64, device=dev)
    for lo in range(0, n, 8192):  # chunked symmetric matvec from the lower triangle
        blk = K[lo:lo + 8192, :n].double()
        rows = torch.arange(lo, min(lo + 8192, n), device=dev)[:, None]; cols = torch.arange(n, device=dev)[None, :]
        low = torch.where(cols <= rows, blk, torch.zeros((), dtype=torch.float64, device=dev))
        Ka[lo:lo + 8192] += low @ v
        strict = torch.where(cols < rows, blk, torch.zeros((), dtype=torch.float64, device=dev))
        Ka += strict.T @ v[lo:lo + 8192]
        del blk, low, strict
    res = (torch.linalg.norm(Ka - yd.double()) / torch.linalg.norm(yd.double())).item()
    print(f"n={n} fp32: {dtm*1e3:.1f} ms/fit  {p.N**3/dtm/1e12:.1f} TFLOP/s  info={int(o[3])} nll={o[0].item():.6g}  ||K alpha - r||/||r||={res:.2e}", flush=True)
    del p, K; torch.cuda.empty_cache()
