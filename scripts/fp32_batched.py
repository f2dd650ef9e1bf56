import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from discontinuum_amd import _lib
dev = torch.device("cuda:0")
for dtn, n, S in (("f32", 8192, 32), ("f32", 4096, 64), ("f32", 16384, 8), ("f64", 8192, 32)):
    dt = torch.float32 if dtn == "f32" else torch.float64
    plan, th, r, noise = bench.make_plan("loadest", n, 3, dt, dev, S, 1)
    plan.set_timing(True)
    for _ in range(3):
        plan.fit_step(th, r, noise)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            out = plan.fit_step(th, r, noise)[0]
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    ms = plan.get_timing()
    N = plan.N
    tf = S * float(N) ** 3 / best / 1e12
    peak = 157.3 if dtn == "f32" else 78.6
    fl = S * float(N) ** 3 / 3
    print(f"{S} x n={n} {dtn}: step {best*1e3:8.2f} ms {S/best:8.1f} fits/s  {tf:6.1f} TF = {tf/peak:.3f} of peak | gram {ms[_lib.TIME_GRAM]:.2f} potrf {ms[_lib.TIME_POTRF]:.2f} ({fl/ms[_lib.TIME_POTRF]/1e9/peak:.3f}) (bulk {ms[_lib.TIME_SYRK_SUM]:.2f}) trtri {ms[_lib.TIME_TRTRI]:.2f} ({fl/ms[_lib.TIME_TRTRI]/1e9/peak:.3f}) lauum {ms[_lib.TIME_LAUUM]:.2f} ({fl/ms[_lib.TIME_LAUUM]/1e9/peak:.3f}) solve {ms[_lib.TIME_SOLVE]:.2f} grad {ms[_lib.TIME_GRAD]:.2f}", flush=True)
    del plan; torch.cuda.empty_cache()
