"""Stage breakdown of the batched fit step (the bench's headline plan) under the library's tuning knobs.
usage: python scripts/batched_stages.py [n [sites [dtype [steps]]]]   (knobs: DGP_POTRF_LL, DGP_LAUUM_SUPER, DGP_GROUP ...)
Prints one line: total ms per step, fits/s, per-stage ms / TFLOP/s, NLL of site 0 (to compare variants)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from discontinuum_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dtn = sys.argv[3] if len(sys.argv) > 3 else "f64"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
model = os.environ.get("MODEL", "loadest")
dt = torch.float64 if dtn == "f64" else torch.float32
dev = torch.device("cuda:0")
level = int(os.environ.get("LEVEL", 1 if S > 1 else 2))
plan, th, r, noise = bench.make_plan(model, n, 3 if model == "loadest" else 2, dt, dev, S, level)
plan.set_timing(True)
for _ in range(2):
    out = plan.fit_step(th, r, noise)[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = plan.fit_step(th, r, noise)[0]
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
rep = bench.stage_report(plan, S, dtn, level, _lib)
o = out.reshape(S, -1)[0].cpu()
knobs = {k: v for k, v in os.environ.items() if k.startswith("DGP_")}
print(f"{knobs} n={n} S={S} {dtn}: {ms:.2f} ms/step {S/ms*1e3:.2f} fits/s {S*plan.N**3/ms/1e9:.1f} TF | "
      + " ".join(f"{k}={v:.2f}" for k, v in rep["stages_ms"].items())
      + f" | potrf {rep['potrf_stage_tflops']:.1f} TF, bulk {rep['stages_tflops']['syrk_kernel'] or 0:.1f}, trtri {rep['stages_tflops']['trtri_level_kernel'] or 0:.1f}, "
      f"lauum {rep['stages_tflops']['lauum_kernel']:.1f} | gram {rep['gram_hbm']['achieved']:.0f} GB/s grad {rep['gram_grad_hbm']['achieved']:.0f} GB/s | "
      f"nll0={o[0].item():.12g} info={int(o[3])} g0={o[4].item():.10g}", flush=True)
