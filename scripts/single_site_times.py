"""Fit-step and factorisation times of ONE site alone on the GPU at the sizes the panel chain matters for.
usage: python scripts/single_site_times.py [f64|f32] [n ...]   (A/B: run once per setting of DGP_CU_HOLDER etc.)"""
import json, sys
import torch
sys.path.insert(0, ".")
import bench
from discontinuum_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dtn = sys.argv[1] if len(sys.argv) > 1 else "f64"
ns = [int(a) for a in sys.argv[2:]] or [2048, 4096, 8192, 16384]
for n in ns:
    model, d = ("rating", 2) if (dtn == "f32" and n == 16384) else ("loadest", 3)
    steps = 20 if n <= 8192 else 6
    r = bench.time_config("x", model, n, d, dtn, 1, steps, 3, dev, _lib)
    print(json.dumps({"model": model, "n": n, "dtype": dtn, "ms": round(r["ms_per_step"], 3), "tflops": round(r["tflops"], 1), "ok": r["ok"],
                      "nll": r["nll_site0"], "stages_ms": {k: round(v, 3) for k, v in r["stages_ms"].items()}}), flush=True)
