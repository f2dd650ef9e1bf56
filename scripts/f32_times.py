"""fp32 fit-step times of the shapes that matter (config 3, config 5 on one GPU, n = 8192 / 4096 single site, a batch):
run once per setting of DGP_F32_DIAG64 to A/B the mixed-precision panel.  usage: python scripts/f32_times.py [big]"""
import json, sys
import torch
sys.path.insert(0, ".")
import bench
from discontinuum_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = _lib
rows = [("rating", 16384, 2, 1, 8), ("loadest", 8192, 3, 1, 10), ("loadest", 4096, 3, 1, 20), ("rating", 2048, 2, 1, 30), ("loadest", 4096, 3, 16, 5)]
if len(sys.argv) > 1:
    rows.append(("loadest", 65536, 3, 1, 2))
for model, n, d, S, steps in rows:
    r = bench.time_config("x", model, n, d, "f32", S, steps, 2, dev, lib)
    print(json.dumps({"model": model, "n": n, "S": S, "ms": round(r["ms_per_step"], 3), "tflops": round(r["tflops"], 1),
                      "stages_ms": {k: round(v, 3) for k, v in r["stages_ms"].items()}}), flush=True)
