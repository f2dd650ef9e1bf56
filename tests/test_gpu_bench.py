"""The bench contract on a small shape: one JSON line with the keys the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("extra", [[], ["--dtype", "f32", "--sites-per-gpu", "1"], ["--model", "rating", "--sites-per-gpu", "4"]])
def test_bench_prints_one_contract_line(extra, gpu_device):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n", "1024", "--steps", "3", "--warmup", "1", "--cpu-steps", "2"] + extra
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout
    rec = json.loads(lines[0])
    for key in REQUIRED:
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["higher_is_better"] is True and rec["vs_baseline"] is None and rec["data"] == "synthetic"
    assert rec["value"] > 0 and abs(rec["ms_per_step"] * rec["value"] / 1e3 - rec["config"]["fits_per_step"]) < 1e-6
    assert "workload" in rec["config"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    assert len(cpu["steps_s"]) == 2 and cpu["fp32"]["value"] > 0 and cpu["affinity_cores"] >= cpu["cores"]
    assert "configs" not in rec  # only the default n = 8192 run carries the other BASELINE configurations
    assert roof["gram_hbm"]["achieved"] > 0 and roof["gram_grad_hbm"]["achieved"] > 0
