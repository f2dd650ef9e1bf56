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
    # the clock-normalised fraction: the shader clock the chip held under the dominant kernel's stage (in-kernel stamps of a
    # separate probe launch, dgp_debug_clock_probe) and the fraction of the peak at that clock
    for key in ("clock_mhz", "frac_at_clock", "clock_probe"):
        assert key in roof, key
    assert 500 < roof["clock_mhz"] < 3000, roof["clock_mhz"]
    assert abs(roof["frac_at_clock"] - roof["frac"] * 2400.0 / roof["clock_mhz"]) < 1e-9
    assert roof["clock_probe"]["whole_steps"]["workgroups"] >= 8
    assert rec["cpu_baseline"]["sweep_minimum_interior"] in (True, False) and len(rec["cpu_baseline"]["thread_sweep_half_n_s"]) >= 1
    # the box's state during the timed region: null (rocm-smi unavailable / region shorter than a sample) or clock + power
    assert "gpu_state" in rec
    if rec["gpu_state"] is not None:
        assert 500 < rec["gpu_state"]["sclk_mhz"]["mean"] < 3000 and rec["gpu_state"]["samples"] >= 1


# ---------------------------------------------------------------------------------------------------------------------
# The driver's multi-GPU command line, rehearsed on the one GPU of the test box: torch.distributed.run starts two ranks
# (the launcher itself never touches the GPU) that share the card over gloo (DGP_BENCH_BACKEND; RCCL refuses two ranks
# on one device).  Checks the ONE contract line of rank 0 for both entry points.
def _torchrun(extra, nproc=2):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DGP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout
    return json.loads(lines[0])


def test_torchrun_two_ranks_headline_line(gpu_device):
    S = 3
    rec = _torchrun(["--size", "1024", "--steps", "2", "--warmup", "1", "--sites-per-gpu", str(S)])
    for key in REQUIRED:
        assert key in rec, key
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["fits_per_step"] == 2 * S and rec["config"]["sites_per_gpu"] == S
    assert rec["value"] > 0 and abs(rec["ms_per_step"] * rec["value"] / 1e3 - 2 * S) < 1e-6
    assert rec["cpu_baseline"] is None  # rank 0 at N = 1 only
    assert 0 < rec["roofline"]["frac"] < 1


def test_torchrun_two_ranks_config5_line(gpu_device):
    rec = _torchrun(["--config", "5", "--size", "2500", "--dtype", "f32", "--steps", "2", "--warmup", "1"])
    for key in REQUIRED:
        assert key in rec, key
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["dtype"] == "f32"
    assert rec["config"]["info"] == 0 and rec["config"]["backend"] == "gloo"
    roof = rec["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["kernel"].startswith("slab_")
    ng = -(-2500 // 512)
    assert roof["collectives_issued"] == {"broadcast": 2 * ng, "all_reduce": 5, "all_gather": 1}  # fp32: + the refinement's two sums


def test_bench_forced_collectives_enter_rccl_with_one_rank(gpu_device):
    """`DGP_DIST_FORCE_COLLECTIVES=1 python bench.py --config 5`: one rank, backend nccl, every collective issued."""
    env = dict(os.environ, DGP_DIST_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DGP_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--size", "3000", "--dtype", "f32", "--steps", "2", "--warmup", "1"]
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    rec = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["config"]["backend"] == "nccl" and rec["config"]["info"] == 0
    ng = -(-3000 // 512)
    assert rec["roofline"]["collectives_issued"] == {"broadcast": 2 * ng, "all_reduce": 5, "all_gather": 1}
