"""`python bench.py --gpus N` starts its own N ranks.

The driver's multi-GPU command wraps bench.py in torch.distributed.run itself; a user (or a driver that does not)
types the plain form.  The plain form must become a launcher BEFORE it touches the GPU: it starts the ranks as a child
`python -m torch.distributed.run ...`, hands the child's stdout through unchanged and returns its exit code.
Reference analogue of the fan-out: /root/reference/examples/nwqn-loadest-example/nwqn-loadest-example.py:156-159.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_launcher_returns_the_childs_failure_without_a_gpu():
    """No GPU here: both ranks die in torch.cuda.set_device.  The launcher must come back with a non-zero code, no JSON
    line, the tail of the child's stderr and its own one-line summary -- not hang, not print a contract line."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU (the success path is the -m gpu test below)")
    run = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--size", "256", "--steps", "1", "--warmup", "1", "--sites-per-gpu", "2"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=_clean_env(DGP_BENCH_BACKEND="gloo"))
    assert run.returncode != 0
    assert not [ln for ln in run.stdout.splitlines() if ln.startswith("{")], run.stdout
    assert "2-rank child" in run.stderr and "exited with code" in run.stderr, run.stderr[-2000:]


def test_a_foreign_world_size_is_refused():
    """Launched by something else with a different rank count: say so instead of running a wrong-sized job."""
    run = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--size", "256"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert run.returncode != 0 and "WORLD_SIZE=2" in run.stderr


@pytest.mark.gpu
def test_plain_gpus_2_starts_two_ranks_and_prints_one_line(gpu_device):
    """The plain command on the one-GPU box: two ranks share the card over gloo (RCCL refuses two ranks on one device)."""
    S = 3
    cmd = [sys.executable, BENCH, "--gpus", "2", "--size", "1024", "--steps", "2", "--warmup", "1", "--sites-per-gpu", str(S),
           "--train", "4"]
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=_clean_env(DGP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), run.stdout  # stdout is exactly rank 0's line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["fits_per_step"] == 2 * S and rec["config"]["sites_per_gpu"] == S
    assert rec["value"] > 0 and abs(rec["ms_per_step"] * rec["value"] / 1e3 - 2 * S) < 1e-6
    assert rec["cpu_baseline"] is None
    assert 0 < rec["roofline"]["frac"] < 1
    # the headline configuration's own parity record travels with the line (site 0 of the batch against a single-site plan)
    par = rec["config"]["parity"]
    assert par["ok"] is True and par["nll_rel"] <= 1e-11
    # --train K: the 2 S sites trained for K iterations across the two ranks (fit_many_distributed), next to the step metric
    tr = rec["train"]
    assert tr["sites"] == 2 * S and tr["iterations"] == 4 and tr["ranks"] == 2 and tr["ok"] is True
    assert tr["site_iterations_per_s"] > 0 and abs(tr["site_iterations_per_s"] * tr["seconds"] - 2 * S * 4) < 1e-6


@pytest.mark.gpu
def test_plain_gpus_2_config5(gpu_device):
    cmd = [sys.executable, BENCH, "--gpus", "2", "--config", "5", "--size", "2500", "--dtype", "f32", "--steps", "2", "--warmup", "1"]
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=_clean_env(DGP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, run.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["config"]["info"] == 0 and rec["config"]["backend"] == "gloo"


def test_roofline_check_puts_the_three_figures_side_by_side(tmp_path):
    """scripts/roofline_check.py on a synthetic collection: tracer launches from the kernel trace (the fit steps' first, not the
    clock probe's loop behind them), the traced process's own HIP events, the counter pass and the bench line -> the table and
    `lauum_three_ways.json` with `frac` recomputed from the profile and its agreement with the line."""
    import csv

    out = tmp_path / "prof"
    out.mkdir()
    flops = 32 * 8192.0 ** 3 / 3.0
    with open(out / "kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerow(["void dgp::lauum_kernel<double>(double const*, double*, long, int, long, int)", 6, 6 * 81e6, 81.5e6, 24.0, 80e6, 83e6, 1.0])
    trace = out / "trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        t = 0
        for k, dur in enumerate((80.0e6, 80.2e6, 79.8e6, 80.0e6, 83e6, 83e6)):  # four fit steps, then the probe's loop
            w.writerow(["void dgp::lauum_kernel<double>(double const*, double*, long, int, long, int)", t, int(t + dur)])
            t += int(dur) + 1000
        w.writerow(["void dgp::gram_sym_kernel<double>()", t, t + 10])
    json.dump({"lauum_flops": flops, "lauum_ms_per_step": [80.01, 80.19, 79.81, 80.0], "lauum_clock": {"mhz": 2350.0}}, open(out / "stats_stdout.json", "w"))
    json.dump({"lauum_kernel<double>": {"total_ms": 240.3, "launches": 3, "effective_clock_mhz": 2340.0, "mfma_busy_fraction": 0.96}},
              open(out / "mfma_busy.json", "w"))
    frac_line = flops / 80.4e-3 / 1e12 / 78.6
    json.dump({"value": 116.0, "roofline": {"ms_per_step": 80.4, "frac": frac_line, "clock_mhz": 2360.0, "frac_at_clock": frac_line * 2400 / 2360.0,
                                           "kernel": "lauum_kernel"}}, open(out / "bench_line.json", "w"))
    run = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "roofline_check.py"), str(out), str(trace)], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    res = json.load(open(out / "lauum_three_ways.json"))
    s = res["stats_pass"]
    assert s["tracer_ms_per_step"] == [80.0, 80.2, 79.8, 80.0] and abs(s["tracer_avg_ms"] - 80.0) < 1e-9  # not the 83 ms of the probe's loop
    assert abs(s["tracer_minus_events_rel"]) < 1e-3 and abs(s["frac_from_tracer"] - flops / 80.0e-3 / 1e12 / 78.6) < 1e-12
    assert abs(res["counter_pass"]["tracer_avg_ms"] - 80.1) < 1e-9 and abs(res["agreement"]["frac_tracer_vs_line_rel"] - (80.4 / 80.0 - 1)) < 1e-9
    assert "agreement" in run.stdout and "tracer - events" in run.stdout
