#!/usr/bin/env python3
"""A/B of an ENVIRONMENT switch of libdgp_hip.so on ONE box, in alternating child processes, by WALL time per fit step
(stage sums do not see overlap across streams).

    python scripts/env_ab.py VAR=a,b[,c] [--shapes 8192x32,4096x64,8192x1] [--steps 8] [--rounds 3] [--dtype f64]

Every (value, round) is a fresh process: plan of `sites` loadest sites of n observations (d = 3), 3 warm-up steps, `steps`
timed steps (perf_counter around the loop + synchronize)."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def child(n, S, steps, dtype):
    import torch

    sys.path.insert(0, ROOT)
    from discontinuum_amd import _lib

    if os.environ.get("DGP_LIB_PATH"):  # A/B of two BUILDS: DGP_LIB_PATH=discontinuum_amd/libdgp_hip.so,scripts/libdgp_prev.so
        _lib.LIB_PATH = os.path.abspath(os.environ["DGP_LIB_PATH"])
        import ctypes

        probe = ctypes.CDLL(_lib.LIB_PATH)  # an older build may lack entry points added since: bind what it has
        for name in [k for k in _lib.SIGNATURES if not hasattr(probe, k)]:
            del _lib.SIGNATURES[name]
    import bench

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dt = torch.float64 if dtype == "f64" else torch.float32
    plan, th, r, noise = bench.make_plan("loadest", n, 3, dt, dev, S, 1 if S > 1 else 2)
    plan.set_timing(True)
    for _ in range(3):
        out = plan.fit_step(th, r, noise)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = plan.fit_step(th, r, noise)[0]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = plan.get_timing()
    nll = float(out.reshape(S, -1)[0, _lib.OUT_NLL])
    print(f"RESULT {ms:.4f} {st[_lib.TIME_POTRF]:.3f} {st[_lib.TIME_TRTRI]:.3f} {st[_lib.TIME_LAUUM]:.3f} {st[_lib.TIME_SOLVE]:.3f} "
          f"{st[_lib.TIME_GRAD]:.3f} {st[_lib.TIME_GRAM]:.3f} {nll!r} {st[_lib.TIME_SYRK_SUM]:.3f} {int(st[_lib.TIME_SYRK_N])}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    ap = argparse.ArgumentParser()
    ap.add_argument("switch")
    ap.add_argument("--shapes", default="8192x32,4096x64,8192x1")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    var, vals = a.switch.split("=")
    vals = vals.split(",")
    for shape in a.shapes.split(","):
        n, S = (int(v) for v in shape.split("x"))
        steps = a.steps if S * n ** 3 > 1e12 else a.steps * 8
        rows = {v: [] for v in vals}
        for _ in range(a.rounds):
            for v in vals:
                env = dict(os.environ, **{var: v})
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(S), str(steps), a.dtype],
                                     env=env, capture_output=True, text=True)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
                if not line:
                    print(out.stderr[-1500:], flush=True)
                    raise SystemExit(f"child failed for {var}={v}")
                rows[v].append(line[0].split()[1:])
        for v in vals:
            ms = [float(r[0]) for r in rows[v]]
            last = rows[v][-1]
            print(f"{var}={v:>3s}  n={n} S={S} {a.dtype}: wall ms/step min {min(ms):8.3f}  all {[round(x, 3) for x in ms]}  "
                  f"stages(last) potrf {last[1]} trtri {last[2]} lauum {last[3]} solve {last[4]} grad {last[5]} gram {last[6]} bulk-sum {last[8]} ({last[9]} launches)  nll {last[7]}",
                  flush=True)


if __name__ == "__main__":
    main()
