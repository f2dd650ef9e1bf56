#!/usr/bin/env python3
"""The dominant kernel's duration THREE ways, side by side: can `roofline.frac` be recomputed from
profiles/ alone?

    python scripts/roofline_check.py OUTDIR   (after scripts/collect_profiles.sh OUTDIR ...)

reads  OUTDIR/kernel_stats.csv             rocprofv3 --kernel-trace --stats: average lauum_kernel duration (the tracer's clock)
       OUTDIR/stats_stdout.json            the SAME process's HIP events per step + its in-kernel shader clock
       OUTDIR/mfma_stdout.json, mfma_busy.json   the counter pass: its own HIP events, GRBM-derived clock, MFMA busy
       OUTDIR/bench_line.json              the un-profiled `python bench.py` line of the same lease
writes OUTDIR/lauum_three_ways.json and a table on stdout."""
import csv
import json
import os
import sys

PEAK = 78.6


def main():
    out = sys.argv[1]
    rows = list(csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))))
    lau = next(r for r in rows if "lauum_kernel<" in r["Name"] or "lauum_grad_kernel<" in r["Name"])
    st = json.load(open(os.path.join(out, "stats_stdout.json")))
    flops = st["lauum_flops"]
    ev = st["lauum_ms_per_step"]
    ev_ms = sum(ev) / len(ev)
    # the tracer's launches of the fit STEPS (the first len(ev) of the process; later ones belong to the clock probe's loop, with
    # the probe's workgroups resident beside them): from the kernel trace when it is still there, else the statistics' average
    per_launch = None
    if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
        tr = [r for r in csv.DictReader(open(sys.argv[2])) if "lauum_kernel<" in r["Kernel_Name"] or "lauum_grad_kernel<" in r["Kernel_Name"]]
        tr.sort(key=lambda r: int(r["Start_Timestamp"]))
        per_launch = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr][: len(ev)]
    stats_ms = sum(per_launch) / len(per_launch) if per_launch else float(lau["AverageNs"]) / 1e6
    res = {"flops_per_launch": flops, "peak_tflops": PEAK,
           "stats_pass": {"tracer_avg_ms": stats_ms, "tracer_ms_per_step": per_launch, "stats_csv_avg_ms_all_calls": float(lau["AverageNs"]) / 1e6,
                          "calls": int(lau["Calls"]), "hip_event_avg_ms_same_process": ev_ms,
                          "hip_event_ms_per_step": ev, "tracer_minus_events_rel": stats_ms / ev_ms - 1.0,
                          "clock_mhz_in_kernel_probe": (st.get("lauum_clock") or {}).get("mhz"),
                          "frac_from_tracer": flops / (stats_ms * 1e-3) / 1e12 / PEAK}}
    c = res["stats_pass"]["clock_mhz_in_kernel_probe"]
    if c:
        res["stats_pass"]["frac_at_clock_from_tracer"] = res["stats_pass"]["frac_from_tracer"] * 2400.0 / c
    try:
        mf = json.load(open(os.path.join(out, "mfma_busy.json")))
        k = next(k for k in mf if k.startswith("lauum"))
        ms = mf[k]["total_ms"] / mf[k]["launches"]
        res["counter_pass"] = {"tracer_avg_ms": ms, "clock_mhz_grbm": mf[k]["effective_clock_mhz"], "mfma_busy_fraction": mf[k]["mfma_busy_fraction"],
                               "frac": flops / (ms * 1e-3) / 1e12 / PEAK,
                               "frac_at_clock": flops / (ms * 1e-3) / 1e12 / (PEAK * mf[k]["effective_clock_mhz"] / 2400.0)}
    except Exception as e:  # noqa: BLE001
        res["counter_pass"] = {"error": str(e)}
    try:
        b = json.load(open(os.path.join(out, "bench_line.json")))
        r = b["roofline"]
        res["bench_line"] = {"value_fits_per_s": b["value"], "hip_event_ms": r["ms_per_step"], "frac": r["frac"], "clock_mhz": r["clock_mhz"],
                             "frac_at_clock": r["frac_at_clock"], "kernel": r["kernel"]}
        res["agreement"] = {"frac_tracer_vs_line_rel": res["stats_pass"]["frac_from_tracer"] / r["frac"] - 1.0,
                            "frac_at_clock_tracer_vs_line_rel": (res["stats_pass"].get("frac_at_clock_from_tracer", float("nan")) / r["frac_at_clock"] - 1.0)
                            if r.get("frac_at_clock") else None}
    except Exception as e:  # noqa: BLE001
        res["bench_line"] = {"error": str(e)}
    json.dump(res, open(os.path.join(out, "lauum_three_ways.json"), "w"), indent=1)
    s = res["stats_pass"]
    print(f"lauum launch = {flops:.4e} flop; peak {PEAK} TFLOP/s")
    print(f"stats pass   : tracer avg {s['tracer_avg_ms']:.3f} ms over {s['calls']} calls | HIP events of the same process {s['hip_event_avg_ms_same_process']:.3f} ms"
          f" | tracer - events {100 * s['tracer_minus_events_rel']:+.2f} % | in-kernel clock {s['clock_mhz_in_kernel_probe']} MHz | frac {s['frac_from_tracer']:.4f}"
          f" | at clock {s.get('frac_at_clock_from_tracer', float('nan')):.4f}")
    if "error" not in res["counter_pass"]:
        c = res["counter_pass"]
        print(f"counter pass : tracer avg {c['tracer_avg_ms']:.3f} ms | GRBM clock {c['clock_mhz_grbm']:.0f} MHz | MFMA busy {100 * c['mfma_busy_fraction']:.1f} % | frac {c['frac']:.4f}"
              f" | at clock {c['frac_at_clock']:.4f}")
    if "error" not in res["bench_line"]:
        b = res["bench_line"]
        print(f"bench line   : HIP events {b['hip_event_ms']:.3f} ms | in-kernel clock {b['clock_mhz']} MHz | frac {b['frac']:.4f} | at clock {b['frac_at_clock']:.4f}"
              f" | {b['value_fits_per_s']:.2f} fits/s")
        print(f"agreement    : frac tracer vs line {100 * res['agreement']['frac_tracer_vs_line_rel']:+.2f} %, at clock "
              f"{100 * (res['agreement']['frac_at_clock_tracer_vs_line_rel'] or float('nan')):+.2f} %")


if __name__ == "__main__":
    main()
