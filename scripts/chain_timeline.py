"""Per-panel timeline of the last potrf in a rocprofv3 kernel trace (csv or rocpd .db): diag / trsm / column-update
durations per chain step and the bulk launches that overlap them.  usage: chain_timeline.py <trace.csv|results.db>"""
import csv, sqlite3, sys

def load(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        rows = []
        for r in c.execute("select * from kernels"):
            d = dict(zip(cols, r))
            rows.append({"name": d["name"], "start": int(d["start"]), "end": int(d["end"]), "grid": d.get("grid_x", d.get("grid_size_x", 0))})
        return rows
    return [{"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]), "grid": r["Grid_Size_X"]}
            for r in csv.DictReader(open(path))]

rows = sorted(load(sys.argv[1]), key=lambda r: r["start"])
idx = max(i for i, r in enumerate(rows) if "gram_sym" in r["name"])
last = rows[idx:]
t0 = last[0]["start"]
pot = [r for r in last if any(k in r["name"] for k in ("syrk", "trsm", "potrf_diag"))]
diags = [r for r in pot if "potrf_diag" in r["name"]]
bulk = [r for r in pot if "syrk_kernel" in r["name"]]
print(f"potrf wall {(max(r['end'] for r in pot) - diags[0]['start']) / 1e3:.1f} us; bulk total {sum(r['end'] - r['start'] for r in bulk) / 1e3:.1f} us")
print(" k  start    step    diag    trsm     col    wait   bulk-overlap")
for k, d in enumerate(diags[:-1]):
    s, e, nxt = d["start"], d["end"], diags[k + 1]["start"]
    inter = [r for r in pot if e <= r["start"] < nxt and "syrk_kernel" not in r["name"]]
    td = sum(r["end"] - r["start"] for r in inter if "trsm" in r["name"]) / 1e3
    cd = sum(r["end"] - r["start"] for r in inter if "syrk_col" in r["name"]) / 1e3
    ov = sum(max(0, min(b["end"], nxt) - max(b["start"], s)) for b in bulk) / 1e3
    step, dd = (nxt - s) / 1e3, (e - s) / 1e3
    print(f"{k:2d} {(s - t0) / 1e3:7.1f} {step:7.1f} {dd:7.1f} {td:7.1f} {cd:7.1f} {step - dd - td - cd:7.1f} {ov:7.1f}")
print("bulk launches (start, dur us):", [(round((b["start"] - t0) / 1e3), round((b["end"] - b["start"]) / 1e3)) for b in bulk])
