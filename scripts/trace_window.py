"""Kernels of the last fit step in a rocprofv3 kernel trace (csv), as (start us, duration us, short name) rows inside a
window of the factorisation.  usage: trace_window.py trace.csv first_diag_index count"""
import csv, sys
rows = sorted(({"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["s"])
idx = max(i for i, r in enumerate(rows) if "gram_sym" in r["name"])
last = rows[idx:]
t0 = last[0]["s"]
diags = [i for i, r in enumerate(last) if "potrf_diag" in r["name"]]
k0, cnt = int(sys.argv[2]), int(sys.argv[3])
lo, hi = last[diags[k0]]["s"], last[diags[min(k0 + cnt, len(diags) - 1)]]["e"]
short = lambda n: n.split("(")[0].replace("void dgp::", "").replace("_kernel", "")[:28]
for r in last:
    if lo - 20000 <= r["s"] <= hi:
        print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:7.1f}  {short(r['name'])}")
print("potrf diag count", len(diags), "first diag at", (last[diags[0]]["s"] - t0) / 1e3, "last diag end", (last[diags[-1]]["e"] - t0) / 1e3)
