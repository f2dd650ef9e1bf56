"""Kernel sequence of the LAST factorisation in a rocprofv3 kernel trace, between two diagonal-block launches:
python3 scripts/group_trace.py <kernel_trace.csv> [first_diag=8] [ndiag=5]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 5
idx = max(i for i, r in enumerate(rows) if "gram_sym" in r["Kernel_Name"])
last = rows[idx:]
t0 = int(last[0]["Start_Timestamp"])
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void dgp::", "").replace("void ", "")[:40]  # noqa: E731
diags = [i for i, r in enumerate(last) if "potrf_diag" in r["Kernel_Name"]]
lo, hi = diags[first], diags[min(first + nd, len(diags) - 1)]
for r in last[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} +{(e - s) / 1e3:8.1f} us  q={r['Queue_Id']:>3s} grid={int(r['Grid_Size_X']) // 256:>6d} x {r.get('Grid_Size_Z', '1'):>3s}  {nm(r)}")
pot = [r for r in last if any(k in r["Kernel_Name"] for k in ("syrk", "trsm", "potrf_diag", "trtri_level"))]
fin = max(int(r["End_Timestamp"]) for r in last if "potrf_diag" in r["Kernel_Name"])
print(f"last diag ends at {(fin - t0) / 1e3:.1f} us")
