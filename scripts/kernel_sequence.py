"""The kernels of the LAST complete fit step in a rocprofv3 kernel trace, in start order, with durations and gaps (small n:
which launches a step is made of).  usage: kernel_sequence.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(({"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]), "grid": r["Grid_Size_X"], "wg": r["Workgroup_Size_X"]} for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["start"])
idx = [i for i, r in enumerate(rows) if "gram_sym" in r["name"]]
a, b = idx[-2], idx[-1]
prev = None
for r in rows[a:b]:
    gap = (r["start"] - prev) / 1e3 if prev else 0.0
    print(f"{(r['start']-rows[a]['start'])/1e3:8.1f} us  dur {(r['end']-r['start'])/1e3:6.1f}  gap {gap:5.1f}  grid {r['grid']:>6s}  {r['name'][:90]}")
    prev = r["end"]
print("step", (rows[b]["start"] - rows[a]["start"]) / 1e3, "us,", b - a, "kernels")
