"""Summarise a rocprofv3 kernel trace: per-kernel stats for the LAST fit step + potrf timeline samples."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void dgp::", "").replace("void ", "")[:44]
# last fit = after the last gram_sym
idx = max(i for i, r in enumerate(rows) if "gram_sym" in r["Kernel_Name"])
last = rows[idx:]
# ... up to the step's own finish_kernel: what follows (the clock probe's loop of lauum launches in --roofline-only runs) is not the step
fin = [i for i, r in enumerate(last) if "finish_kernel" in r["Kernel_Name"] and "symv" not in r["Kernel_Name"] and "refine" not in r["Kernel_Name"] and "predict" not in r["Kernel_Name"]]
if fin:
    last = last[: fin[0] + 1]
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
print(f"last fit wall {(t1-t0)/1e3:.1f} us, {len(last)} kernels")
agg = {}
for r in last:
    a = agg.setdefault(nm(r), [0, 0.0, 1e18, 0.0])
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:46s} n={a[0]:4d} total={a[1]:9.1f} us  avg={a[1]/a[0]:8.1f}  min={a[2]:8.1f}  max={a[3]:8.1f}")
if len(sys.argv) > 2:
    pot = [r for r in last if any(k in r["Kernel_Name"] for k in ("syrk", "trsm", "potrf_diag"))]
    for r in pot[:14] + pot[len(pot)//2:len(pot)//2+8] + pot[-8:]:
        print(f"  {nm(r):28s} q={r['Queue_Id']} grid={r['Grid_Size_X']:>8s} start={(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
