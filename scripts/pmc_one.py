"""Per-kernel totals of ONE counter from a rocprofv3 --pmc pass, with kernel durations from the same run's trace.
usage: python scripts/pmc_one.py DIR [COUNTER]"""
import collections, csv, glob, sys
d = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
dur = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    if r["Counter_Name"] != name:
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void dgp::", "").replace("void ", "")
    a = agg[k]
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += dur.get(r["Dispatch_Id"], 0)
for k, (n, v, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:14]:
    print(f"{k[:60]:60s} launches={n:4d} {name}={v:14.0f}  time={ns/1e6:9.3f} ms")
