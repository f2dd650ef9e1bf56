"""MFMA pipe utilisation per kernel from one rocprofv3 pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -o m -- python3 bench.py --roofline-only --steps 2
    python scripts/mfma_util.py DIR [out.json]
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel duration x 2.4 GHz); calibrated on the stand-alone GEMM
core (scripts/gemm_pmc.hip: 89.2 % busy at 70.2 of 78.6 TFLOP/s = 89.3 %)."""
import collections, csv, glob, json, sys
d = sys.argv[1]
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    cnt[r["Dispatch_Id"]]["name"] = r["Kernel_Name"].split("(")[0].replace("void dgp::", "").replace("void ", "")
dur = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k, v in cnt.items():
    if k in dur and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        a = agg[v["name"]]
        a[0] += 1; a[1] += v["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += dur[k]
out = {}
for name, (n, busy, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    util = busy / (1024 * ns * 2.4) if ns else 0.0
    out[name] = {"launches": n, "total_ms": ns / 1e6, "mfma_busy_fraction": util}
    if ns > 2e5:
        print(f"{name[:52]:52s} launches={n:4d} total={ns / 1e6:9.2f} ms  MFMA busy {100 * util:5.1f} %")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
