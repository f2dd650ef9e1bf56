"""MFMA pipe utilisation and effective clock per kernel from one rocprofv3 pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -o m -- python3 bench.py --roofline-only --steps 2
    python scripts/mfma_util.py DIR [out.json]
GRBM_GUI_ACTIVE is the sum over the 8 XCDs of their active shader cycles, so per dispatch (MI355X_MICROARCH.md "DVFS
give-back"):
    effective clock   = GRBM_GUI_ACTIVE / 8 / duration                      (within 3 % of the in-kernel clock for >= 10 ms)
    MFMA busy         = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   -- of the cycles the chip RAN
    MFMA busy @2.4GHz = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x duration x 2.4 GHz)          -- of the nominal-clock cycles
The second is the utilisation, the third what rounds 1-3 reported (it folds the clock the chip held into the figure).
NOTE: a --pmc pass serialises the streams and runs at a different clock than an un-profiled run (give-back item 2): the
effective clock here documents the PROFILED run; bench.py's `roofline.clock_mhz` is the un-profiled one."""
import collections, csv, glob, json, sys
d = sys.argv[1]
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    cnt[r["Dispatch_Id"]]["name"] = r["Kernel_Name"].split("(")[0].replace("void dgp::", "").replace("void ", "")
dur = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for k, v in cnt.items():
    if k in dur and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        a = agg[v["name"]]
        a[0] += 1; a[1] += v["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += dur[k]; a[3] += v.get("GRBM_GUI_ACTIVE", 0.0)
out = {}
for name, (n, busy, ns, gui) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    nominal = busy / (1024 * ns * 2.4) if ns else 0.0
    clock_mhz = gui / 8.0 / ns * 1e3 if (ns and gui) else None
    util = busy / (1024 * gui / 8.0) if gui else None
    out[name] = {"launches": n, "total_ms": ns / 1e6, "mfma_busy_fraction": util, "mfma_busy_fraction_at_2400mhz": nominal,
                 "effective_clock_mhz": clock_mhz, "grbm_gui_active": gui, "sq_valu_mfma_busy_cycles": busy}
    if ns > 2e5:
        print(f"{name[:52]:52s} launches={n:4d} total={ns / 1e6:9.2f} ms  clock {clock_mhz or 0:7.1f} MHz  MFMA busy "
              f"{100 * (util or 0):5.1f} % of the cycles run, {100 * nominal:5.1f} % of 2.4 GHz")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
