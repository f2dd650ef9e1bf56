"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).

Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of coalesced streaming reads -- calibrated here on kernels with a
known byte count (trmv_n reads N(N+128)/2*8 B of the lower triangle, gram_grad reads N(N+64)/2*8 B):
both report 0.50x -- so fetch bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is exact (gram_sym writes
N(N+64)/2*8 B = 270.7 MB; counter 270.5 MB).
usage: python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.json> [steps_in_run [sites_per_launch [commit]]]
The output records bench.source_hash() of the tree it is run in: bench.py emits `traffic` only while that still matches.
("fit" in the output = one step of the batched plan, i.e. one launch sequence carrying sites_per_launch sites)
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

def per_kernel(d, name):
    rows = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void dgp::", "").replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
fits = int(sys.argv[4]) if len(sys.argv) > 4 else 3
out = {"_note": "bytes; fetch = 2 * FETCH_SIZE KiB (gfx950 half-count correction, calibrated), write = WRITE_SIZE KiB",
       "fits_in_run": fits, "sites_per_launch": int(sys.argv[5]) if len(sys.argv) > 5 else None,
       "commit": sys.argv[6] if len(sys.argv) > 6 else None, "kernels": {}}
try:
    import bench
    out["source_hash"] = bench.source_hash()
except Exception as e:  # noqa: BLE001
    out["source_hash"] = None
for k in sorted(set(fetch) | set(write)):
    nf, vf = fetch.get(k, [0, 0.0]); nw, vw = write.get(k, [0, 0.0])
    n = max(nf, nw)
    fb, wb = 2 * vf * 1024, vw * 1024
    out["kernels"][k] = {"launches": n, "launches_per_fit": n / fits, "fetch_bytes_per_launch": fb / max(n, 1),
                         "write_bytes_per_launch": wb / max(n, 1), "hbm_bytes_per_launch": (fb + wb) / max(n, 1),
                         "hbm_bytes_per_fit": (fb + wb) / fits}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_fit"])[:10]:
    print(f"{k[:44]:44s} launches/fit={v['launches_per_fit']:6.1f}  HBM/launch={v['hbm_bytes_per_launch']/1e6:9.1f} MB  HBM/fit={v['hbm_bytes_per_fit']/1e9:6.2f} GB")
