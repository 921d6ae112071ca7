#!/usr/bin/env python3
"""Per-launch HBM traffic of the roofline kernel (3x3 forward/dgrad MFMA conv) from the two --pmc passes of
tools/run_pmc_bench.sh.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64
bytes, so it is doubled.  Writes gpurun_out/pmc_bench/summary.json (copy it to profiles/)."""
import csv, glob, json, sys
d = sys.argv[1]
KEY = "conv_fwd_kernel<32, 9, false, true, "     # both epilogue variants (plain, and dgrad + SwitchNorm reduction)
res = {}
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals = []
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (d, sub)):
        for r in csv.DictReader(open(f)):
            if KEY in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    res[name] = vals
n = len(res["FETCH_SIZE"])
fetch = 2.0 * 1024.0 * sum(res["FETCH_SIZE"]) / max(n, 1)
write = 1024.0 * sum(res["WRITE_SIZE"]) / max(len(res["WRITE_SIZE"]), 1)
out = {"kernel": "conv_fwd_kernel<32, 9, false, true, *> (forward + dgrad launches of 3 training steps, both epilogue variants)", "launches": n,
       "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
       "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); Infinity-Cache hits are included"}
json.dump(out, open(d + "/summary.json", "w"), indent=1)
print(json.dumps(out))
