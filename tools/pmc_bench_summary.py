#!/usr/bin/env python3
"""Per-launch HBM traffic of the roofline kernel (3x3 forward/dgrad MFMA conv) from the two --pmc passes of
tools/run_pmc_bench.sh.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64
bytes, so it is doubled.  Writes gpurun_out/pmc_bench/summary.json (copy it to profiles/)."""
import csv, glob, hashlib, json, os, sys
d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC_HASH = hashlib.sha256(open(os.path.join(ROOT, "semantic-segmentation_amd", "csrc", "conv_mfma.hip"), "rb").read()).hexdigest()
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
       "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); Infinity-Cache hits are included",
       "source_sha256": SRC_HASH}
json.dump(out, open(d + "/summary.json", "w"), indent=1)
print(json.dumps(out))

# every kernel of the step: launches and average HBM bytes per launch (same corrections), largest total traffic first
import collections, re
tab = collections.defaultdict(lambda: {"launches": 0, "fetch": 0.0, "write": 0.0})
for name, sub, scale in (("FETCH_SIZE", "fetch", 2048.0), ("WRITE_SIZE", "write", 1024.0)):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (d, sub)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*", "", re.sub(r"std::conditional<.*", "...", k))[:72]
            tab[k][sub] += scale * float(r["Counter_Value"])
            if sub == "fetch":
                tab[k]["launches"] += 1
rows = []
for k, v in tab.items():
    n_ = max(v["launches"], 1)
    rows.append({"kernel": k, "launches": v["launches"], "fetch_bytes_per_launch": v["fetch"] / n_,
                 "write_bytes_per_launch": v["write"] / n_, "total_GB": (v["fetch"] + v["write"]) / 1e9})
rows.sort(key=lambda r_: -r_["total_GB"])
json.dump({"note": "all kernels of the profiled steps; FETCH_SIZE doubled, WRITE_SIZE as counted, per launch", "kernels": rows},
          open(d + "/all_kernels.json", "w"), indent=1)
for r_ in rows[:16]:
    print("%-72s %5d  fetch %9.1f MB  write %9.1f MB" % (r_["kernel"], r_["launches"], r_["fetch_bytes_per_launch"] / 1e6,
                                                         r_["write_bytes_per_launch"] / 1e6))
