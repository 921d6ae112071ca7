#!/usr/bin/env python3
"""Regenerates profiles/README.md from a rocprofv3 kernel_stats.csv and the bench JSON line taken on the same box.
    python tools/make_profile_readme.py <kernel_stats.csv> <bench.json> <steps in the profiled run> [pmc summary txt]"""
import csv
import json
import os
import re
import sys

stats, bench_json, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
pmc = sys.argv[4] if len(sys.argv) > 4 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
b = json.loads(open(bench_json).read().strip().splitlines()[-1])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^_ZN12_GLOBAL__N_1", "", n)
    return n[:70]


out = ["# profiles/", "",
       "Round 1, MI355X (gfx950), ROCm 7.2.  `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
       "--steps %d --warmup 2 --no-cpu-baseline --no-kernel-events` (UNet3p_SNws training steps, batch 16, 512x512, bf16; "
       "kernel stats over all %d steps incl. warm-up)." % (steps - 2, steps), "",
       "* `%s` - rocprofv3 per-kernel summary (source of the table below)." % os.path.basename(stats),
       "* `%s` - the JSON line of the default `python bench.py` run on the same kind of box: **%.1f %s**, %.1f ms/step."
       % (os.path.basename(bench_json), b["value"], b["unit"], b["ms_per_step"])]
if pmc:
    out.append("* `%s` - rocprofv3 `--pmc` passes (separate runs) for the dominant kernel on the 320->320 @512^2 layer." % os.path.basename(pmc))
r = b.get("roofline") or {}
dom = [x for x in rows if "conv_fwd_kernel<32, 9, false, true, " in x["Name"]]
if dom and r:
    calls = sum(int(x["Calls"]) for x in dom)
    avg = sum(float(x["TotalDurationNs"]) for x in dom) / calls / 1e3
    out += ["", "The dominant kernel `conv_fwd_kernel<32, 9, false, true, *>` (3x3 MFMA implicit GEMM; `..., false>` = forward and "
            "plain dgrad launches, `..., true>` = dgrad launches whose epilogue also runs the SwitchNorm backward reduction) "
            "averages %.1f us per launch over its %d launches in the rocprofv3 trace; `bench.py`'s HIP events measured %.1f us "
            "for the same launches (%.0f TFLOP/s = %.3f of the 2500 TFLOP/s dense bf16 peak)."
            % (avg, calls, r["avg_launch_ms"] * 1e3, r["achieved"], r["frac"])]
out += ["", "GPU time per step %.1f ms (sum of kernel durations / %d steps)." % (tot / steps / 1e6, steps), "",
        "| kernel | calls | ms / step | avg us | % of GPU time |", "|---|---|---|---|---|"]
for x in rows[:28]:
    out.append("| `%s` | %d | %.2f | %.1f | %.1f |" % (short(x["Name"]), int(x["Calls"]), float(x["TotalDurationNs"]) / steps / 1e6,
                                                      float(x["AverageNs"]) / 1e3, 100 * float(x["TotalDurationNs"]) / tot))
# the other files of the directory, and the per-kernel HBM traffic table when its summary is there
out += ["", "Other files:", "",
        "* `r01_pmc_bench_conv.json` - HBM bytes per launch of the dominant kernel inside the same workload (`tools/run_pmc_bench.sh`: "
        "two `rocprofv3 --pmc` runs, FETCH_SIZE and WRITE_SIZE, no tracing beside `--kernel-trace`; FETCH_SIZE doubled as gfx950 "
        "tallies 128-byte requests at 64 bytes); `bench.py` copies it into `roofline.traffic`.",
        "* `r01_pmc_bench_all_kernels.json` - the same two passes for EVERY kernel of the step (table below).",
        "* `r01_pmc_sq_conv_wgrad.txt` - SQ / GRBM counter passes (`tools/run_pmc.sh conv|wgrad`) of the 320->320 3x3 layer at 512x512: "
        "MFMA pipe busy 69 % (forward) / 61 % (wgrad) of active cycles at an effective clock of 1.86 / 1.77 GHz.",
        "* `r01_res_ocr_bs16_512_kernel_stats.csv`, `r01_bench_res_ocr_gpus1.json` - `bench.py --model unet3p_res_ocr` "
        "(UNet3p_res_ocr_SNws, batch 16, 512x512): kernel summary and JSON line.",
        "* `r01_bench_disk_gpus1.json` - `bench.py --disk`: adds full-disk tiled inference (36 tiles of one 2748x2748 disk)."]
allk = os.path.join(ROOT, "profiles", "r01_pmc_bench_all_kernels.json")
if os.path.exists(allk):
    ks = json.load(open(allk))["kernels"]
    out += ["", "HBM traffic per launch from the PMC passes (MB; algorithmic traffic of the streaming SwitchNorm passes is "
            "2 tensors forward, 3 backward - the counters show exactly that ratio, i.e. no wasted re-reads):", "",
            "| kernel | launches | fetch MB | write MB |", "|---|---|---|---|"]
    for k in ks[:18]:
        out.append("| `%s` | %d | %.1f | %.1f |" % (short(k["kernel"]), k["launches"], k["fetch_bytes_per_launch"] / 1e6,
                                                   k["write_bytes_per_launch"] / 1e6))
open(os.path.join(ROOT, "profiles", "README.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
