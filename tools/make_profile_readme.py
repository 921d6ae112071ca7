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
open(os.path.join(ROOT, "profiles", "README.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
