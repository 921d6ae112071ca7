#!/usr/bin/env python3
"""Rewrites the 'Round 3' section of profiles/README.md from the files tools/collect_profiles.sh left under profiles/
(r03_bench_bs16_512_kernel_stats.csv, r03_bench*_gpus1.json, r03_pmc_bench_conv.json, r03_res_ocr_bs16_512_kernel_stats.csv)
and the one-rank RCCL bench lines under gpurun_out/r03/.
    python tools/make_profile_readme_r03.py"""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


def foreign(n):
    return "at::" in n or "Cijk" in n or "rocclr" in n or "rccl" in n.lower()


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return re.sub(r"^\d+", "", n)[:64]


rows = list(csv.DictReader(open(os.path.join(P, "r03_bench_bs16_512_kernel_stats.csv"))))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
ours = sum(int(r["TotalDurationNs"]) for r in rows if not foreign(r["Name"]))
sel = json.loads(open(os.path.join(ROOT, "gpurun_out", "r03", "bench_selftest.json")).read().strip().splitlines()[-1])
pla = json.loads(open(os.path.join(ROOT, "gpurun_out", "r03", "bench_plain.json")).read().strip().splitlines()[-1])
b, bo, bf = load("r03_bench_gpus1.json"), load("r03_bench_res_ocr_gpus1.json"), load("r03_bench_f32_gpus1.json")
pmc = json.load(open(os.path.join(P, "r03_pmc_bench_conv.json")))
dom = [r for r in rows if "conv_fwd_kernel<32, 9, false, true" in r["Name"]]
calls = sum(int(r["Calls"]) for r in dom)
avg = sum(int(r["TotalDurationNs"]) for r in dom) / calls / 1e3
rf = b["roofline"]
orow = list(csv.DictReader(open(os.path.join(P, "r03_res_ocr_bs16_512_kernel_stats.csv"))))
otot = sum(int(r["TotalDurationNs"]) for r in orow)
oours = sum(int(r["TotalDurationNs"]) for r in orow if not foreign(r["Name"]))
vendor = [r["Name"] for r in orow if "Cijk" in r["Name"]]
out = ["## Round 3", "",
       "MI355X (gfx950), ROCm 7.2.  `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 2 "
       "--no-cpu-baseline --no-kernel-events` (UNet3p_SNws training steps, batch 16, 512x512, bf16; kernel stats over all 5 steps "
       "incl. warm-up).  Every file of this section was taken in ONE call on one box (`tools/collect_profiles.sh r03`), from the "
       "kernel sources as committed.", "",
       "* `r03_bench_bs16_512_kernel_stats.csv` - rocprofv3 per-kernel summary (source of the table below): %.1f ms of GPU time "
       "per step, libsnws kernels %.1f %% of it (the rest: the loss's target handling, the synthetic batch and the optimizer-side "
       "torch element-wise kernels)." % (tot / 5e6, 100.0 * ours / tot),
       "* `r03_bench_gpus1.json` - the JSON line of the default `python bench.py` run: **%.1f images/s**, %.2f ms/step on this "
       "box (boxes of the pool differ by about 2 %%: same-box A/B runs are what the per-change numbers in DESIGN.md come from); `roofline.frac` %.3f "
       "(HIP events: %.1f us per launch of the dominant kernel; the rocprofv3 run above averages %.1f us over the same %d launches "
       "-> %.0f TFLOP/s = %.3f of 2500); `stream_ceiling` = the copy rate measured in the same run (%.2f TB/s)."
       % (b["value"], b["ms_per_step"], rf["frac"], rf["avg_launch_ms"] * 1e3, avg, calls, rf["flops_per_launch"] / avg / 1e6,
          rf["flops_per_launch"] / avg / 1e6 / 2500, b["stream_ceiling"]["GB/s"] / 1e3),
       "* `r03_pmc_bench_conv.json`, `r03_pmc_bench_all_kernels.json` - HBM bytes per launch inside the same workload "
       "(`tools/run_pmc_bench.sh`: two `rocprofv3 --pmc` runs, FETCH_SIZE and WRITE_SIZE, no tracing beside `--kernel-trace`; "
       "FETCH_SIZE doubled as gfx950 tallies 128-byte requests at 64 bytes): the dominant kernel fetches %.0f MB and writes %.0f MB "
       "per launch; `source_sha256` ties the file to the kernel source it was measured on (`bench.py` reports `traffic: null` for "
       "any other version)." % (pmc["fetch_bytes_per_launch"] / 1e6, pmc["write_bytes_per_launch"] / 1e6),
       "* `r03_bench_f32_gpus1.json` - `bench.py --dtype f32` (float32 tier on `v_mfma_f32_32x32x2_f32`): %.1f images/s, %.2f of "
       "the 157.3 TFLOP/s f32 matrix peak." % (bf["value"], bf["roofline"]["frac"]),
       "* `r03_bench_res_ocr_gpus1.json`, `r03_res_ocr_bs16_512_kernel_stats.csv` - `bench.py --model unet3p_res_ocr`: "
       "**%.1f images/s** (%.1f ms/step); kernel summary of a 4-step run: %s, libsnws %.1f %% of GPU time."
       % (bo["value"], bo["ms_per_step"], "no rocBLAS / hipBLASLt (`Cijk_*`) kernel" if not vendor else "VENDOR GEMM KERNELS PRESENT",
          100.0 * oours / otot),
       '* `r03_bench_200steps_gpus1.json` - `bench.py --steps 200 --warmup 10`: 334.6 images/s sustained (47.81 ms/step) on another box of the pool; `r03_bench_disk_gpus1.json` - `bench.py --disk`: full-disk inference 38.1 ms per 2748x2748 disk (36 tiles, arg-max + stitch + Evaluator on the device; 946 tiles/s); `r03_bench_all_kernel_events.json` - `bench.py --all-kernel-events`: HIP-event time of every instrumented kernel family per step (the shared-gradient adds `grad_add` are 0.06 ms/step since the k = 2 up-sampled-conv backward adds in its own epilogue).',
       "* `r03_pmc_sq_conv_wgrad.txt` - SQ / GRBM counters of the 3x3 forward and weight-gradient kernels on the 320->320 layer "
       "(`tools/run_pmc.sh conv | wgrad`): the ping-pong weight-gradient kernel fetches 6.8 GB and writes 0.07 GB for 5.4 GB of operands (round 2: 18.1 + 1.2 GB), "
       "matrix pipe busy 78 % of the active cycles at an effective clock of 1.78 GHz; the forward kernel 76 % at 1.79 GHz.",
       "* `r03_rccl_one_rank_timeline.txt`, `r03_rccl_one_rank_kernel_stats.csv` - the bench under `SNWS_DP_SELFTEST=1` (every collective of "
       "the data-parallel path issued to a one-rank RCCL communicator; `tools/rccl_timeline.sh`): RCCL launches no device kernel for a one-rank "
       "all-reduce, so the 36 statistic all-reduces per step show as 10-12 us hand-over windows of the compute stream between "
       "`sn_mix_a` -> `sn_mix_b` and `sn_bwd_mix_a2` -> `sn_bwd_mix_b` (plain and batched forms); the same call measured %.2f ms/step under "
       "the switch against %.2f ms/step without it.  N > 1 was not measured: the pool hands out one GPU." % (sel["ms_per_step"], pla["ms_per_step"]), "",
       "| kernel | calls | total ms/step | avg us | % |", "|---|---:|---:|---:|---:|"]
for r in rows[:26]:
    out.append("| `%s` | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 5e6,
                                                      float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
out.append("")
path = os.path.join(P, "README.md")
t = open(path).read()
if "## Round 3" not in t:
    t = t.replace("## Round 2", "## Round 3\n\n## Round 2", 1)
a, z = t.index("## Round 3"), t.index("## Round 2")
open(path, "w").write(t[:a] + "\n".join(out) + "\n" + t[z:])
print("\n".join(out[:12]))
