#!/bin/bash
# Collects the per-round profile deliverables on the GPU box: rocprofv3 kernel stats of the bench command, the two PMC
# passes for the roofline kernel's HBM traffic, and the bench JSON lines.  usage: tools/collect_profiles.sh <tag, e.g. r02>
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -o bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $out/kstats.log 2>&1
cd $R
bash tools/run_pmc_bench.sh > $out/pmc.log 2>&1
cp gpurun_out/pmc_bench/summary.json $out/pmc_bench_conv.json
cp gpurun_out/pmc_bench/all_kernels.json $out/pmc_bench_all_kernels.json
# bench.py takes roofline.traffic from profiles/<tag>_pmc_bench_conv.json when its source hash matches the tree: put the
# fresh summary there (in this run's copy of the tree) before the bench lines are taken
cp $out/pmc_bench_conv.json profiles/${tag}_pmc_bench_conv.json
timeout -k 10 200 python bench.py > $out/bench_gpus1.json 2> $out/bench_gpus1.err
timeout -k 10 200 python bench.py --no-cpu-baseline --model unet3p_res_ocr > $out/bench_res_ocr_gpus1.json 2> $out/bench_res_ocr.err
timeout -k 10 200 python bench.py --no-cpu-baseline --dtype f32 > $out/bench_f32_gpus1.json 2> $out/bench_f32.err
cp $out/kstats/bench_kernel_stats.csv $out/bench_bs16_512_kernel_stats.csv
cut -c1-200 $out/bench_gpus1.json $out/bench_res_ocr_gpus1.json $out/bench_f32_gpus1.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_ocr -o bench -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-events --model unet3p_res_ocr > $out/kstats_ocr.log 2>&1
cp $out/kstats_ocr/bench_kernel_stats.csv $out/res_ocr_bs16_512_kernel_stats.csv
