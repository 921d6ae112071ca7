#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02ocr
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -o bench -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-events --model unet3p_res_ocr > $out/kstats.log 2>&1
cp $out/kstats/bench_kernel_stats.csv $out/res_ocr_kernel_stats.csv
tail -2 $out/kstats.log | cut -c1-300
