#!/bin/bash
# rocprofv3 kernel trace of the bench under a ONE-rank RCCL group (every collective of the data-parallel path is
# issued on the one GPU of the box) and of the plain single-process bench, same box, for the scaling budget.
# usage: tools/rccl_timeline.sh <tag, e.g. r03>
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
export SNWS_DP_SELFTEST=1
export SNWS_TRACE_COLLECTIVES=$out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rccl -o bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $out/rccl_bench.log 2>&1
cd $R
python3 tools/rccl_timeline.py $out/rccl/bench_kernel_trace.csv 2 > $out/rccl_one_rank_timeline.txt
cp $out/rccl/bench_kernel_stats.csv $out/rccl_one_rank_kernel_stats.csv
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 > $out/bench_selftest.json 2> $out/bench_selftest.err
unset SNWS_DP_SELFTEST SNWS_TRACE_COLLECTIVES
timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 > $out/bench_plain.json 2> $out/bench_plain.err
cat $out/rccl_one_rank_timeline.txt
cut -c1-160 $out/bench_selftest.json $out/bench_plain.json
