#!/bin/bash
# HBM traffic of the roofline kernel inside the bench workload: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
# cannot share a pass), each its own run of the same command, no tracing options beside --kernel-trace.
# usage: tools/run_pmc_bench.sh   -> gpurun_out/pmc_bench/{fetch,write}/..., summary in gpurun_out/pmc_bench/summary.json
set -e
out=gpurun_out/pmc_bench
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $out/write.log 2>&1
python tools/pmc_bench_summary.py $out
