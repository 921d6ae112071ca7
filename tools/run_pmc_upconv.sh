#!/bin/bash
# SQ counter pass over the up-sampled-convolution kernels (one rocprofv3 --pmc run, no tracing beside --kernel-trace).
set -e
out=gpurun_out/pmc_upconv
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/sq -- python tools/pmc_target.py upconv > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/sq2 -- python tools/pmc_target.py upconv > $out/sq2.log 2>&1 || true
find $out -name "*counter_collection.csv" | head
