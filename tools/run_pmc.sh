#!/bin/bash
# rocprofv3 counter passes for one kernel family (each pass its own run; no tracing options beside --kernel-trace).
# usage: tools/run_pmc.sh conv|wgrad|sn|hist
set -e
what=$1
out=gpurun_out/pmc_$what
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/sq -- python tools/pmc_target.py $what > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python tools/pmc_target.py $what > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/write -- python tools/pmc_target.py $what > $out/write.log 2>&1
find $out -name "*counter_collection.csv" | head
