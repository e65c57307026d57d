#!/bin/bash
# PMC collection (separate passes; --pmc is never combined with trace domains other than kernel-trace).
# usage: profiles/run_pmc.sh <tag> [bench args]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--size-mib 1024 --steps 1 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -- python3 $ROOT/bench.py $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
python3 $ROOT/profiles/summarize_pmc.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
