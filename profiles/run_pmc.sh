#!/bin/bash
# PMC collection (separate passes; --pmc is never combined with trace domains other than kernel-trace).
# usage: [PASSES="sq sq2 fetch write rdreq"] [SIZE_MIB=4096] profiles/run_pmc.sh <tag> [bench args]
# rdreq: the L2's memory-side read requests by size (TCC_EA0_RDREQ_32B / _64B / _128B): bytes = 32 a + 64 b + 128 c, the
# calibration of FETCH_SIZE for this code's own access patterns (FETCH_SIZE tallies gfx950's 128-byte requests at 64).
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
PASSES=${PASSES:-"sq sq2 fetch write rdreq"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--size-mib ${SIZE_MIB:-4096} --steps 1 --warmup 1 --no-cpu-baseline --no-api --no-heldout $@"      # the shape the driver times: 32 768 units per launch
for P in $PASSES; do
  case $P in
    sq)    C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" ;;
    sq2)   C="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" ;;
    fetch) C="FETCH_SIZE" ;;
    write) C="WRITE_SIZE" ;;
    rdreq) C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" ;;
    *) echo "unknown pass $P"; exit 2 ;;
  esac
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$P -- python3 $ROOT/bench.py $ARGS > $OUT/$P.log 2>&1
done
python3 $ROOT/profiles/summarize_pmc.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
