#!/bin/bash
# round 6, final evidence C: counter passes on the final build (one launch set per kernel: 32 768 units, so that every kernel's
# per-launch average is per 32 768 units)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export ZNGAMD_CHUNK_UNITS=32768 ZNGAMD_UNIT_BATCH=32768
PASSES="sq fetch write sq2" timeout -k 10 1100 bash profiles/run_pmc.sh r06k > gpurun_out/pmc_r06j.log 2>&1
tail -5 gpurun_out/pmc_r06k.log
