#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s16_fuzzix 280 python3 profiles/fuzz_indexed_chain.py 7 60
$G s16_bench 900 python3 bench.py --no-api --no-heldout --no-cpu-baseline
