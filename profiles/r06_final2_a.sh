#!/bin/bash
# round 6, final evidence A: the whole GPU suite and the wide fuzzers on the final build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G fa_suite 1100 python3 -m pytest tests -m gpu -x -q
SEED0=600 $G fa_fuzz 1150 bash profiles/run_fuzz_wide.sh r06k
