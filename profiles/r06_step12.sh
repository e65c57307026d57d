#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s12_units 600 python3 -m pytest tests/test_gpu_indexed_chain.py -x -q
$G s12_all 1150 python3 -m pytest tests -m gpu -x -q
