#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s13_units 600 python3 -m pytest tests/test_gpu_indexed_chain.py tests/test_gpu_deflate_parity.py tests/test_gpu_api_threaded.py tests/test_gpu_inflate_parity.py -x -q
$G s13_bench 900 python3 bench.py --no-api --no-heldout --no-cpu-baseline
