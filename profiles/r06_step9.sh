#!/bin/bash
# round 6, ninth GPU call: the indexed unit decoder (za_k_inflate_units_marked) -- parity
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s9_units 600 python3 -m pytest tests/test_gpu_indexed_chain.py -x -q
$G s9_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_inflate_parity.py tests/test_gpu_fuzz_seeds.py -x -q
