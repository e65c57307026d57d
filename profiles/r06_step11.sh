#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s11_units 600 python3 -m pytest tests/test_gpu_indexed_chain.py -x -q
$G s11_api 1100 python3 -m pytest tests/test_gpu_api_threaded.py tests/test_gpu_gzip_compliance.py tests/test_gpu_api_zlib.py tests/test_gpu_inflate_parity.py tests/test_gpu_random_property.py -x -q
