#!/bin/bash
# search: four consecutive positions per thread -- stage-wise parity, then times against the old mapping at levels 1, 4, 6
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py -x -q -m gpu > gpurun_out/s33_tests.log 2>&1 || { tail -30 gpurun_out/s33_tests.log; exit 1; }
tail -2 gpurun_out/s33_tests.log
bash profiles/cmp_deflate.sh "" "-DZA_SEARCH_CONSEC=0" > gpurun_out/s33_cmp.log 2>&1
LEVEL=1 bash profiles/cmp_deflate.sh "" "-DZA_SEARCH_CONSEC=0" >> gpurun_out/s33_cmp.log 2>&1
LEVEL=4 bash profiles/cmp_deflate.sh "" "-DZA_SEARCH_CONSEC=0" >> gpurun_out/s33_cmp.log 2>&1
cat gpurun_out/s33_cmp.log
