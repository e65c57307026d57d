#!/bin/bash
# deflate + index in one engine call; two writers on one context
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_gpu_indexed_chain.py tests/test_gpu_api_threaded.py tests/test_gpu_fuzz_seeds.py -x -q -m gpu > gpurun_out/s36_tests.log 2>&1 || { tail -40 gpurun_out/s36_tests.log; exit 1; }
tail -2 gpurun_out/s36_tests.log
