#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 python3 profiles/fuzz_small_calls.py 1 300 > gpurun_out/s34_fuzz.log 2>&1
tail -8 gpurun_out/s34_fuzz.log
