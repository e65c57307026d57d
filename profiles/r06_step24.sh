#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
python3 profiles/time_small_calls.py > gpurun_out/s24_small.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/s24_prof -o small -- python3 profiles/time_small_calls.py > gpurun_out/s24_prof.log 2>&1
cat gpurun_out/s24_small.log | cut -c1-400
