#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 profiles/time_small_calls.py 2>&1 | cut -c1-480 > gpurun_out/s28_small.log
PS_SCRIPT=profiles/ps_stats_small.py bash profiles/ps_stats.sh >> gpurun_out/s28_small.log 2>&1
cat gpurun_out/s28_small.log
