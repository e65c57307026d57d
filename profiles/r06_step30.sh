#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
PS_SCRIPT=profiles/ps_stats_small.py bash profiles/ps_stats.sh > gpurun_out/s30_small.log 2>&1
cat gpurun_out/s30_small.log
