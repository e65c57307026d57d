#!/bin/bash
# inflate of small streams: the output assembled in LDS -- all inflate-side tests + fuzz seeds, then the small streams
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests -x -q -m gpu -k "inflate or gunzip or fuzz or api or chain or foreign or reader or stream or zlib or compliance or compat or random" > gpurun_out/s39_tests.log 2>&1 || { tail -40 gpurun_out/s39_tests.log; exit 1; }
tail -2 gpurun_out/s39_tests.log
python3 profiles/time_small_calls.py 2>&1 | cut -c1-480 > gpurun_out/s39_small.log
PS_SCRIPT=profiles/ps_stats_small.py bash profiles/ps_stats.sh >> gpurun_out/s39_small.log 2>&1
cat gpurun_out/s39_small.log
