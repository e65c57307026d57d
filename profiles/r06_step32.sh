#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
ZNGAMD_TRACE=1 python3 profiles/time_small_calls.py > gpurun_out/s32_small.log 2> gpurun_out/s32_trace.log
grep -v "^zng_amd trace" gpurun_out/s32_small.log | cut -c1-480
awk '/zng_amd trace/ { for (i = 1; i <= NF; i++) if ($i == "ms" && $(i-1) + 0 > 2.0) print }' gpurun_out/s32_trace.log | head -20
