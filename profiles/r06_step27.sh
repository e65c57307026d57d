#!/bin/bash
# small calls: one-copy result paths (deflate and inflate), pinned uploads -- the whole GPU suite, then the latencies
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests -x -q -m gpu > gpurun_out/s27_tests.log 2>&1 || { tail -40 gpurun_out/s27_tests.log; exit 1; }
tail -2 gpurun_out/s27_tests.log
python3 profiles/time_small_calls.py 2>&1 | cut -c1-330 > gpurun_out/s27_small.log
cat gpurun_out/s27_small.log
