#!/bin/bash
# the general one-stream kernel with a helper wavefront: inflate-side tests, then lone streams of several sizes
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests -x -q -m gpu -k "inflate or gunzip or fuzz or api or chain or foreign or reader or stream or zlib or compliance or compat or random" > gpurun_out/s41_tests.log 2>&1 || { tail -40 gpurun_out/s41_tests.log; exit 1; }
tail -2 gpurun_out/s41_tests.log
python3 profiles/time_small_streams.py > gpurun_out/s41_streams.log 2>&1
ZNGAMD_LIB=$PWD/gpurun_out/variants/libzng_amd_prev.so python3 profiles/time_small_streams.py > gpurun_out/s41_streams_prev.log 2>&1
tail -12 gpurun_out/s41_streams.log; echo "--- before (HEAD library)"; tail -12 gpurun_out/s41_streams_prev.log
