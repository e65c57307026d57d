#!/bin/bash
# Sweep of the deflate chunk size (units per launch): tail effects of the wave-per-unit kernels.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for cu in 2048 3072 4096 6144 8192 16384 32768; do
  ZNGAMD_CHUNK_UNITS=$cu python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print($cu, j['value'], j['compress_MBps'], j['decompress_MBps'], j['kernel_ms_per_step'])"
done
