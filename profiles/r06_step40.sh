#!/bin/bash
# the bench's own small-call numbers (text) on the current build
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 bench.py --size-mib 1024 --no-cpu-baseline --no-foreign --no-heldout 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['api']['small_calls'])); print(d['value'], d['api']['compress_MBps'], d['api']['decompress_MBps'])" > gpurun_out/s40.log 2>&1
cat gpurun_out/s40.log
