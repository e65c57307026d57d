#!/bin/bash
# chain-kernel run lengths (ZNGAMD_CHAIN_RUN; unset = sized to the device) on the bench of MIB MiB, one box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for r in "$@"; do
  if [ "$r" = "auto" ]; then unset ZNGAMD_CHAIN_RUN; else export ZNGAMD_CHAIN_RUN=$r; fi
  echo "[run $r]"
  python3 bench.py --size-mib ${MIB:-1024} --no-cpu-baseline --no-foreign --no-api 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ratio', d['ratio'], d['kernel_ms_per_step'])"
done
