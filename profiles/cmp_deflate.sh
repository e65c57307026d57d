#!/bin/bash
# The 1 GiB bench (MIB) of variant libraries shipped under build/variants/ and of flag variants of the working tree, on ONE box.
# usage: profiles/cmp_deflate.sh [path/to/variant.so | "-Dflags" | ""] ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  if [ -f "$ROOT/$v" ]; then SO=$ROOT/$v; KEEP=1
  else
    SO=$ROOT/gpurun_out/variants/libzng_amd_c$i.so; KEEP=0
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  fi
  echo "[$v]"
  ZNGAMD_LIB=$SO python3 bench.py --size-mib ${MIB:-1024} --level ${LEVEL:-6} --no-cpu-baseline --no-foreign 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ratio', d['ratio'], d['kernel_ms_per_step'])"
  [ $KEEP = 0 ] && rm -f $SO
  i=$((i+1))
done
