#!/bin/bash
# Variant builds of the library (each in a scratch path: the product library is never touched), timed with the 1 GiB bench on ONE box
# (kernel times differ by several percent between boxes, so variants are only compared inside one call).
# usage: profiles/abl_deflate.sh "<flags of variant 1>" "<flags of variant 2>" ...     ("" = the product source as it is)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_d$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  echo "[$v]"
  ZNGAMD_LIB=$SO python3 bench.py --size-mib ${MIB:-1024} --level ${LEVEL:-6} --no-cpu-baseline --no-foreign | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ratio', d['ratio'], d['kernel_ms_per_step'])"
  rm -f $SO
  i=$((i+1))
done
