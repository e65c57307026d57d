#!/bin/bash
# Variant builds timed on the 256 MiB BGZF file of time_bgzf.py (4 097 members of 64 KiB), one box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_b$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  echo "[$v]"
  ZNGAMD_LIB=$SO python3 profiles/time_bgzf.py | grep "gunzip:"
  rm -f $SO
  i=$((i+1))
done
