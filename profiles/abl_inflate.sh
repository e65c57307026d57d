#!/bin/bash
# Ablation / variant builds of the library (each in a scratch path: the product library is never touched), timed on the member inflate.
# usage: profiles/abl_inflate.sh "<flags of variant 1>" "<flags of variant 2>" ...     ("" = the product source as it is)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_v$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  ABL="[$v]" ZNGAMD_LIB=$SO python3 profiles/time_inflate_members.py
  rm -f $SO
  i=$((i+1))
done
