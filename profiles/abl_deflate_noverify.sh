#!/bin/bash
# Ablation builds whose output is wrong on purpose (parts of a kernel switched off): kernel times of the 1 GiB deflate only, no checks.
# usage: profiles/abl_deflate_noverify.sh "<flags of variant 1>" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_n$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  echo "[$v]"
  ZNGAMD_LIB=$SO python3 profiles/abl_deflate.py 2>&1 | grep ablate
  rm -f $SO
  i=$((i+1))
done
