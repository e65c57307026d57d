#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in "" "-DZA_ABL_NO_CRC" "-DZA_ABL_NO_B" "-DZA_ABL_NO_B -DZA_ABL_NO_CRC"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o python-zlib-ng_amd/zlib_ng_amd/libzng_amd.so python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null
  ABL="$v" python3 profiles/abl_inflate.py
done
