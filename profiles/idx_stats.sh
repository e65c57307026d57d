#!/bin/bash
# measurement build of the library with per-phase timers in za_k_inflate_indexed (scratch path: the product library is not touched)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
SO=$ROOT/gpurun_out/variants/libzng_amd_idxstats.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZA_IDX_STATS $IDX_EXTRA -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || exit 1
ZNGAMD_LIB=$SO python3 profiles/idx_stats.py
