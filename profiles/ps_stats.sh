#!/bin/bash
# profiling build of the library with the sweep counters (kept in a scratch path: the product library is not touched), then run
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
SO=$ROOT/gpurun_out/variants/libzng_amd_psstats.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZA_PS_STATS $PS_EXTRA -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || exit 1
ZNGAMD_LIB=$SO python3 ${PS_SCRIPT:-profiles/ps_stats.py}
