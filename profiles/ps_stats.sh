#!/bin/bash
# profiling build of the library with the sweep counters, run, then the normal build again
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
SO=python-zlib-ng_amd/zlib_ng_amd/libzng_amd.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZA_PS_STATS $PS_EXTRA -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || exit 1
python3 profiles/ps_stats.py
