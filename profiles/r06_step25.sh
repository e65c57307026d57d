#!/bin/bash
# small calls: the new checksum kernel (tests first), the plan kernel's phases
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests -x -q -m gpu -k "checksum or crc or adler or api_zlib or small" > gpurun_out/s25_tests.log 2>&1 || { tail -30 gpurun_out/s25_tests.log; exit 1; }
tail -2 gpurun_out/s25_tests.log
python3 profiles/time_small_calls.py 2>&1 | cut -c1-330 > gpurun_out/s25_small.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZA_PLAN_STATS -o gpurun_out/variants_plan.so python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null
ZNGAMD_LIB=$PWD/gpurun_out/variants_plan.so python3 profiles/time_small_calls.py 2>&1 | cut -c1-330 >> gpurun_out/s25_small.log
rm -f gpurun_out/variants_plan.so
cat gpurun_out/s25_small.log
