#!/bin/bash
# plan kernel: header section by all lanes -- parity (all deflate tests + fuzz seeds), then the phases' clocks
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests -x -q -m gpu -k "deflate or parity or fuzz or api_zlib or members or threaded" > gpurun_out/s26_tests.log 2>&1 || { tail -30 gpurun_out/s26_tests.log; exit 1; }
tail -2 gpurun_out/s26_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZA_PLAN_STATS -o gpurun_out/variants_plan.so python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null
ZNGAMD_LIB=$PWD/gpurun_out/variants_plan.so python3 profiles/time_small_calls.py 2>&1 | cut -c1-330 > gpurun_out/s26_small.log
rm -f gpurun_out/variants_plan.so
python3 profiles/time_small_calls.py 2>&1 | cut -c1-330 >> gpurun_out/s26_small.log
cat gpurun_out/s26_small.log
