#!/bin/bash
# round 6, eighth GPU call: parity of the unconditional own-link loads, the 4 GiB bench at 32 768 and at 16 384 units per launch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s8_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s8_cmp 600 profiles/cmp_deflate.sh "" ""
$G s8_b32k 600 python3 bench.py --no-api --no-heldout --no-cpu-baseline --no-foreign
ZNGAMD_CHUNK_UNITS=16384 $G s8_b16k 600 python3 bench.py --no-api --no-heldout --no-cpu-baseline --no-foreign
ZNGAMD_CHUNK_UNITS=8192 $G s8_b8k 600 python3 bench.py --no-api --no-heldout --no-cpu-baseline --no-foreign
