#!/bin/bash
# round 6: own links through a staging area in the byte ring's unreachable part (ZA_OWN_STAGE) against eight 2-byte loads
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s15_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s15_cmp 600 profiles/cmp_deflate.sh "" "-DZA_OWN_STAGE=0" "" "-DZA_OWN_STAGE=0"
for lv in 1 4 9; do LEVEL=$lv $G s15_level$lv 300 profiles/cmp_deflate.sh "" "-DZA_OWN_STAGE=0"; done
