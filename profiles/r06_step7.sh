#!/bin/bash
# round 6, seventh GPU call: own links as one 8-byte load per table and thread, handed out through the LDS crossbar
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s7_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s7_cmp 600 profiles/cmp_deflate.sh "" ""
for lv in 1 4 9; do LEVEL=$lv $G s7_level$lv 300 profiles/cmp_deflate.sh ""; done
$G s7_abl 600 profiles/abl_deflate_noverify.sh "" "-DZA_ABL_NO_LINKLOADS"
