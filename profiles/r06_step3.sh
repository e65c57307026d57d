#!/bin/bash
# round 6, third GPU call: parity again (debug_fetch after debug_keep was switched off), statistics folded into the search against
# the separate kernel on one box, the search's own-link loads against its LDS reads (no checks: wrong output), the programme at 12
# waves per CU (a ring of 32 slots, long matches ignored: timing only)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s3_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s3_cmp 600 profiles/cmp_deflate.sh "" "-DZA_STATS_FOLD=0" "" "-DZA_STATS_FOLD=0"
$G s3_abl 900 profiles/abl_deflate_noverify.sh "" "-DZA_ABL_NO_LINKLOADS" "-DZA_ABL_BC_SELF" "-DZA_ABL_NO_LINKLOADS -DZA_ABL_BC_SELF" "-DZA_ABL_DP_NOLONG" "-DZA_ABL_DP_NOLONG -DZA_DP_NEAR=32" "-DZA_ABL_DP_NOLONG -DZA_DP_NEAR=32 -DZA_DP_PAD=2048" "-DZA_ABL_DP_NOLONG -DZA_DP_NEAR=16"
