#!/bin/bash
# round 6, fourth GPU call: the search without the wait at the top of every tile (own links taken over before the new loads go
# out, one branch-free form of the byte prefetch), the statistics inside against beside it, the programme at eight waves per CU with
# its two instantiations
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s4_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s4_cmp 600 profiles/cmp_deflate.sh "" "-DZA_STATS_FOLD=0" "" "-DZA_STATS_FOLD=0" "-DZA_DP_PAD=0"
for lv in 1 4 9; do LEVEL=$lv $G s4_level$lv 300 profiles/cmp_deflate.sh ""; done
