#!/bin/bash
# round 6, second GPU call: parity of the folded statistics and the shared workspace, where the search's time goes now (own-link
# loads against LDS reads), the programme's occupancy sweep with a pad the compiler keeps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s2_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s2_cmp 900 profiles/cmp_deflate.sh build/variants/r05.so "" "-DZA_ABL_NO_LINKLOADS" "-DZA_ABL_BC_SELF" "-DZA_ABL_NO_LINKLOADS -DZA_ABL_BC_SELF" "-DZA_ABL_SEARCH_NO_B -DZA_ABL_SEARCH_NO_C"
$G s2_occ 600 profiles/cmp_deflate.sh "-DZA_DP_PAD=2304" "-DZA_DP_PAD=4864" "-DZA_DP_PAD=8960" "-DZA_DP_PAD=14336" "-DZA_DP_PAD=22528"
