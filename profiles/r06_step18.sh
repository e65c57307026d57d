#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s18_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py -x -q
$G s18_cmp 600 profiles/cmp_deflate.sh "" "-DZA_DP_TAGS_SHIFT" "" "-DZA_DP_TAGS_SHIFT"
