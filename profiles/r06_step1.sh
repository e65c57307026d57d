#!/bin/bash
# round 6, first GPU call: parity of the new B / C rule, r05 against the working tree on one box, the cost of every candidate
# (levels 1-6 = A1+B, A2+B, A3+B greedy; A2+B, A2+B+C, A3+B+C with the programme; and ablation builds), the programme's occupancy sweep
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s1_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py -x -q
$G s1_cmp 600 profiles/cmp_deflate.sh build/variants/r05.so "" "-DZA_ABL_SEARCH_NO_B" "-DZA_ABL_SEARCH_NO_C" "-DZA_ABL_NO_EXTEND"
for lv in 1 2 3 4 5; do LEVEL=$lv $G s1_level$lv 300 profiles/cmp_deflate.sh build/variants/r05.so ""; done
$G s1_occ 600 profiles/cmp_deflate.sh "-DZA_DP_PAD=2304" "-DZA_DP_PAD=4864" "-DZA_DP_PAD=8960" "-DZA_DP_PAD=14336"
