#!/bin/bash
# round 6, sixth GPU call: where the search's time goes on the current build (timing only, no checks)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s6_abl 1100 profiles/abl_deflate_noverify.sh "" "-DZA_ABL_NO_LINKLOADS" "-DZA_ABL_BC_SELF" "-DZA_ABL_SEARCH_NO_B" "-DZA_ABL_SEARCH_NO_C" "-DZA_ABL_SEARCH_NO_B -DZA_ABL_SEARCH_NO_C" "-DZA_ABL_NO_EXTEND" "-DZA_ABL_SEARCH_NOLIT"
LEVEL=5 $G s6_level5 300 profiles/abl_deflate_noverify.sh ""
LEVEL=3 $G s6_level3 300 profiles/abl_deflate_noverify.sh ""
LEVEL=2 $G s6_level2 300 profiles/abl_deflate_noverify.sh ""
