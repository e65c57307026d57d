#!/bin/bash
# round 6, fifth GPU call: the search with its results stored a tile late, the tile barrier on LDS only, the 8-byte prefetch;
# statistics beside (default) against inside the search
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
G=profiles/gpu_step.sh
rm -f gpurun_out/.stop
$G s5_parity 900 python3 -m pytest tests/test_gpu_deflate_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_ratio_heldout.py tests/test_gpu_device_resident.py tests/test_gpu_configs.py -x -q
$G s5_cmp 600 profiles/cmp_deflate.sh "" "-DZA_STATS_FOLD=1" ""
for lv in 1 4 9; do LEVEL=$lv $G s5_level$lv 300 profiles/cmp_deflate.sh ""; done
$G s5_bench4g 600 python3 bench.py --no-api --no-heldout
