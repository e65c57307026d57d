#!/bin/bash
# chain kernel: which wave of the four sets the tick (parts switched off one at a time; 1 GiB, kernel times only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/abl_deflate_noverify.sh "" "-DZA_ABL_CH_NOHASH" "-DZA_ABL_CH_NOATOMIC" "-DZA_ABL_CH_NOCHECK" "-DZA_ABL_CH_NOSTORE" "-DZA_ABL_CH_NOHASH -DZA_ABL_CH_NOSTORE" > gpurun_out/s19_chains.log 2>&1
