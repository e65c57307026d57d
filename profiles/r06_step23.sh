#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/abl_deflate_noverify.sh "-DZA_CH_STATS -DZA_CH_THREADS=512" "-DZA_CH_STATS -DZA_CH_THREADS=320"  "-DZA_CH_STATS -DZA_CH_THREADS=512 -DZA_ABL_CH_NOATOMIC" > gpurun_out/s23.log 2>&1
cat gpurun_out/s23.log
