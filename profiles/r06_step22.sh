#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/abl_deflate_noverify.sh "-DZA_CH_STATS" "-DZA_CH_STATS -DZA_CH_THREADS=384" "-DZA_CH_STATS -DZA_CH_THREADS=512" > gpurun_out/s22.log 2>&1
cat gpurun_out/s22.log
