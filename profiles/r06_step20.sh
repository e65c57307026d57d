#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/abl_deflate_noverify.sh "-DZA_CH_STATS" > gpurun_out/s20_chstats.log 2>&1
