#!/bin/bash
# chain kernel: would fewer LDS instructions in the inserting wave shorten the tick? (timing only: outputs are wrong)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/abl_deflate_noverify.sh "" "-DZA_ABL_CH_HALFWRITES" "-DZA_ABL_CH_HALFREADS" "-DZA_ABL_CH_HALFWRITES -DZA_ABL_CH_HALFREADS" "-DZA_ABL_CH_HALFWRITES -DZA_ABL_CH_HALFREADS -DZA_ABL_CH_NOCHECK" > gpurun_out/s38.log 2>&1
cat gpurun_out/s38.log
