#!/bin/bash
# chain kernel with five waves per stream: parity of the link tables, the forced lane-by-lane path, times
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 -m pytest tests/test_gpu_deflate_parity.py -x -q -m gpu > gpurun_out/s21_parity.log 2>&1 || { tail -20 gpurun_out/s21_parity.log; exit 1; }
tail -2 gpurun_out/s21_parity.log
bash profiles/cmp_deflate.sh "" "-DZA_CH_FORCE_FIX" > gpurun_out/s21_cmp.log 2>&1
bash profiles/abl_deflate_noverify.sh "-DZA_CH_STATS" >> gpurun_out/s21_cmp.log 2>&1
cat gpurun_out/s21_cmp.log
