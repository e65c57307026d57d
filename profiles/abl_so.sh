#!/bin/bash
# Kernel times of the 1 GiB deflate (no checks: for ablation builds whose output is wrong on purpose) of libraries built on the build host.
# usage: profiles/abl_so.sh build/variants/a.so build/variants/b.so ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for so in "$@"; do
  echo "[$so]"
  ZNGAMD_LIB=$ROOT/$so python3 profiles/abl_deflate.py 2>&1 | grep ablate
done
