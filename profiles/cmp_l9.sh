#!/bin/bash
# Level 9 (and 6) on the Silesia-like mix, per data class, for variant libraries built on the build host, on ONE box.
# usage: profiles/cmp_l9.sh build/variants/a.so "" ...     ("" = the working tree's library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for so in "$@"; do
  echo "[$so]"
  if [ -n "$so" ]; then ZNGAMD_LIB=$ROOT/$so python3 profiles/time_l9_classes.py 2>/dev/null; else python3 profiles/time_l9_classes.py 2>/dev/null; fi
done
