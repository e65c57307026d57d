#!/bin/bash
# Member inflate of the product build against variant builds shipped under build/variants/ (made on the build host from another
# revision or with other flags): same box, one call.  usage: profiles/cmp_inflate.sh [variant.so ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for so in "$@"; do
  ABL="[$(basename $so)]" ZNGAMD_LIB=$ROOT/$so python3 profiles/time_inflate_members.py
done
ABL="[product]" python3 profiles/time_inflate_members.py
