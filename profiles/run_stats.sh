#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench (4 GiB); usage: profiles/run_stats.sh <tag>
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-heldout > $OUT/bench.json 2> $OUT/bench.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
tail -1 $OUT/bench.json | cut -c1-400
