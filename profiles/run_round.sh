#!/bin/bash
# The round's evidence in one call: --stats table of the default bench, the full default bench line, the other BASELINE
# configs, the level-9 profile.  usage: profiles/run_round.sh <tag>     (outputs under gpurun_out/round_<tag>/)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/round_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-heldout > $OUT/bench_prof.json 2> $OUT/bench_prof.err || exit 1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace9 -- python3 $ROOT/profiles/prof_level9.py > $OUT/level9.txt 2> $OUT/level9.err || exit 1
find $OUT/trace9 -name "*kernel_stats.csv" -exec cp {} $OUT/level9_kernel_stats.csv \;
rm -rf $OUT/trace9
cd $ROOT
timeout -k 10 500 python3 bench.py > $OUT/bench_4096MiB.json 2> $OUT/bench.err || exit 1
timeout -k 10 400 python3 bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/configs.err || exit 1
tail -c 1500 $OUT/bench_4096MiB.json; cat $OUT/level9.txt
