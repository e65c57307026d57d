#!/bin/bash
# The wide fuzz scripts over several seeds on one box (a round's last check of parity).  usage: [SEED0=400] profiles/run_fuzz_wide.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/fuzz_$1
mkdir -p $OUT
S=${SEED0:-400}
for seed in $((S+1)) $((S+2)) $((S+3)) $((S+4)); do
  timeout -k 10 280 python3 profiles/fuzz_deflate_parity.py $seed 600 > $OUT/deflate_$seed.txt 2>&1 || { echo "deflate $seed failed"; tail -3 $OUT/deflate_$seed.txt; exit 1; }
  tail -1 $OUT/deflate_$seed.txt
done
for seed in $((S+1)) $((S+2)); do
  timeout -k 10 280 python3 profiles/fuzz_inflate_parity.py $seed 600 > $OUT/inflate_$seed.txt 2>&1 || { echo "inflate $seed failed"; tail -3 $OUT/inflate_$seed.txt; exit 1; }
  tail -1 $OUT/inflate_$seed.txt
  timeout -k 10 280 python3 profiles/fuzz_chain_runs.py $seed 60 > $OUT/chain_runs_$seed.txt 2>&1 || { echo "chain runs $seed failed"; tail -3 $OUT/chain_runs_$seed.txt; exit 1; }
  tail -1 $OUT/chain_runs_$seed.txt
  timeout -k 10 280 python3 profiles/fuzz_compress_objects.py $seed 300 > $OUT/cobj_$seed.txt 2>&1 || { echo "compress objects $seed failed"; tail -3 $OUT/cobj_$seed.txt; exit 1; }
  tail -1 $OUT/cobj_$seed.txt
  timeout -k 10 280 python3 profiles/fuzz_stream_objects.py $seed 300 > $OUT/sobj_$seed.txt 2>&1 || { echo "stream objects $seed failed"; tail -3 $OUT/sobj_$seed.txt; exit 1; }
  tail -1 $OUT/sobj_$seed.txt
done
for seed in $((S+1)) $((S+2)); do
  timeout -k 10 280 python3 profiles/fuzz_indexed_chain.py $seed 120 > $OUT/indexed_$seed.txt 2>&1 || { echo "indexed chain $seed failed"; tail -3 $OUT/indexed_$seed.txt; exit 1; }
  tail -1 $OUT/indexed_$seed.txt
done
for seed in $((S+1)) $((S+2)); do
  timeout -k 10 280 python3 profiles/fuzz_small_calls.py $seed 400 > $OUT/small_$seed.txt 2>&1 || { echo "small calls $seed failed"; tail -3 $OUT/small_$seed.txt; exit 1; }
  tail -1 $OUT/small_$seed.txt
done
timeout -k 10 280 python3 profiles/fuzz_reader_windows.py $((S+1)) 30 > $OUT/reader_401.txt 2>&1 || { echo "reader windows failed"; tail -3 $OUT/reader_401.txt; exit 1; }
tail -1 $OUT/reader_401.txt
