#!/bin/bash
# Variant builds of the block decoder (any RFC 1951 stream: foreign members, chunk-parallel inflate) timed in both of its regimes on
# one box: many wavefronts (the foreign-member leg of the 1 GiB bench: 8 192 plain members) and few (one deflate stream of
# 64 MiB - 1 GiB cut at its sync points, profiles/sweep_chunk_finder.py: a wavefront per 128 KiB block, the device mostly idle).
# usage: profiles/abl_block_decoder.sh "<flags of variant 1>" ...      ("" = the product's flags)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_b$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  echo "[$v]"
  ZNGAMD_LIB=$SO python3 bench.py --size-mib ${MIB:-1024} --no-cpu-baseline --no-api | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline_inflate_foreign']; print('    foreign members, 1 GiB: ms', f['ms'], 'MB/s', f['decompress_MBps'])"
  ZNGAMD_LIB=$SO python3 profiles/sweep_chunk_finder.py child | cut -c1-100
  rm -f $SO
  i=$((i+1))
done
