#!/bin/bash
# Variant builds timed on the foreign-member leg of the 1 GiB bench (8 192 plain gzip members of the system zlib), one box.
# usage: profiles/abl_foreign.sh "<flags of variant 1>" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_f$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  echo "[$v]"
  ZNGAMD_LIB=$SO python3 bench.py --size-mib ${MIB:-1024} --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline_inflate_foreign']; print('   foreign ms', f['ms'], 'MB/s', f['decompress_MBps'], ' indexed inflate ms', d['kernel_ms_per_step']['inflate'])"
  rm -f $SO
  i=$((i+1))
done
