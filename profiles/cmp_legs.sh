#!/bin/bash
# The decode legs of the bench (indexed members, foreign members, BGZF, the compress leg's own chained stream) of variant libraries
# built on the build host, on ONE box.  usage: [MIB=1024] profiles/cmp_legs.sh build/variants/a.so build/variants/b.so ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for so in "$@"; do
  echo "[$so]"
  ZNGAMD_LIB=$ROOT/$so python3 bench.py --size-mib ${MIB:-1024} --no-cpu-baseline --no-api 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   inflate_members ms', d['kernel_ms_per_step']['inflate'], '| foreign', d['roofline_inflate_foreign']['ms'], '| bgzf', d['roofline_inflate_bgzf']['ms'],
      '| chained wall', d['roofline_inflate_chained']['ms'], 'kernels', d['roofline_inflate_chained']['kernel_ms'], '| deflate', {k: d['kernel_ms_per_step'][k] for k in ('chains','search','parse','pack')})"
done
