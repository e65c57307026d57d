#!/bin/bash
# sub-sequence length / queue size / pass limit of the block decoder: rebuild with each setting and time foreign streams.
# (How r01j_foreign_ps_tune.txt was made, when one setting served all kernels.  Since then BITS and Q are template arguments
# per kernel -- ZaParBufT<BITS, Q> in za_inflate.hip -- and only ZA_PS_MAXIT is still a macro.)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
SO=$ROOT/gpurun_out/variants/libzng_amd_pstune.so        # scratch path: the product library is not touched
export ZNGAMD_LIB=$SO
for v in "-DZA_PS_BITS=1024 -DZA_PS_Q=3072" "-DZA_PS_BITS=512 -DZA_PS_Q=2048" "-DZA_PS_BITS=512 -DZA_PS_Q=3072" "-DZA_PS_BITS=768 -DZA_PS_Q=3072" "-DZA_PS_BITS=1024 -DZA_PS_Q=3072 -DZA_PS_MAXIT=4" "-DZA_PS_BITS=1024 -DZA_PS_Q=3072 -DZA_PS_MAXIT=10"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || exit 1
  echo "== $v"
  python3 profiles/time_small_streams.py | grep "131072\|1048576" | cut -c1-120
  python3 profiles/time_serial_inflate.py | grep "threaded-writer\|gzip -6\|gzip -1" | sed 's/.*kernel ms/kernel ms/' | cut -c1-150
done
