#!/bin/bash
# Per-phase INSTRUCTION split of the member decoder (za_k_inflate_members): ablation builds of the library (scratch paths; the
# product library is never touched), each run under rocprofv3 --pmc on 1 GiB of level-6 members (profiles/time_inflate_members.py).
# Prints, per variant, the kernel's time and its vector / LDS / scalar / vector-memory wave-instructions per member.
# usage: profiles/abl_inflate_insts.sh "<flags of variant 1>" "<flags of variant 2>" ...     ("" = the product source as it is)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/variants
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  SO=$ROOT/gpurun_out/variants/libzng_amd_i$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $SO $ROOT/python-zlib-ng_amd/csrc/zng_amd.hip 2>/dev/null || { echo "build failed: $v"; exit 1; }
  D=$ROOT/gpurun_out/variants/pmc_i$i
  rm -rf $D
  ABL="[$v]" ZNGAMD_LIB=$SO rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d $D -- python3 $ROOT/profiles/time_inflate_members.py 2>/dev/null | grep inflate
  python3 - "$D" "$v" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc, calls = defaultdict(float), defaultdict(int)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "za_k_inflate_members" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); calls[row["Counter_Name"]] += 1
members = 8192.0
print("   per member: " + "  ".join(f"{c.replace('SQ_INSTS_', '').lower()} {acc[c] / max(1, calls[c]) / members:9.0f}" for c in sorted(acc)))
PY
  rm -rf $SO $D
  i=$((i+1))
done
