#!/usr/bin/env python3
"""summary.txt of profiles/run_pmc.sh -> profiles/pmc_traffic.json: HBM bytes per unit (128 KiB block / member) of every kernel.

    python3 profiles/pmc_to_json.py gpurun_out/pmc_<tag>/summary.txt <units per launch> <build tag>

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half the bytes of wide (16-byte-per-lane) reads
(MI355X_MICROARCH.md, HBM section), which is how every kernel here reads; WRITE_SIZE is exact for 16-byte stores."""
import json
import re
import sys

path, units, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
per = {}
cur = None
for line in open(path):
    if not line.startswith(" "):
        cur = line.strip().split("<")[0]
        per.setdefault(cur, {})
        continue
    m = re.match(r"\s+(\S+)\s+total\s+(\d+)\s+launches\s+(\d+)\s+per-launch\s+([\d.]+)", line)
    if m and cur:
        # the instantiations of a template (za_k_chains<0>, <1>, <2>: one launch each per step) add up under the plain name
        per[cur][m.group(1)] = per[cur].get(m.group(1), 0.0) + float(m.group(4))
out = {"_comment": "HBM bytes PER UNIT (128 KiB block / member) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB units, separate passes, "
                   "profiles/run_pmc.sh): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / units per launch; FETCH_SIZE doubled because gfx950 counts "
                   "half the bytes of wide coalesced reads.  bench.py multiplies by the units of one launch.",
       "per": "unit", "build": tag, "units_per_launch": units}
for k, c in sorted(per.items()):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and c["FETCH_SIZE"] + c["WRITE_SIZE"] > 1000:
        out[k] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / units)
        out[k + "_fetch_raw"] = int(c["FETCH_SIZE"] * 1024 / units)
        out[k + "_write"] = int(c["WRITE_SIZE"] * 1024 / units)
# vector wave-instructions per unit (SQ pass of the same build): bench.py's `dominant_kernel.valu` view
valu = {k: round(c["SQ_INSTS_VALU"] / units, 1) for k, c in sorted(per.items()) if c.get("SQ_INSTS_VALU", 0) > 1e6}
if valu:
    out["valu_wave_insts_per_unit"] = valu
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
