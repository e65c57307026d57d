"""One-off wide fuzz of the chain kernel's runs: random mixes of block sizes and dictionaries, random run lengths
(ZNGAMD_CHAIN_RUN), every unit against the oracle (which knows no runs), every level; run by hand on a GPU box.

    python profiles/fuzz_chain_runs.py [seed] [cases]
"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from oracle import oracle as O
from zlib_ng_amd import _lib, corpus
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 131072
srcs = [corpus.text(12 << 20, seed=1).tobytes(), corpus.fastq(12 << 20, seed=2).tobytes(), corpus.mixed(12 << 20, seed=5).tobytes(),
        (b"abcdefghij" * 1300000), bytes(rng.integers(0, 4, 12 << 20, dtype=np.uint8))]
bad = 0
for case in range(N):
    os.environ["ZNGAMD_CHAIN_RUN"] = str(int(rng.choice([1, 2, 3, 4, 7, 16, 1000])))
    ctx = _lib.Context(device=0)
    src = srcs[int(rng.integers(0, len(srcs)))]
    nblk = int(rng.integers(2, 24))
    sizes = [int(rng.choice([B, B, B, 2 * B, 3 * B + 5, 100000, 40000, 32768, 32767, 20000, 6, 1, 0, int(rng.integers(1, 3 * B))])) for _ in range(nblk)]
    level = int(rng.integers(1, 10))
    blocks, off = [], 0
    for i, sz in enumerate(sizes):
        dl = 0 if (i == 0 or rng.integers(0, 8) == 0) else min(32768, off)
        blocks.append((off, sz, dl, 0))
        off += sz
    data = src[:off]
    outs, crcs, ovf = ctx.deflate_blocks(data, blocks, level, max(sizes) + max(sizes) // 8 + 1000)
    ok = not ovf
    for (o, s, d, _), out, crc in zip(blocks, outs, crcs):
        ref, pos, dl, first = b"", o, d, True
        for k in range(max(1, (s + B - 1) // B)):
            lo, hi = o + k * B, min(o + (k + 1) * B, o + s)
            udl = min(32768, d + k * B)
            r, c = O.deflate_unit(data[lo:hi], data[lo - udl:lo], level=level)
            ref += r
        ok = ok and out == ref and crc == zlib.crc32(data[o:o + s])
    if not ok:
        bad += 1
        print("MISMATCH case", case, "run", os.environ["ZNGAMD_CHAIN_RUN"], "level", level, "sizes", sizes)
    del ctx
print("cases", N, "mismatches", bad)
