"""One-off wide fuzz: HIP deflate == oracle deflate, byte for byte, on random (data kind, size, dictionary length, level,
FINAL flag) -- more cases than the committed parity tests, run by hand on a GPU box."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from oracle import oracle as O
from zlib_ng_amd import _lib, corpus
ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
srcs = [corpus.text(2 << 20, seed=1).tobytes(), corpus.fastq(2 << 20, seed=2).tobytes(), corpus.mixed(4 << 20, seed=5).tobytes(),
        bytes(2 << 20), rng.bytes(1 << 20), (b"abc" * 700000), bytes(rng.integers(0, 4, 2 << 20, dtype=np.uint8))]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for case in range(N):
    src = srcs[int(rng.integers(0, len(srcs)))]
    n = int(rng.choice([1, 2, 5, 6, 7, 63, 64, 65, 2047, 2048, 2049, 4095, 4097, 65536, 131071, 131072, int(rng.integers(1, 131073))]))
    dl = int(rng.choice([0, 0, 1, 5, 6, 63, 255, 4095, 32767, 32768, int(rng.integers(0, 32769))]))
    o = int(rng.integers(0, len(src) - n - dl))
    level = int(rng.integers(0, 10))
    flags = int(rng.integers(0, 2))
    buf = src[o:o + dl + n]
    outs, crcs, ovf = ctx.deflate_blocks(buf, [(dl, n, dl, flags)], level, n + n // 8 + 700)
    ref, rcrc = O.deflate_unit(buf[dl:], buf[:dl], level=level, flags=flags)
    ok = outs[0] == ref and crcs[0] == rcrc
    if ok:
        d = zlib.decompressobj(-15, zdict=buf[:dl]) if dl else zlib.decompressobj(-15)
        ok = d.decompress(outs[0]) == buf[dl:]
    if not ok:
        bad += 1
        print("MISMATCH case", case, "n", n, "dict", dl, "level", level, "flags", flags, "src", srcs.index(src), "off", o)
print("cases", N, "mismatches", bad)
