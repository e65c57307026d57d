"""One-off wide fuzz of the decoders against the system zlib: random data kinds / sizes / levels / strategies / flush
patterns / containers through zngamd_gunzip, zngamd_inflate_raw and the indexed-member path."""
import gzip, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import _lib, corpus
ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
srcs = [corpus.text(6 << 20, seed=1).tobytes(), corpus.fastq(6 << 20, seed=2).tobytes(), corpus.mixed(8 << 20, seed=5).tobytes(),
        bytes(4 << 20), rng.bytes(2 << 20), (b"abcdefg" * 600000), bytes(rng.integers(0, 3, 4 << 20, dtype=np.uint8))]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for case in range(N):
    src = srcs[int(rng.integers(0, len(srcs)))]
    n = min(int(rng.choice([0, 1, 100, 70000, 300000, int(rng.integers(1, 4 << 20))])), len(src) - 1)
    o = int(rng.integers(0, len(src) - n))
    d = src[o:o + n]
    kind = int(rng.integers(0, 4))
    level = int(rng.integers(0, 10))
    if kind == 0:                                       # gzip member(s), maybe flush points
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, int(rng.integers(0, 5)))
        step = int(rng.choice([n + 1, 50000, 200000]))
        parts = []
        for i in range(0, max(n, 1), step):
            parts.append(co.compress(d[i:i + step]))
            if rng.integers(0, 3) == 0:
                parts.append(co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))))
        blob = b"".join(parts) + co.flush()
        if rng.integers(0, 2):
            blob = blob + bytes(int(rng.integers(0, 9))) + gzip.compress(d[:1000], 6)
            want = d + d[:1000]
        else:
            want = d
        code, out, nm = ctx.gunzip(blob, len(want) + 64)
        ok = code == 0 and out == want
    elif kind == 1:                                     # raw stream
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        blob = co.compress(d) + co.flush()
        code, out, used, crc, ad = ctx.inflate_raw(blob, n + 64)
        if code == _lib.BUF_ERROR and ctx.last_needed:
            code, out, used, crc, ad = ctx.inflate_raw(blob, ctx.last_needed)
        ok = code == _lib.STREAM_END and out == d and used == len(blob) and crc == zlib.crc32(d)
    elif kind == 2:                                     # indexed members
        blob = ctx.gzip_members(d, int(rng.choice([1024, 4096, 65536, 131072])), level)
        ok = gzip.decompress(blob) == d
        code, out, nm = ctx.gunzip(blob, n + 64)
        ok = ok and code == 0 and out == d
    else:                                               # our own one-shot stream through the system zlib and back
        raw, crc, ad = ctx.deflate_stream(d, level)
        ok = zlib.decompressobj(-15).decompress(raw) == d and crc == zlib.crc32(d)
        code, out, used, _, _ = ctx.inflate_raw(raw, n + 64)
        if code == _lib.BUF_ERROR and ctx.last_needed:
            code, out, used, _, _ = ctx.inflate_raw(raw, ctx.last_needed)
        ok = ok and code == _lib.STREAM_END and out == d
    if not ok:
        bad += 1
        print("MISMATCH case", case, "kind", kind, "n", n, "level", level, "src", srcs.index(src), "off", o)
print("cases", N, "mismatches", bad)
