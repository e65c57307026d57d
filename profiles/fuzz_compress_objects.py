"""Fuzz of compressobj (random pieces, every flush mode, copies, dictionaries, window sizes, levels, strategies) with
CPython's zlib as the reader: after every sync / full flush the reader must have exactly what was fed so far."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import zlib_ng, corpus
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
srcs = [corpus.text(3 << 20, seed=1).tobytes(), corpus.fastq(3 << 20, seed=2).tobytes(), corpus.mixed(4 << 20, seed=5).tobytes(),
        bytes(1 << 20), rng.bytes(1 << 20), bytes(rng.integers(0, 4, 2 << 20, dtype=np.uint8))]
FL = [zlib_ng.Z_NO_FLUSH, zlib_ng.Z_PARTIAL_FLUSH, zlib_ng.Z_SYNC_FLUSH, zlib_ng.Z_FULL_FLUSH, zlib_ng.Z_BLOCK]
bad = 0
for case in range(N):
    src = srcs[int(rng.integers(0, len(srcs)))]
    n = min(int(rng.choice([0, 1, 777, 70000, 300000, int(rng.integers(1, 2 << 20))])), len(src) - 1)
    o = int(rng.integers(0, len(src) - n))
    d = src[o:o + n]
    wbits = int(rng.choice([15, -15, 31, 9, -9, 12, 25]))
    level, strategy, mem = int(rng.integers(-1, 10)), int(rng.integers(0, 5)), int(rng.integers(1, 10))
    zdict = src[max(0, o - 9000):o] if (wbits in (15, -15, 12) and rng.integers(0, 3) == 0 and o > 100) else None
    try:
        co = zlib_ng.compressobj(level, zlib_ng.DEFLATED, wbits, mem, strategy, *([zdict] if zdict else []))
        rd = zlib.decompressobj(wbits, *([zdict] if zdict else []))
        got = bytearray()
        fed = 0
        ok = True
        branch = None                                # (copy of the compressor, bytes fed, reader copy, output so far)
        while fed < n:
            step = int(rng.choice([1, 10, 1000, 40000, 200000, 1 << 20]))
            got += rd.decompress(co.compress(d[fed:fed + step]))
            fed = min(n, fed + step)
            mode = FL[int(rng.integers(0, len(FL)))] if rng.integers(0, 3) == 0 else zlib_ng.Z_NO_FLUSH
            got += rd.decompress(co.flush(mode))
            if mode in (zlib_ng.Z_SYNC_FLUSH, zlib_ng.Z_FULL_FLUSH) and bytes(got) != d[:fed]:
                ok = "after flush %d: reader has %d of %d" % (mode, len(got), fed)
                break
            if branch is None and rng.integers(0, 6) == 0:
                branch = (co.copy(), fed, rd.copy(), bytes(got))
        if ok is True:
            got += rd.decompress(co.flush())
            got += rd.flush()
            if bytes(got) != d or not rd.eof or rd.unused_data:
                ok = "final: %d of %d, eof %s" % (len(got), n, rd.eof)
        if ok is True and branch is not None:
            c2, f2, r2, g2 = branch
            other = b"and now for something different" * 50
            tail = r2.decompress(c2.compress(other) + c2.flush()) + r2.flush()
            if g2 + tail != d[:f2] + other or not r2.eof:
                ok = "copy: %d + %d" % (len(g2), len(tail))
    except Exception as e:                            # noqa: BLE001
        ok = repr(e)
    if ok is not True:
        bad += 1
        print("FAIL", case, n, wbits, level, strategy, mem, bool(zdict), ok)
print("cases", N, "mismatches", bad)
