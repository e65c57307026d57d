"""Fuzz of the incremental objects and of corrupted streams against CPython's zlib: decompressobj fed in random pieces with
random output limits (block-resumable decode: every resume point, start bit and history length the block decoder can
meet), and single-bit corruptions of whole streams (both sides must accept or both refuse; accepted output must agree)."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import zlib_ng, corpus
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
srcs = [corpus.text(3 << 20, seed=1).tobytes(), corpus.fastq(3 << 20, seed=2).tobytes(), corpus.mixed(4 << 20, seed=5).tobytes(),
        bytes(1 << 20), rng.bytes(1 << 20), bytes(rng.integers(0, 4, 2 << 20, dtype=np.uint8))]
bad = 0
for case in range(N):
    src = srcs[int(rng.integers(0, len(srcs)))]
    n = int(rng.choice([0, 1, 777, 70000, int(rng.integers(1, 1 << 20))]))
    o = int(rng.integers(0, len(src) - n))
    d = src[o:o + n]
    wbits = int(rng.choice([15, -15, 31, 12, -10]))
    level, strategy, mem = int(rng.integers(0, 10)), int(rng.integers(0, 5)), int(rng.integers(1, 10))
    zdict = src[max(0, o - 20000):o] if (wbits in (15, -15) and rng.integers(0, 3) == 0 and o > 100) else None
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy, *([zdict] if zdict else []))
    z = co.compress(d) + co.flush()
    tail = bytes(rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8)) if rng.integers(0, 2) else b""
    if case % 3 != 2:
        # piecewise feeding
        do = zlib_ng.decompressobj(wbits, *([zdict] if zdict else []))
        out = bytearray()
        blob = z + tail
        pos = 0
        try:
            while pos < len(blob) or do.unconsumed_tail:
                step = int(rng.choice([1, 100, 5000, 70000, 300000]))
                limit = int(rng.choice([0, 0, 1000, 50000]))
                piece = do.unconsumed_tail + blob[pos:pos + step]
                pos += step
                out += do.decompress(piece, limit)
                if do.eof:
                    break
            while not do.eof:
                more = do.decompress(do.unconsumed_tail, 100000) if do.unconsumed_tail else do.flush()
                out += more
                if not more and not do.unconsumed_tail:
                    break
            ok = bytes(out) == d and do.eof and (do.unused_data + blob[pos:]).endswith(tail) if tail else bytes(out) == d and do.eof
        except Exception as e:                    # noqa: BLE001
            ok = repr(e)
        if ok is not True:
            bad += 1
            print("FAIL feed", case, n, wbits, level, strategy, mem, bool(zdict), ok)
    else:
        if zdict:
            continue
        zz = bytearray(z)
        if len(zz) > 12:
            p = int(rng.integers(2, len(zz) - 4))
            zz[p] ^= 1 << int(rng.integers(0, 8))
        try:
            want = zlib.decompress(bytes(zz), wbits)
        except zlib.error:
            want = None
        try:
            got = zlib_ng.decompress(bytes(zz), wbits)
        except zlib_ng.error:
            got = None
        if got != want:
            bad += 1
            print("FAIL flip", case, n, wbits, level, strategy, None if want is None else len(want), None if got is None else len(got))
print("cases", N, "mismatches", bad)
