"""Wide fuzz of the one-shot calls' small paths (r06): zngamd_deflate_stream == the oracle's stream byte for byte (sizes around
the 16 KiB units a call of up to 128 KiB is cut into, every level, every window), CRC-32 / Adler-32 of the call == zlib's, system
zlib decodes it; zlib_ng.decompress of system-zlib streams through the small inflate path (one copy for result, checksum and
bytes) == the data, damaged streams get zlib's verdict class.  usage: fuzz_small_calls.py [seed] [cases]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from oracle import oracle as O
from zlib_ng_amd import _lib, corpus, zlib_ng
ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
srcs = [corpus.text(1 << 20, seed=1).tobytes(), corpus.fastq(1 << 20, seed=2).tobytes(), corpus.mixed(2 << 20, seed=5).tobytes(),
        bytes(1 << 20), rng.bytes(1 << 20), (b"abc" * 350000), bytes(rng.integers(0, 4, 1 << 20, dtype=np.uint8))]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for case in range(N):
    src = srcs[int(rng.integers(0, len(srcs)))]
    n = int(rng.choice([0, 1, 2, 15, 16, 17, 1023, 1024, 4096, 16383, 16384, 16385, 32768, 49153, 65536, 131071, 131072, 131073,
                        int(rng.integers(0, 20000)), int(rng.integers(0, 140000)), int(rng.integers(0, 300000))]))
    o = int(rng.integers(0, len(src) - n))
    level = int(rng.integers(0, 10))
    wb = int(rng.choice([15, 15, 15, 9, 12]))
    data = src[o:o + n]
    ok = True
    why = ""
    raw, crc, ad = ctx.deflate_stream(data, level, wb)
    ref = O.deflate_stream(data, level, window_bits=wb)
    if raw != ref: ok = False; why = "stream differs from the oracle's"
    if ok and (crc != zlib.crc32(data) or ad != zlib.adler32(data)): ok = False; why = "checksums"
    if ok and zlib.decompressobj(-15).decompress(raw) != data: ok = False; why = "zlib does not decode it"
    if ok:
        z = zlib_ng.compress(data, level) if wb == 15 else raw
        if wb == 15 and (zlib.decompress(z) != data or zlib_ng.decompress(z) != data): ok = False; why = "container round trip"
    if ok:
        zl = int(rng.choice([1, 6, 9]))
        zc = zlib.compress(data, zl)
        if zlib_ng.decompress(zc) != data: ok = False; why = "decompress of a zlib stream"
        gz = zlib.compressobj(zl, zlib.DEFLATED, 31); gzs = gz.compress(data) + gz.flush()
        if ok and zlib_ng.decompress(gzs, 31) != data: ok = False; why = "decompress of a gzip stream"
        if ok and len(zc) > 12:
            dmg = bytearray(zc); i = int(rng.integers(2, len(zc))); dmg[i] ^= 1 << int(rng.integers(0, 8))
            try: want = zlib.decompress(bytes(dmg)); werr = None
            except zlib.error as e: want = None; werr = e
            try: got = zlib_ng.decompress(bytes(dmg)); gerr = None
            except zlib_ng.error as e: got = None; gerr = e
            if (werr is None) != (gerr is None) or (werr is None and want != got): ok = False; why = f"damaged stream: zlib {werr!r}, engine {gerr!r}"
    if not ok:
        bad += 1
        print("MISMATCH case", case, why, "n", n, "level", level, "wbits", wb, "src", srcs.index(src), "off", o)
print("cases", N, "mismatches", bad)
