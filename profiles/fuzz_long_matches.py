"""Wide fuzz of the dynamic programme's long-match path and of MULTI-UNIT batches: HIP deflate == oracle deflate, byte for byte,
on data made of zero runs, periodic pieces, sparse bytes and text spliced at random (matches far longer than the programme's
64-slot ring), levels 4-9, batches of 1-6 dictionary-chained units of random size.  usage: fuzz_long_matches.py <seed> <cases>"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from oracle import oracle as O
from zlib_ng_amd import _lib, corpus
ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
text = corpus.text(1 << 20, seed=7).tobytes()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = 131072


def piece(n):
    k = int(rng.integers(0, 6))
    if k == 0: return bytes(n)
    if k == 1:
        per = text[int(rng.integers(0, 1000)):][:int(rng.integers(1, 600))]
        return (per * (n // len(per) + 1))[:n]
    if k == 2: return np.where(rng.random(n) < 0.02, rng.integers(1, 256, n), 0).astype(np.uint8).tobytes()
    if k == 3:
        o = int(rng.integers(0, len(text) - n)); return text[o:o + n]
    if k == 4: return bytes([int(rng.integers(0, 256))]) * n
    o = int(rng.integers(0, len(text) - 300)); chunk = text[o:o + int(rng.integers(70, 300))]
    return (chunk * (n // len(chunk) + 1))[:n]


bad = 0
for case in range(N):
    total = int(rng.choice([int(rng.integers(1, 3 * B)), int(rng.integers(B, 6 * B)), 2 * B, 3 * B + 17]))
    parts = []
    while sum(map(len, parts)) < total:
        parts.append(piece(int(rng.integers(1, 70000))))
    data = b"".join(parts)[:total]
    level = int(rng.integers(4, 10))
    nb = (total + B - 1) // B
    blocks = [(b * B, min(B, total - b * B), 32768 if b else 0, 1 if (b == nb - 1 and rng.integers(0, 2)) else 0) for b in range(nb)]
    outs, crcs, ovf = ctx.deflate_blocks(data, blocks, level, B + B // 8 + 700)
    ok = not ovf
    for b, (off, ln, dl, fl) in enumerate(blocks):
        ref, rcrc = O.deflate_unit(data[off:off + ln], data[off - dl:off], level=level, flags=fl)
        if outs[b] != ref or crcs[b] != rcrc:
            ok = False
            print("MISMATCH case", case, "unit", b, "of", nb, "level", level, "total", total)
            break
    if ok:
        tail = b"" if blocks[-1][3] else b"\x03\x00"
        ok = zlib.decompressobj(-15).decompress(b"".join(outs) + tail) == data
        if not ok: print("ROUND TRIP case", case)
    bad += not ok
print("cases", N, "mismatches", bad)
