#!/usr/bin/env python3
"""Regenerates tests/golden/inflate_vectors.json with the system zlib / gzip (any CPython has them).
The vectors are DATA for the inflate side: compressed streams in hex plus the SHA-256 of what they must
decode to.  (The reference's own arithmetic library, zlib-ng, is an un-vendored submodule that is absent
from /root/reference and cannot be run here; its decompressed output is fixed by the format, so a
conformant third-party encoder is a valid source of inflate vectors.  The reference's data files
tests/data/*.gz are committed next to this file unchanged.)"""
import gzip
import hashlib
import io
import json
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(2024)
    text = gzip.open(os.path.join(HERE, "test.fastq.gz")).read()[:6000]
    words = (b"the quick brown fox jumps over the lazy dog " * 40)[:1500]
    rnd = rng.bytes(700)
    zeros = bytes(3000)
    samples = {"text": text[:2500], "words": words, "rand": rnd, "zeros": zeros, "empty": b"", "one": b"Z",
               "run258": b"ab" + b"c" * 600, "far": rng.bytes(300) + bytes(32500) + rng.bytes(300)[:0] + b"END"}
    far_src = rng.bytes(400)
    samples["far"] = far_src + bytes(32768 - 400) + far_src       # distance exactly 32768
    vec = []

    def add(name, kind, blob, out, **kw):
        vec.append(dict(name=name, kind=kind, hex=blob.hex(), size=len(out), sha256=hashlib.sha256(out).hexdigest(), **kw))

    for sname, data in samples.items():
        for level, strat, tag in ((6, 0, "dyn"), (1, 0, "fast"), (9, 0, "best"), (0, 0, "stored"), (6, zlib.Z_FIXED, "fixed"),
                                  (6, zlib.Z_HUFFMAN_ONLY, "huff"), (6, zlib.Z_RLE, "rle")):
            if len(data) > 4000 and tag not in ("dyn", "stored"):
                continue
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            add(f"raw-{sname}-{tag}", "raw", co.compress(data) + co.flush(), data)
    # flush points
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    blob = co.compress(text[:1000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(text[1000:2000]) + \
        co.flush(zlib.Z_FULL_FLUSH) + co.compress(text[2000:3000]) + co.flush(zlib.Z_PARTIAL_FLUSH) + \
        co.compress(text[3000:3500]) + co.flush()
    add("raw-flushpoints", "raw", blob, text[:3500])
    # preset dictionary
    zd = text[3000:5000]
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0, zd)
    add("raw-zdict", "raw", co.compress(text[4000:6000]) + co.flush(), text[4000:6000], zdict=zd.hex())
    # small windows
    for wb in (9, 12):
        co = zlib.compressobj(6, zlib.DEFLATED, -wb)
        add(f"raw-wbits{wb}", "raw", co.compress(text) + co.flush(), text)
    # zlib container
    for level in (1, 6, 9):
        add(f"zlib-l{level}", "zlib", zlib.compress(text[:3000], level), text[:3000])
    # gzip layouts
    g1 = gzip.compress(text[:2000], 6, mtime=0)
    g2 = gzip.compress(words, 9, mtime=0)
    add("gzip-single", "gzip", g1, text[:2000])
    add("gzip-two-members", "gzip", g1 + g2, text[:2000] + words)
    add("gzip-nul-padded", "gzip", g1 + bytes(64) + g2 + bytes(7), text[:2000] + words)
    add("gzip-empty-member", "gzip", gzip.compress(b"", mtime=0) + g1, text[:2000])
    body = g1[10:]
    flags_blob = b"\x1f\x8b\x08" + bytes([2 | 4 | 8 | 16]) + bytes(4) + b"\x00\xff" + struct.pack("<H", 5) + b"extra" + \
        b"name.txt\0" + b"a comment\0"
    flags_blob += struct.pack("<H", zlib.crc32(flags_blob) & 0xFFFF) + body
    add("gzip-all-header-fields", "gzip", flags_blob, text[:2000])
    bad = bytearray(flags_blob)
    bad[12] ^= 1
    add("gzip-bad-header-crc", "gzip", bytes(bad), b"", code=-103)
    add("gzip-bad-crc", "gzip", g1[:-8] + bytes([g1[-8] ^ 0x55]) + g1[-7:], b"", code=-104)
    add("gzip-bad-isize", "gzip", g1[:-1] + bytes([g1[-1] ^ 1]), b"", code=-105)
    add("gzip-truncated", "gzip", g1[:-5], b"", code=-106)
    add("gzip-bad-magic-second", "gzip", g1 + b"\x1f\x8c" + g2[2:], b"", code=-101)
    json.dump(vec, open(os.path.join(HERE, "inflate_vectors.json"), "w"), indent=0)
    print(len(vec), "vectors,", os.path.getsize(os.path.join(HERE, "inflate_vectors.json")), "bytes")


if __name__ == "__main__":
    main()
