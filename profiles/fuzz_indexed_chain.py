"""Fuzz of the indexed unit decoder (za_k_inflate_units_marked) and of the file format around it: random mixes of data kinds, unit
sizes, levels, header forms and dictionaries -- compress with the engine, decode with the writer's index, compare with the input
(and with the system zlib's decode of the same stream); then whole files through the threaded writer and both readers with
random window sizes.  usage: python3 profiles/fuzz_indexed_chain.py <seed> <cases>"""
import io
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, "python-zlib-ng_amd")
from zlib_ng_amd import _lib, corpus, devmem, gzip_ng, gzip_ng_threaded  # noqa: E402

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = _lib.default_context()
B = 131072


def piece(n):
    kind = int(rng.integers(0, 8))
    s = int(rng.integers(0, 1 << 30))
    if kind == 0:
        return corpus.text(n, s).tobytes()
    if kind == 1:
        return corpus.fastq(n, s).tobytes()
    if kind == 2:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 3:
        return bytes(n)
    if kind == 4:
        w = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
        return (w * (n // len(w) + 1))[:n]
    if kind == 5:
        return rng.integers(0, 4, n, dtype=np.uint8).tobytes()
    if kind == 6:
        return corpus.mixed(max(n, 64), s).tobytes()[:n]
    return np.cumsum(rng.integers(-3, 4, n)).astype(np.uint8).tobytes()


def data_of(total):
    out = b""
    while len(out) < total:
        out += piece(int(rng.integers(1, max(2, min(total, 400000)))))
    return out[:total]


bad = 0
for case in range(cases):
    total = int(rng.integers(1, 24 * B))
    block = int(rng.choice([B, B, B, 65536, 40000, 16384, 4096, 70000, 1000]))
    level = int(rng.integers(1, 10))
    flag = int(rng.choice([_lib.FLAG_SEG2K, _lib.FLAG_FLATHDR]))
    dlen = int(rng.choice([0, 0, 32768, 1000, 32767]))
    d0 = data_of(dlen) if dlen else b""
    data = data_of(total)
    buf = d0 + data
    blocks, off = [], len(d0)
    while off < len(buf):
        n = min(block, len(buf) - off)
        blocks.append((off, n, min(32768, off), flag))
        off += n
    outs, crcs, ovf = ctx.deflate_blocks(buf, blocks, level, block + block // 8 + 600)
    assert not ovf
    d_index = ctx.deflate_index(len(blocks))
    stream = b"".join(outs) + b"\x03\x00"
    ref = (zlib.decompressobj(-15, zdict=d0) if d0 else zlib.decompressobj(-15)).decompress(stream)
    d_def = devmem.from_host(ctx, stream + bytes(64))
    d_out = devmem.empty(ctx, len(data) + 64)
    d_dict = devmem.from_host(ctx, d0) if d0 else None
    r, n = ctx.inflate_units_indexed_dev(d_def.ptr, len(stream), [len(c) for c in outs], [b[1] for b in blocks], d_index.ptr, d_out.ptr, len(data),
                                         d_dict.ptr if d_dict else None, len(d0))
    back = d_out[0:n].cpu().tobytes() if r == _lib.STREAM_END else b""
    if not (ref == data and r == _lib.STREAM_END and back == data):
        bad += 1
        print("MISMATCH case", case, "total", total, "block", block, "level", level, "flag", flag, "dict", dlen, "r", r, ctx.err())
    if case % 5 == 0:                                  # a whole file through the writer and both readers
        os.environ["ZNGAMD_READ_WINDOW"] = str(int(rng.choice([1 << 16, 1 << 20, 3 << 20, 64 << 20])))
        bio = io.BytesIO()
        with gzip_ng_threaded.open(bio, "wb", compresslevel=level, threads=1, block_size=int(rng.choice([B, 1 << 20, 65536, 300000]))) as f:
            pos = 0
            while pos < len(data):
                k = int(rng.integers(1, 3 << 20))
                f.write(data[pos:pos + k])
                pos += k
        blob = bio.getvalue()
        with gzip_ng_threaded.open(io.BytesIO(blob), "rb") as f:
            a = f.read()
        with gzip_ng.open(io.BytesIO(blob), "rb") as f:
            b = f.read()
        if not (a == data and b == data and zlib.decompress(blob, 31) == data):
            bad += 1
            print("FILE MISMATCH case", case, "total", total, "level", level, "window", os.environ["ZNGAMD_READ_WINDOW"])
print("fuzz_indexed_chain seed %d: %d units through an index; cases %d mismatches %d" % (seed, ctx.L.zngamd_indexed_units(ctx.h, 0), cases, bad))
sys.exit(1 if bad else 0)
