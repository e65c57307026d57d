"""GPU: randomized parity.  Inputs are stitched from text / random / zero / periodic pieces with random
lengths; levels, dictionary lengths and FINAL flags are random; every case must give the oracle's exact
bytes from the HIP pipeline, inflate with the system zlib to the input, and decode identically on the HIP
sequential decoder.  A second test forces several workspace chunks per call."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stitch(rng, fastq, total):
    parts, n = [], 0
    while n < total:
        kind = int(rng.integers(0, 5))
        ln = int(rng.integers(1, 40000))
        if kind == 0:
            o = int(rng.integers(0, len(fastq) - ln))
            p = fastq[o:o + ln]
        elif kind == 1:
            p = rng.bytes(ln)
        elif kind == 2:
            p = bytes(ln)
        elif kind == 3:
            pat = rng.bytes(int(rng.integers(1, 40)))
            p = (pat * (ln // len(pat) + 1))[:ln]
        else:
            p = bytes(rng.integers(0, 4, ln, dtype=np.uint8) + 65)      # 2-bit entropy "DNA"
        parts.append(p)
        n += ln
    return b"".join(parts)[:total]


def test_random_blocks_match_oracle(ctx, fastq):
    from oracle import oracle as O
    rng = np.random.default_rng(2026)
    for case in range(40):
        n = int(rng.choice([0, 1, 3, 4, 5, 17, 2047, 2048, 2049, 4096, 65535, 65536, 70001, 131071, 131072]))
        dl = int(rng.choice([0, 0, 1, 100, 32767, 32768]))
        level = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, -1]))
        if level in (8, 9) and n > 70001:
            n = 70001
        flags = int(rng.integers(0, 2))
        buf = _stitch(rng, fastq, dl + n)
        zd, data = buf[:dl], buf[dl:]
        outs, crcs, ovf = ctx.deflate_blocks(buf, [(dl, n, dl, flags)], level, n + n // 8 + 600)
        exp, ecrc = O.deflate_unit(data, zd, level, flags)
        assert not ovf and outs[0] == exp and crcs[0] == ecrc == zlib.crc32(data), (case, n, dl, level, flags)
        d = zlib.decompressobj(-15, zdict=zd) if dl else zlib.decompressobj(-15)
        assert d.decompress(outs[0]) == data and d.eof == bool(flags)
        code, back, used, crc, ad = ctx.inflate_raw(outs[0], n + 16, zd)
        assert back == data and code == (1 if flags else -5) and crc == ecrc


def test_many_units_over_several_workspace_chunks(fastq):
    """ZNGAMD_CHUNK_UNITS=3: a 13-unit call runs the five kernels five times over a 3-unit workspace."""
    from oracle import oracle as O
    from zlib_ng_amd import _lib
    old = os.environ.get("ZNGAMD_CHUNK_UNITS")
    os.environ["ZNGAMD_CHUNK_UNITS"] = "3"
    try:
        c = _lib.Context(device=0)
    finally:
        if old is None:
            os.environ.pop("ZNGAMD_CHUNK_UNITS", None)
        else:
            os.environ["ZNGAMD_CHUNK_UNITS"] = old
    data = fastq[:13 * 131072 - 777]
    blocks = [(off, min(131072, len(data) - off), min(32768, off), 0) for off in range(0, len(data), 131072)]
    outs, crcs, ovf = c.deflate_blocks(data, blocks, 6, 131072 + 13107)
    assert not ovf and len(outs) == 13
    for (off, ln, dl, _), out, crc in zip(blocks, outs, crcs):
        exp, ecrc = O.deflate_unit(data[off:off + ln], data[off - dl:off], 6, 0)
        assert out == exp and crc == ecrc
    assert zlib.decompress(b"".join(outs) + b"\x03\x00", -15) == data
    stream = c.gzip_members(data, 131072, 1)
    code, back, nm = c.gunzip(stream, len(data))
    assert code == 0 and back == data and nm == 13
    c.close()


def test_windowed_reader_random_streams(monkeypatch, fastq):
    """Random concatenations of member kinds (ordinary at several levels, stored-only, sync-flushed, indexed, BGZF
    fixture, empty, NUL padding) read through _GzipReader with random window sizes equal the stdlib's result."""
    import gzip
    import io
    import zlib
    from conftest import GOLDEN
    from zlib_ng_amd import _lib, corpus, zlib_ng
    ctx = _lib.default_context()
    rng = np.random.default_rng(77)
    text = corpus.text(8 << 20, seed=3).tobytes()
    mixed = corpus.mixed(4 << 20, seed=4).tobytes()
    bgzf = open(os.path.join(GOLDEN, "test.fastq.bgzip.gz"), "rb").read()

    def piece():
        kind = int(rng.integers(0, 8))
        src = text if rng.integers(0, 2) else mixed
        n = int(rng.integers(1, 3 << 20))
        o = int(rng.integers(0, len(src) - n))
        d = src[o:o + n]
        if kind == 0:
            return gzip.compress(d, int(rng.integers(1, 10)))
        if kind == 1:
            return gzip.compress(d[:200000], 0)                       # stored blocks only
        if kind == 2:                                                  # sync-flushed every ~100 KB
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            return b"".join(co.compress(d[i:i + 100000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(d), 100000)) + co.flush()
        if kind == 3:
            return ctx.gzip_members(d, int(rng.choice([4096, 65536, 131072])), int(rng.integers(1, 10)))
        if kind == 4:
            return bgzf
        if kind == 5:
            return gzip.compress(b"", 6)
        if kind == 6:
            return gzip.compress(d[:int(rng.integers(1, 3000))], 9) + bytes(int(rng.integers(0, 40)))
        return gzip.compress(d, 1)
    for trial in range(40):
        blob = b"".join(piece() for _ in range(int(rng.integers(1, 6))))
        want = gzip.decompress(blob)
        monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(int(rng.choice([65536, 100000, 262144, 1 << 20, 3 << 20]))))
        r = zlib_ng._GzipReader(io.BytesIO(blob))
        got = bytearray()
        while True:
            part = r.read(int(rng.integers(1, 2 << 20)))
            if not part:
                break
            got += part
        assert bytes(got) == want, (trial, len(got), len(want))


def test_random_block_tables_and_chain_runs_match_oracle(monkeypatch, fastq):
    """Random tables of blocks (sizes from 0 to several units, dictionaries taken or not), random run lengths of the chain and
    search kernels (ZNGAMD_CHAIN_RUN: where tables and windows are carried from unit to unit and where they are not), random
    workspace chunking, random levels: every block must come out as the concatenation of the oracle's units, which knows
    neither runs nor chunks (the wide version of this is profiles/fuzz_chain_runs.py)."""
    from oracle import oracle as O
    from zlib_ng_amd import _lib
    rng = np.random.default_rng(3026)
    B = 131072
    for case in range(14):
        monkeypatch.setenv("ZNGAMD_CHAIN_RUN", str(int(rng.choice([1, 2, 3, 5, 9, 1000]))))
        monkeypatch.setenv("ZNGAMD_CHUNK_UNITS", str(int(rng.choice([4, 7, 32768]))))
        ctx = _lib.Context(device=0)
        nblk = int(rng.integers(2, 14))
        sizes = [int(rng.choice([B, B, B, 2 * B, 3 * B + 5, 100000, 40000, 32768, 32764, 20000, 6, 1, 0, int(rng.integers(1, 2 * B))])) for _ in range(nblk)]
        level = int(rng.integers(1, 10))
        if level >= 8:
            sizes = [min(s, 70001) for s in sizes]
        blocks, off = [], 0
        for i, sz in enumerate(sizes):
            dl = 0 if (i == 0 or rng.integers(0, 6) == 0) else min(32768, off)
            blocks.append((off, sz, dl, 0))
            off += sz
        data = _stitch(rng, fastq, off + 1)[:off]
        outs, crcs, ovf = ctx.deflate_blocks(data, blocks, level, max(sizes) + max(sizes) // 8 + 1000)
        assert not ovf
        for (o, s, d, _), out, crc in zip(blocks, outs, crcs):
            ref = b""
            for k in range(max(1, (s + B - 1) // B)):
                lo, hi = o + k * B, min(o + (k + 1) * B, o + s)
                udl = min(32768, d + k * B)
                ref += O.deflate_unit(data[lo:hi], data[lo - udl:lo], level=level)[0]
            assert out == ref and crc == zlib.crc32(data[o:o + s]), (case, level, sizes, os.environ["ZNGAMD_CHAIN_RUN"])
        del ctx
