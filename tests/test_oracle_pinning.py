"""CPU: pin the oracle (plain-C restatement) against the reference's own known answers and data files, and
against the system zlib in both directions.  No GPU involved."""
import gzip
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O


def test_checksum_known_answers():
    # reference tests/test_zlib_compliance.py:94-120
    assert O.crc32(b"penguin", 0) == 0x0e5c1a120 & 0xFFFFFFFF
    assert O.crc32(b"penguin", 1) == 0x43b6aa94
    assert O.adler32(b"penguin", 0) == 0x0bcf02f6
    assert O.adler32(b"penguin", 1) == 0x0bd602f7
    assert O.crc32(b"abcdefghijklmnop") == 2486878355
    assert O.crc32(b"spam") == 1138425661
    assert O.adler32(b"abcdefghijklmnop" * 2) == 3573550353
    assert O.adler32(b"spam") == 72286642
    a, b = b"abcdefghijklmnop", b"spam and eggs"
    assert O.crc32_combine(O.crc32(a), O.crc32(b), len(b)) == O.crc32(a + b)
    assert O.crc32_combine(O.crc32(a), O.crc32(b""), 0) == O.crc32(a)


def test_checksums_vs_zlib_over_seeds(fastq):
    seeds = [int(s) & 0xFFFFFFFF for s in open(os.path.join(GOLDEN, "seeds.txt")).read().split()]
    for i in range(3, 20):
        d = fastq[:2 ** i]
        for s in seeds[:8] + [0, 1, 0xFFFFFFFF]:
            assert O.crc32(d, s) == zlib.crc32(d, s)
            assert O.adler32(d, s) == zlib.adler32(d, s)


def test_reference_data_files():
    # reference tests/test_gzip_ng.py:420-425, :466-473 and tests/data/README
    want = {"test.fastq.gz": (1, 3578369, 0x473f3477), "concatenated.fastq.gz": (2, 178528, 0x9c59a70c),
            "test.fastq.bgzip.gz": (56, 3578369, 0x473f3477)}
    for name, (nm, size, crc) in want.items():
        raw = open(os.path.join(GOLDEN, name), "rb").read()
        code, out, members = O.gunzip(raw, size + 16)
        assert (code, members, len(out), zlib.crc32(out)) == (0, nm, size, crc)
        assert out == gzip.decompress(raw)


def test_inflate_known_answers():
    assert O.zlib_decompress(b'x\x9cK\xcb\xcf\x07\x00\x02\x82\x01E', 16) == (0, b"foo")
    # "abc" with the last Adler-32 byte missing (test_zlib_compliance.py:497 feeds it to a decompressobj)
    assert O.zlib_decompress(b"x\x9cKLJ\x06\x00\x02M\x01", 16) == (O.BUF_ERROR, b"abc")
    assert O.zlib_decompress(b"x\x9cKLJ\x06\x00\x02M\x01\x27", 16) == (0, b"abc")
    gz_extra = (b'\x1f\x8b\x08\x04\xb2\x17cQ\x02\xff\x09\x00XX\x05\x00Extra\x0bI-.\x01\x002\xd1Mx\x04\x00\x00\x00')
    assert O.gunzip(gz_extra, 16)[:2] == (0, b"Test")


def test_golden_vectors():
    """tests/golden/inflate_vectors.json (made by tests/golden/make_golden.py with the system zlib):
    streams covering stored / fixed / dynamic blocks, every strategy, flush points, dictionaries, all
    gzip header fields, padding and multi-member layouts, each with the SHA-256 of the expected output."""
    vec = json.load(open(os.path.join(GOLDEN, "inflate_vectors.json")))
    assert len(vec) >= 40
    for v in vec:
        blob = bytes.fromhex(v["hex"])
        if v["kind"] == "raw":
            code, out, used = O.inflate_raw(blob, v["size"] + 8, bytes.fromhex(v.get("zdict", "")))
            assert code == 1 and used == len(blob), v["name"]
        elif v["kind"] == "zlib":
            code, out = O.zlib_decompress(blob, v["size"] + 8)
            assert code == 0, v["name"]
        else:
            code, out, nm = O.gunzip(blob, max(v["size"], 1 << 17) + 8)
            assert code == v.get("code", 0), v["name"]
            if code:
                continue
        assert len(out) == v["size"] and hashlib.sha256(out).hexdigest() == v["sha256"], v["name"]


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, -1])
def test_deflate_round_trips_through_zlib(fastq, level):
    rng = np.random.default_rng(level + 2)
    cases = [fastq[:131072], fastq[5000:5000 + 70001], bytes(131072), rng.bytes(40000), b"", b"a", b"abcd" * 9000,
             fastq[:1000] + rng.bytes(3000) + fastq[:1000]]
    for data in cases:
        for flags in (0, 1):
            c, crc = O.deflate_unit(data, b"", level, flags)
            d = zlib.decompressobj(-15)
            assert d.decompress(c) == data and crc == zlib.crc32(data)
            assert d.eof == bool(flags)
            if not flags:
                assert c.endswith(b"\x00\x00\xff\xff")
    blk0, blk1 = fastq[:131072], fastq[131072:262144]
    c, _ = O.deflate_unit(blk1, blk0[-32768:], level)
    assert zlib.decompressobj(-15, zdict=blk0[-32768:]).decompress(c) == blk1


def test_ratio_not_worse_than_zlib_at_same_level(fastq):
    """the level table is calibrated so that ratio(level) >= zlib's at the same level."""
    data = fastq[:1 << 20]
    for level in (1, 3, 6, 9):
        ours = len(O.deflate_stream(data, level))
        assert ours <= len(zlib.compress(data, level)) - 6, level


def test_stream_and_window_limit(fastq):
    data = fastq[:400000]
    for wb in (9, 11, 15):
        s = O.deflate_stream(data, 6, window_bits=wb)
        assert zlib.decompress(s, -wb) == data
    code, out, used = O.inflate_raw(O.deflate_stream(data, 6), len(data) + 1)
    assert (code, out) == (1, data)


def test_incompressible_block_fits_reference_buffer():
    """gzip_ng_threaded.py:229-231 sizes the block buffer B + max(B//10, 500); tests/test_gzip_ng_threaded.py:82-97."""
    rng = np.random.default_rng(0)
    for b in (8192, 65536, 131072):
        c, _ = O.deflate_unit(rng.bytes(b), b"", 3)
        assert len(c) < b + max(b // 10, 500)


def test_gzip_framing_errors():
    raw = gzip.compress(b"hello world" * 500, mtime=0)
    assert O.gunzip(raw, 10000)[0] == 0
    assert O.gunzip(b"\x1f\x8c" + raw[2:], 10000)[0] == O.GZ_BAD_MAGIC
    assert O.gunzip(raw[:2] + b"\x07" + raw[3:], 10000)[0] == O.GZ_BAD_METHOD
    assert O.gunzip(raw[:-8] + bytes([raw[-8] ^ 1]) + raw[-7:], 10000)[0] == O.GZ_BAD_CRC
    assert O.gunzip(raw[:-4] + bytes([raw[-4] ^ 1]) + raw[-3:], 10000)[0] == O.GZ_BAD_LENGTH
    assert O.gunzip(raw[:-3], 10000)[0] == O.GZ_TRUNCATED
    assert O.gunzip(raw + bytes(37) + raw, 20000)[2] == 2
    assert O.inflate_raw(b"\x07", 10)[0] == O.DATA_ERROR
