"""GPU parity: HIP inflate (sequential decoder, two-pass indexed members, gzip reader) against the
oracle / the reference's fixtures / the system zlib.  All compute calls go through the C ABI."""
import gzip
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_inflate_raw_vs_zlib_streams(ctx, fastq):
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    cases = [fastq[:200000], bytes(100000), rng.bytes(70000), b"", b"a", fastq[:3] * 30000]
    for data in cases:
        for level, strategy in ((1, 0), (6, 0), (9, 0), (0, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            comp = co.compress(data) + co.flush()
            code, out, used, crc, ad = ctx.inflate_raw(comp, len(data) + 16)
            ocode, oout, oused = O.inflate_raw(comp, len(data) + 16)
            assert (code, out, used) == (ocode, oout, oused) == (1, data, len(comp))
            assert crc == zlib.crc32(data) and ad == zlib.adler32(data)


def test_inflate_raw_dictionary_truncation_and_errors(ctx, fastq):
    from oracle import oracle as O
    data = fastq[:50000]
    zd = fastq[60000:90000]
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0, zd)
    comp = co.compress(data) + co.flush()
    code, out, used, _, _ = ctx.inflate_raw(comp, len(data) + 1, zd)
    assert (code, out) == (1, data)
    # truncated input -> BUF_ERROR with the same partial output as the oracle
    for cut in (1, 10, len(comp) // 2, len(comp) - 1):
        code, out, used, _, _ = ctx.inflate_raw(comp[:cut], len(data) + 1, zd)
        ocode, oout, _ = O.inflate_raw(comp[:cut], len(data) + 1, zd)
        assert code == ocode == -5
        assert data.startswith(out)
    # output too small
    code, out, used, _, _ = ctx.inflate_raw(comp, 1000, zd)
    assert code == -5 and out == data[:1000]
    # corrupt data: same verdict as the oracle
    rng = np.random.default_rng(5)
    plain = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp2 = plain.compress(data) + plain.flush()
    for _ in range(40):
        bad = bytearray(comp2)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        code, out, used, _, _ = ctx.inflate_raw(bytes(bad), len(data) + 4096)
        ocode, oout, _ = O.inflate_raw(bytes(bad), len(data) + 4096)
        assert code == ocode, (pos, code, ocode)
        if code == 1:
            assert out == oout
    # invalid block type, bad stored lengths
    assert ctx.inflate_raw(b"\x07", 10)[0] == -3
    assert ctx.inflate_raw(b"\x01\x05\x00\x00\x00hello", 10)[0] == -3


@pytest.mark.parametrize("name", ["test.fastq.gz", "concatenated.fastq.gz", "test.fastq.bgzip.gz"])
def test_reference_gzip_fixtures(ctx, name):
    raw = open(os.path.join(GOLDEN, name), "rb").read()
    exp = gzip.decompress(raw)
    code, out, nm = ctx.gunzip(raw, len(exp))
    assert code == 0 and out == exp
    assert nm == {"test.fastq.gz": 1, "concatenated.fastq.gz": 2, "test.fastq.bgzip.gz": 56}[name]
    if name == "test.fastq.gz":
        assert len(out) == 3578369 and zlib.crc32(out) == 0x473f3477


def test_gzip_reader_errors(ctx, fastq):
    from oracle import oracle as O
    raw = gzip.compress(fastq[:30000], 6, mtime=0)
    n = 30000
    cases = {
        "magic": b"\x1f\x8c" + raw[2:], "method": raw[:2] + b"\x07" + raw[3:],
        "crc": raw[:-8] + bytes([raw[-8] ^ 1]) + raw[-7:], "length": raw[:-4] + bytes([raw[-4] ^ 1]) + raw[-3:],
        "trunc_trailer": raw[:-3], "trunc_body": raw[:len(raw) // 2], "trunc_header": raw[:5],
        "padded": raw + bytes(100), "two": raw + raw,
    }
    for k, blob in cases.items():
        code, out, nm = ctx.gunzip(blob, 2 * n + 10)
        ocode, oout, onm = O.gunzip(blob, 2 * n + 10)
        assert code == ocode, (k, code, ocode)
        if code == 0:
            assert out == oout and nm == onm


@pytest.mark.parametrize("level", [1, 6])
def test_indexed_members_round_trip(ctx, fastq, level):
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    for data, bs in ((fastq[:1000000], 131072), (fastq[:300001], 65536), (fastq[:5000], 131072),
                     (fastq[:200000] + rng.bytes(140000) + bytes(150000), 131072), (b"", 131072)):
        stream = ctx.gzip_members(data, bs, level)
        assert gzip.decompress(stream) == data                      # any gzip reader accepts it
        code, out, nm = ctx.gunzip(stream, len(data))                # two-pass path
        assert code == 0 and out == data and nm == max(1, -(-len(data) // bs))
        ocode, oout, onm = O.gunzip(stream, len(data) + 1)
        assert ocode == 0 and oout == data
    # corrupt one payload byte of an indexed stream: CRC / data error must surface
    stream = bytearray(ctx.gzip_members(fastq[:400000], 131072, level))
    stream[5000] ^= 0x10
    code, out, nm = ctx.gunzip(bytes(stream), 400000)
    ocode, _, _ = O.gunzip(bytes(stream), 400001)
    assert code != 0 and ocode != 0


def test_golden_vectors(ctx):
    """tests/golden/inflate_vectors.json through the HIP decoders (same vectors pin the oracle on CPU)."""
    import hashlib
    import json
    vec = json.load(open(os.path.join(GOLDEN, "inflate_vectors.json")))
    for v in vec:
        blob = bytes.fromhex(v["hex"])
        if v["kind"] == "raw":
            code, out, used, _, _ = ctx.inflate_raw(blob, v["size"] + 8, bytes.fromhex(v.get("zdict", "")))
            assert code == 1 and used == len(blob), v["name"]
        elif v["kind"] == "zlib":
            from zlib_ng_amd import zlib_ng
            out = zlib_ng.decompress(blob)
        else:
            code, out, nm = ctx.gunzip(blob, max(v["size"], 1 << 17) + 8)
            assert code == v.get("code", 0), v["name"]
            if code:
                continue
        assert len(out) == v["size"] and hashlib.sha256(out).hexdigest() == v["sha256"], v["name"]
