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
        ctx.decode_paths(True)
        code, out, nm = ctx.gunzip(stream, len(data))                # two-pass path
        assert code == 0 and out == data and nm == max(1, -(-len(data) // bs))
        if data and data in fastq:                                   # (incompressible members are stored blocks: sequential decoder)
            assert ctx.decode_paths(True)["indexed"] == nm, "the indexed path did not decode these members"
        ocode, oout, onm = O.gunzip(stream, len(data) + 1)
        assert ocode == 0 and oout == data
    # runs of more than 510 literals in front of a match (the queue entry of a match counts 9 bits of literals: longer runs
    # get an entry of their own): 700 random bytes, then text, in every 2 KiB segment
    mix = b"".join(rng.bytes(700) + fastq[i * 1348:(i + 1) * 1348] for i in range(150))
    stream = ctx.gzip_members(mix, 131072, level)
    ctx.decode_paths(True)
    code, out, nm = ctx.gunzip(stream, len(mix))
    assert code == 0 and out == mix and ctx.decode_paths(True)["indexed"] == nm == 3
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


def _bgzf(data, block=60000, level=6):
    """Minimal BGZF writer (SAM/BAM spec): members with a 'BC' subfield holding the member size - 1."""
    import struct
    out = []
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]
    for ch in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(ch) + co.flush()
        size = 18 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1) +
                   body + struct.pack("<II", zlib.crc32(ch), len(ch)))
    return b"".join(out)


def test_bgzf_members_one_launch(ctx, fastq):
    from oracle import oracle as O
    data = fastq[:1500000]
    blob = _bgzf(data)
    assert gzip.decompress(blob) == data
    code, out, nm = ctx.gunzip(blob, len(data))
    assert (code, out, nm) == (0, data, -(-len(data) // 60000) + 1)
    # too small an output buffer: the needed size comes back
    code, out, nm = ctx.gunzip(blob, 1000)
    assert code == -5 and ctx.last_needed == len(data)
    # corrupt the 5th member's payload and the 9th member's CRC: output stops at the first bad member
    bad = bytearray(blob)
    offs, pos = [], 0
    while pos < len(blob):
        offs.append(pos)
        pos += (blob[pos + 16] | blob[pos + 17] << 8) + 1
    bad[offs[4] + 40] ^= 0x04
    code, out, nm = ctx.gunzip(bytes(bad), len(data))
    ocode, oout, onm = O.gunzip(bytes(bad), len(data) + 1)
    assert code != 0 and ocode != 0 and nm == 4 and out == data[:4 * 60000]
    bad = bytearray(blob)
    bad[offs[9] - 8] ^= 1          # CRC field of member 8
    code, out, nm = ctx.gunzip(bytes(bad), len(data))
    assert code == -104 and nm == 8 and out == data[:8 * 60000]


def test_indexed_stream_fuzz(ctx, fastq):
    """Random single-byte corruptions of an indexed member stream (headers, index, payload, trailers):
    the engine must answer like the oracle -- never crash, never return wrong bytes as good."""
    from oracle import oracle as O
    data = fastq[:300000]
    stream = ctx.gzip_members(data, 131072, 6)
    rng = np.random.default_rng(99)
    for trial in range(120):
        bad = bytearray(stream)
        pos = int(rng.integers(0, len(bad))) if trial % 3 else int(rng.integers(0, 288))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        code, out, nm = ctx.gunzip(bytes(bad), len(data) + 4096)
        ocode, oout, onm = O.gunzip(bytes(bad), len(data) + 4096)
        assert (code == 0) == (ocode == 0), (pos, code, ocode)
        if code == 0:
            assert out == oout


def _sync_flushed_gzip(data, step, level=6, mark=False):
    """One gzip member whose deflate stream has a sync-flush point every `step` bytes and keeps its history
    across them (what pigz / gzip_ng_threaded style writers produce; here made with the system zlib)."""
    import struct
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    parts = [b"\x1f\x8b\x08\x00" + bytes(4) + b"\x00\xff"]
    for i in range(0, len(data), step):
        parts.append(co.compress(data[i:i + step]) + co.flush(zlib.Z_SYNC_FLUSH))
    parts.append(co.flush() + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF))
    return b"".join(parts)


def test_chunk_parallel_inflate_of_sync_flushed_stream(ctx, fastq):
    """a single member with sync-flush points is decoded chunk-parallel (markers for references
    into the previous chunk), result identical to the sequential decoders."""
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    data = fastq + fastq[:500000]
    for step, level in ((100000, 6), (37000, 9), (250000, 1), (8000, 6)):      # (pieces of 8 000 bytes are too small to be chunks of their own: count pass + merged chunks)
        blob = _sync_flushed_gzip(data, step, level)
        assert gzip.decompress(blob) == data
        ctx.decode_paths(True)
        ctx.profiling(True); ctx.kernel_times(True)
        code, out, nm = ctx.gunzip(blob, len(data))
        kt = ctx.kernel_times(True); ctx.profiling(False)
        assert (code, nm) == (0, 1) and out == data
        # [count,] decode, propagate + resolve ran: one decoding pass where the sync-delimited pieces are chunks as they are
        assert kt["inflate"][1] >= (3 if step < 20000 else 2) and kt["scan"][1] >= 1, kt
        assert ctx.decode_paths(True)["chunked"] == 1
    # payload full of `00 00 FF FF` look-alikes: stored blocks (level 0) and compressed
    tricky = (b"\x00\x00\xff\xff" * 50 + rng.bytes(3000)) * 400
    for level in (0, 6):
        blob = _sync_flushed_gzip(tricky, 90000, level)
        code, out, nm = ctx.gunzip(blob, len(tricky))
        assert code == 0 and out == tricky
    # output buffer too small: needed size reported
    blob = _sync_flushed_gzip(data, 100000, 6)
    code, out, nm = ctx.gunzip(blob, 1000)
    assert code == -5 and ctx.last_needed == len(data)
    # two such members back to back + padding
    code, out, nm = ctx.gunzip(blob + bytes(10) + blob, 2 * len(data))
    assert code == 0 and nm == 2 and out == data + data
    # corruption anywhere gives the oracle's verdict (the chunk path hands over to the sequential decoder)
    for trial in range(25):
        bad = bytearray(blob)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        code, out, nm = ctx.gunzip(bytes(bad), len(data) + 4096)
        ocode, oout, onm = O.gunzip(bytes(bad), len(data) + 4096)
        assert (code == 0) == (ocode == 0), (pos, code, ocode)
        if code == 0:
            assert out == oout


def test_threaded_writer_output_is_read_chunk_parallel(ctx, fastq):
    """What our gzip_ng_threaded writer (reference framing: one member, dictionary-chained sync-flushed blocks,
    trailing empty member) produces is read back through the chunk-parallel path."""
    import io
    from zlib_ng_amd import gzip_ng_threaded
    data = fastq + fastq
    bio = io.BytesIO()
    with gzip_ng_threaded.open(bio, "wb", compresslevel=6, threads=8, block_size=128 * 1024, exact_framing=True) as f:
        f.write(data)
    blob = bio.getvalue()
    ctx.profiling(True); ctx.kernel_times(True)
    code, out, nm = ctx.gunzip(blob, len(data))
    kt = ctx.kernel_times(True); ctx.profiling(False)
    assert code == 0 and out == data and nm == 2 and kt["scan"][1] >= 1
    with gzip_ng_threaded.open(io.BytesIO(blob), "rb") as f:
        assert f.read() == data
    # the default framing: the same member, then the empty members that hold the segment index, the locator, the plain empty one
    bio = io.BytesIO()
    with gzip_ng_threaded.open(bio, "wb", compresslevel=6, threads=8, block_size=128 * 1024) as f:
        f.write(data)
    blob = bio.getvalue()
    code, out, nm = ctx.gunzip(blob, len(data))
    assert code == 0 and out == data and nm == 4
    with gzip_ng_threaded.open(io.BytesIO(blob), "rb") as f:
        assert f.read() == data


def test_chunk_parallel_inflate_of_ordinary_gzip(ctx, fastq):
    """SURVEY.md 8f-3, general case: a plain `gzip` member has no sync points; chunk starts come from the
    bit-level dynamic-block-header finder.  The result is the system zlib's, and the chunk kernels really ran."""
    from zlib_ng_amd import corpus
    rng = np.random.default_rng(33)
    text = corpus.text(12 << 20, seed=9).tobytes()
    mixed = corpus.mixed(12 << 20, seed=4).tobytes()
    cases = [(text, 6), (text, 1), (text, 9), (mixed, 6), (fastq + fastq + fastq, 6)]
    ran = 0
    for data, level in cases:
        blob = gzip.compress(data, level)
        ctx.decode_paths(True)
        code, out, nm = ctx.gunzip(blob, len(data))
        assert (code, nm) == (0, 1) and out == data
        ran += ctx.decode_paths(True)["chunked"]
    assert ran == len(cases), ran
    # the golden single-member file, as shipped by the reference's tests
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "test.fastq.gz"), "rb").read()
    code, out, nm = ctx.gunzip(raw, len(fastq))
    assert code == 0 and out == fastq
    # corrupted ordinary streams: same verdict as the oracle
    from oracle import oracle as O
    blob = gzip.compress(text[:3 << 20], 6)
    for trial in range(20):
        bad = bytearray(blob)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        code, out, nm = ctx.gunzip(bytes(bad), (3 << 20) + 4096)
        ocode, oout, onm = O.gunzip(bytes(bad), (3 << 20) + 4096)
        assert (code == 0) == (ocode == 0), (pos, code, ocode)
        if code == 0:
            assert out == oout


def test_chunk_parallel_inflate_with_more_chunks_than_wavefronts():
    """One stream of 2 640 sync-delimited blocks (330 MiB): more chunks than the middle footprint of the marker decoder holds at
    once, so the smallest one runs (384-bit sub-sequences, queue of 768, 256 symbols of ring, tables of 9 / 8 index bits) --
    every byte compared, both through the one-shot call and through a reader whose window takes the stream whole."""
    import io
    from zlib_ng_amd import corpus, gzip_ng, gzip_ng_threaded
    base = corpus.text(66 << 20, seed=23).tobytes()
    data = base * 5
    bio = io.BytesIO()
    with gzip_ng_threaded.open(bio, "wb", compresslevel=6, threads=8, block_size=128 * 1024) as f:
        f.write(data)
    blob = bio.getvalue()
    out = gzip_ng.decompress(blob)
    assert len(out) == len(data) and out == data
    del out
    os.environ["ZNGAMD_READ_WINDOW"] = str(1 << 30)
    try:
        with gzip_ng.open(io.BytesIO(blob), "rb") as g:
            got = g.read()
    finally:
        del os.environ["ZNGAMD_READ_WINDOW"]
    assert got == data


def test_indexed_members_followed_by_other_members(ctx):
    """A file that starts with indexed members and goes on with members of another writer (appended later, `cat a.gz b.gz`):
    the indexed run is decoded by the member decoder, what follows by the member loop -- one result, the right order."""
    import gzip
    from zlib_ng_amd import corpus, gzip_ng
    data = corpus.text(1500000, seed=31).tobytes()
    head = ctx.gzip_members(data[:1000000], 131072, 6)
    tail_plain = gzip.compress(data[1000000:1300000], 6, mtime=0)
    tail_indexed = ctx.gzip_members(data[1300000:], 131072, 6)
    blob = bytes(head) + tail_plain + bytes(tail_indexed)
    ctx.decode_paths(True)
    assert gzip_ng.decompress(blob) == data
    paths = ctx.decode_paths(True)
    assert paths["indexed"] == 8 and paths["chunked"] + paths["sequential"] + paths["bgzf"] >= 3, paths
    assert gzip.decompress(blob) == data
