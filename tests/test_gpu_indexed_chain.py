"""The default writer's framing -- ONE deflate stream of dict-chained, sync-flushed blocks (gzip_ng_threaded.py:299-338) -- decoded
unit-parallel with the writer's own segment index (zngamd_inflate_units_indexed_dev): inflate output is unique, so parity is
"equals the input", for every kind of unit (dynamic, fixed, stored, empty, short last one), with and without a dictionary in front
of the stream; the same stream must decode with the system zlib (the flat headers are ordinary RFC 1951) and through the engine's
index-free path; damage is reported, never decoded to something else."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B = 131072
FLATHDR = 2
SEG2K = 16


def _mix(n, seed):
    from zlib_ng_amd import corpus
    rng = np.random.default_rng(seed)
    parts = [corpus.text(n // 3, seed + 1).tobytes(), bytes(n // 8), rng.integers(0, 256, n // 6, dtype=np.uint8).tobytes(),
             corpus.fastq(n // 4, seed + 2).tobytes(), (b"abcdefgh" * 4096)[:n // 16]]
    out = b"".join(parts)
    return (out + corpus.text(max(0, n - len(out)), seed + 3).tobytes())[:n]


def _compress_indexed(ctx, data, level, block=B, dict_first=b"", flag=SEG2K):
    """-> (stream bytes incl. a final empty block, unit_in_len, unit_out_len, device index)"""
    buf = dict_first + data
    off0 = len(dict_first)
    blocks, off = [], off0
    while off < len(buf):
        n = min(block, len(buf) - off)
        blocks.append((off, n, min(32768, off), flag))
        off += n
    outs, crcs, ovf = ctx.deflate_blocks(buf, blocks, level, block + block // 8 + 600)
    assert not ovf
    nu = len(blocks)            # (blocks of at most a unit's size: a unit each -- the host call reports sizes per block, and the index is per unit)
    assert block <= B
    d_index = ctx.deflate_index(nu)
    for (o, n, d, _), c, crc in zip(blocks, outs, crcs):
        assert crc == zlib.crc32(buf[o:o + n])
    return b"".join(outs) + b"\x03\x00", [len(c) for c in outs], [b[1] for b in blocks], d_index


def _decode(ctx, stream, uin, uout, d_index, dict_first=b""):
    from zlib_ng_amd import devmem
    d_def = devmem.from_host(ctx, stream + bytes(64))
    total = sum(uout)
    d_out = devmem.empty(ctx, total + 64)
    d_dict = devmem.from_host(ctx, dict_first) if dict_first else None
    r, n = ctx.inflate_units_indexed_dev(d_def.ptr, len(stream), uin, uout, d_index.ptr, d_out.ptr, total,
                                         d_dict.ptr if d_dict else None, len(dict_first))
    return r, n, (d_out[0:n].cpu().tobytes() if r == 1 and n <= total else b"")


@pytest.mark.parametrize("flag", [SEG2K, FLATHDR])
@pytest.mark.parametrize("level", [1, 6, 9])
def test_indexed_chain_equals_input(ctx, level, flag):
    """ordinary (run-length coded) block headers, as the threaded writer leaves them, and flat ones"""
    from zlib_ng_amd import _lib
    data = _mix(5 * B + 12345, seed=level)
    stream, uin, uout, d_index = _compress_indexed(ctx, data, level, flag=flag)
    assert zlib.decompressobj(-15).decompress(stream) == data           # any inflater reads it
    r, n, back = _decode(ctx, stream, uin, uout, d_index)
    assert r == _lib.STREAM_END and n == len(data) and back == data


def test_stored_fixed_empty_and_tiny_units(ctx):
    from zlib_ng_amd import _lib
    rng = np.random.default_rng(7)
    pieces = [rng.integers(0, 256, B, dtype=np.uint8).tobytes(),            # stored (two blocks of <= 65 535)
              b"ab" * 40,                                                   # a tiny unit: fixed code
              _mix(B, 3), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), b"x", _mix(3 * B + 17, 5)]
    # blocks of the pieces' own sizes (each <= one unit), chained through 32 KiB of the input before them
    buf = b"".join(pieces)
    blocks, off = [], 0
    for p in pieces:
        for o in range(0, len(p), B):
            n = min(B, len(p) - o)
            blocks.append((off + o, n, min(32768, off + o), SEG2K))
        off += len(p)
    outs, crcs, ovf = ctx.deflate_blocks(buf, blocks, 6, B + B // 8 + 600)
    assert not ovf
    d_index = ctx.deflate_index(len(blocks))
    stream = b"".join(outs) + b"\x03\x00"
    assert zlib.decompressobj(-15).decompress(stream) == buf
    r, n, back = _decode(ctx, stream, [len(c) for c in outs], [b[1] for b in blocks], d_index)
    assert r == _lib.STREAM_END and back == buf


def test_dictionary_in_front_of_the_stream(ctx):
    from zlib_ng_amd import _lib
    d = _mix(40000, 11)
    data = d[-9000:] * 3 + _mix(2 * B + 999, 12)
    stream, uin, uout, d_index = _compress_indexed(ctx, data, 6, dict_first=d[-32768:])
    zd = zlib.decompressobj(-15, zdict=d[-32768:])
    assert zd.decompress(stream) == data
    r, n, back = _decode(ctx, stream, uin, uout, d_index, dict_first=d[-32768:])
    assert r == _lib.STREAM_END and back == data


def test_many_units_in_batches(ctx, monkeypatch):
    """more units than one batch of the decoder holds (ZNGAMD_UNIT_BATCH is read once per process: the default, 16 384, needs
    2 GiB; 260 units of 16 KiB go through the same code with blocks below a unit's size)"""
    from zlib_ng_amd import _lib
    data = _mix(260 * 16384, 21)
    stream, uin, uout, d_index = _compress_indexed(ctx, data, 6, block=16384)
    assert zlib.decompressobj(-15).decompress(stream) == data
    r, n, back = _decode(ctx, stream, uin, uout, d_index)
    assert r == _lib.STREAM_END and back == data


def test_damage_is_reported(ctx):
    from zlib_ng_amd import _lib
    data = _mix(3 * B, 31)
    stream, uin, uout, d_index = _compress_indexed(ctx, data, 6)
    rng = np.random.default_rng(5)
    verdicts = set()
    for _ in range(24):
        bad = bytearray(stream)
        pos = int(rng.integers(0, len(stream) - 2))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        r, n, back = _decode(ctx, bytes(bad), uin, uout, d_index)
        # raw deflate carries no checksum: a flip that turns one literal code into another of its length decodes, to other bytes,
        # with ANY inflater (the gzip layer's CRC-32 is what catches it) -- so a "fine" verdict must come with exactly the bytes
        # the system zlib decodes
        try:
            ref = zlib.decompressobj(-15).decompress(bytes(bad))
        except zlib.error:
            ref = None
        if r == _lib.STREAM_END:
            assert ref is not None and back == ref, pos
        else:
            assert r in (_lib.E_INDEX, _lib.DATA_ERROR), (r, pos)
        verdicts.add(r)
    assert verdicts - {_lib.STREAM_END}, "no flip was noticed"
    # a wrong index: the neighbour's rows
    r, n, back = _decode(ctx, stream, uin[::-1], uout, d_index)
    assert r != _lib.STREAM_END or back == data


# ---- the same through files: the threaded writer leaves the index behind its data member, the reader finds and uses it ----

def _write(T, data, **kw):
    import io
    bio = io.BytesIO()
    with T.open(bio, "wb", compresslevel=6, threads=1, **kw) as f:
        for o in range(0, len(data), 1 << 20):
            f.write(data[o:o + (1 << 20)])
    return bio.getvalue()


def test_writer_leaves_an_index_any_reader_skips(ctx):
    import gzip
    import io
    import struct
    from zlib_ng_amd import gzip_ng_threaded as T, _lib
    data = _mix(9 * B + 4321, 41)
    blob = _write(T, data, block_size=B)
    # layout: the reference's (header 1f8b0800 00000000 ff xfl | blocks | 03 00 | crc isize), then empty members with FEXTRA, the
    # locator, and the plain empty member close() leaves (gzip_ng_threaded.py:340-342)
    assert blob[:10] == bytes.fromhex("1f8b0800" "00000000" "ff00")
    assert blob.endswith(bytes.fromhex("1f8b0800" "00000000" "ff00" "0300" "00000000" "00000000"))
    loc = blob[-20 - _lib.INDEX_LOCATOR_BYTES:-20]
    assert loc[:4] == b"\x1f\x8b\x08\x04" and loc[12:14] == b"ZA" and loc[16:18] == b"\x03\x02"
    nm, nu, ibytes, dbytes = struct.unpack("<IIQQ", loc[20:44])
    assert nu == 10 and blob[dbytes - 10:dbytes - 8] == b"\x03\x00"
    assert struct.unpack("<II", blob[dbytes - 8:dbytes]) == (zlib.crc32(data), len(data))
    assert gzip.decompress(blob) == data                     # the system gzip reads all of it: one member of data, empty ones behind
    got = _lib.parse_index_tail(io.BytesIO(blob), 0, len(blob))
    assert got is not None and len(got[0]) == 10 and int(got[1].sum()) == len(data)
    # the r05 bytes on request: no index, nothing between the trailer and the plain empty member
    exact = _write(T, data, block_size=B, exact_framing=True)
    assert b"ZA" not in exact[-200:] and gzip.decompress(exact) == data and _lib.parse_index_tail(io.BytesIO(exact), 0, len(exact)) is None


def test_reader_decodes_windows_with_the_index(ctx, monkeypatch):
    import io
    from zlib_ng_amd import gzip_ng_threaded as T, gzip_ng, _lib
    data = _mix(40 * B + 777, 43)
    blob = _write(T, data, block_size=B)
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(1 << 20))         # a member much larger than the window: it is read window by window
    ctx.L.zngamd_indexed_units(ctx.h, 1)
    with T.open(io.BytesIO(blob), "rb") as f:
        back = f.read()
    assert back == data
    used = ctx.L.zngamd_indexed_units(ctx.h, 1)
    assert used == 41, used                                          # every unit, the first window's included
    with gzip_ng.open(io.BytesIO(blob), "rb") as f:
        assert f.read() == data
    # a damaged index is no index: the file still reads
    bad = bytearray(blob)
    at = blob.rindex(b"ZA", 0, len(blob) - 100)
    bad[at + 40] ^= 0x55
    ctx.L.zngamd_indexed_units(ctx.h, 1)
    with T.open(io.BytesIO(bytes(bad)), "rb") as f:
        assert f.read() == data
    # damage in the data is reported as the CRC check reports it (the index changes how, not whether)
    bad = bytearray(blob)
    bad[len(blob) // 3] ^= 0x10
    with pytest.raises(Exception):
        with T.open(io.BytesIO(bytes(bad)), "rb") as f:
            f.read()
    monkeypatch.setenv("ZNGAMD_NO_INDEX", "1")
    ctx.L.zngamd_indexed_units(ctx.h, 1)
    with T.open(io.BytesIO(blob), "rb") as f:
        assert f.read() == data
    assert ctx.L.zngamd_indexed_units(ctx.h, 1) == 0


def test_reader_many_windows_of_varying_size(ctx, monkeypatch):
    """windows that hold a different number of units each time (the decoder's unit areas grow, and with them go the markers in
    front of every unit): a file of 700 units read through windows of 3 MiB, then 5 MiB"""
    import io
    from zlib_ng_amd import gzip_ng_threaded as T, corpus
    data = corpus.text(700 * B + 99, seed=77).tobytes()
    blob = _write(T, data, block_size=B)
    for win in (3 << 20, 5 << 20, 1 << 20):
        monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(win))
        ctx.L.zngamd_indexed_units(ctx.h, 1)
        with T.open(io.BytesIO(blob), "rb") as f:
            back = f.read()
        assert back == data, win
        assert ctx.L.zngamd_indexed_units(ctx.h, 1) == 701, win


def test_seg2k_units_match_the_oracle(ctx):
    """FLAG_SEG2K (2 KiB segments in a unit of any size) is part of the codec's specification: HIP == oracle byte for byte"""
    from oracle import oracle as O
    data = _mix(3 * 20000 + 777, 91)
    blocks = [(o, min(20000, len(data) - o), min(32768, o), SEG2K) for o in range(0, len(data), 20000)]
    outs, crcs, ovf = ctx.deflate_blocks(data, blocks, 6, 40000)
    assert not ovf
    for (o, n, d, f), got in zip(blocks, outs):
        exp, ecrc = O.deflate_unit(data[o:o + n], data[o - d:o], 6, SEG2K)
        assert got == exp
    plain, _, _ = ctx.deflate_blocks(data, [(o, n, d, 0) for o, n, d, _ in blocks], 6, 40000)
    assert zlib.decompressobj(-15).decompress(b"".join(outs) + b"\x03\x00") == data
    assert plain != outs                       # (small units cut otherwise: the flag changes their tokens)


def test_two_writers_on_one_context_keep_their_own_index(ctx):
    """Two threaded writers at work at once share the process's context: a batch's segment-index records must be those of ITS
    engine call (zngamd_deflate_blocks_packed_indexed), not of whichever call the context made last."""
    import gzip
    import io
    import threading
    from zlib_ng_amd import gzip_ng_threaded as T
    datas = [_mix(43 * B + 1001, 61), _mix(43 * B + 1001, 67), _mix(37 * B + 5, 71)]      # (two of the same shape: a foreign index of the same unit count would go unnoticed by a count check)
    blobs = [None] * len(datas)

    def work(i):
        bio = io.BytesIO()
        with T.open(bio, "wb", block_size=B, threads=1) as f:
            for o in range(0, len(datas[i]), 3 * B + 17):       # small writes: many batches per file, so that the writers' engine calls interleave
                f.write(datas[i][o:o + 3 * B + 17])
        blobs[i] = bio.getvalue()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(datas))]
    for t in ths: t.start()
    for t in ths: t.join()
    for i, data in enumerate(datas):
        assert gzip.decompress(blobs[i]) == data
        ctx.L.zngamd_indexed_units(ctx.h, 1)
        with T.open(io.BytesIO(blobs[i]), "rb") as f:
            assert f.read() == data
        assert ctx.L.zngamd_indexed_units(ctx.h, 1) > 0       # ... and it was the index that decoded it
