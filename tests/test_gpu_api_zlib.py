"""zlib_ng / gzip_ng one-shot API on the GPU engine against CPython's zlib / gzip, after the reference's
tests/test_compat.py (cross-decode both directions over levels x wbits x sizes, checksums over seeds) and
the known-answer tests of tests/test_zlib_compliance.py / test_gzip_ng.py."""
import gzip
import io
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 5, 100, 1023, 4096, 65536, 131072, 131073, 500001]
SEEDS = [int(s) for s in open(os.path.join(GOLDEN, "seeds.txt")).read().split()][:12] + \
        [-(2 ** 64 + 5), -3, -1, 0, 1, 2 ** 64 + 5]


@pytest.fixture(scope="module")
def Z():
    from zlib_ng_amd import zlib_ng
    return zlib_ng


@pytest.fixture(scope="module")
def G():
    from zlib_ng_amd import gzip_ng
    return gzip_ng


def test_checksum_kats(Z):
    # reference tests/test_zlib_compliance.py:94-120
    assert Z.crc32(b"penguin", 0) == 0x0e5c1a120 & 0xFFFFFFFF
    assert Z.crc32(b"penguin", 1) == 0x43b6aa94
    assert Z.adler32(b"penguin", 0) == 0x0bcf02f6
    assert Z.adler32(b"penguin", 1) == 0x0bd602f7
    assert Z.crc32(b"penguin") == Z.crc32(b"penguin", 0)
    assert Z.adler32(b"penguin") == Z.adler32(b"penguin", 1)
    assert Z.crc32(b"abcdefghijklmnop") == 2486878355
    assert Z.crc32(b"spam") == 1138425661
    assert Z.adler32(b"spam") == 72286642
    foo = b"abcdefghijklmnop"
    assert Z.adler32(foo + foo) == 3573550353
    assert Z.crc32(b"") == 0 and Z.adler32(b"") == 1
    assert Z.crc32_combine(Z.crc32(b"abc"), Z.crc32(b"defg"), 4) == Z.crc32(b"abcdefg")


def test_checksums_vs_stdlib_over_seeds(Z, fastq):
    for n in (0, 8, 64, 4096, 2 ** 17, 2 ** 19):
        for seed in SEEDS:
            assert Z.crc32(fastq[:n], seed) == zlib.crc32(fastq[:n], seed)
            assert Z.adler32(fastq[:n], seed) == zlib.adler32(fastq[:n], seed)


@pytest.mark.parametrize("level", [-1, 0, 1, 3, 6, 9])
def test_compress_decodes_with_stdlib(Z, fastq, level):
    for wbits in (9, 12, 15, -9, -15, 25, 31):
        for n in SIZES:
            c = Z.compress(fastq[:n], level, wbits)
            assert zlib.decompress(c, wbits) == fastq[:n], (level, wbits, n)


def test_stdlib_compressed_decodes(Z, fastq):
    for level in (0, 1, 6, 9):
        for wbits in (9, 15, -12, -15, 25, 31):
            for mem in (1, 8, 9):
                for strategy in (0, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                    for n in (0, 1, 1000, 200000):
                        co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
                        c = co.compress(fastq[:n]) + co.flush()
                        assert Z.decompress(c, wbits) == fastq[:n]
    z = zlib.compress(fastq[:5000])
    g = gzip.compress(fastq[:5000])
    assert Z.decompress(z, 47) == Z.decompress(g, 47) == fastq[:5000]      # auto-detect
    assert Z.decompress(z + b"trailing garbage") == fastq[:5000]


def test_kats_and_errors(Z):
    assert Z.decompress(b'x\x9cK\xcb\xcf\x07\x00\x02\x82\x01E') == b"foo"       # test_zlib_compliance.py:612-614
    assert Z.decompress(b"x\x9cKLJ\x06\x00\x02M\x01\x27") == b"abc"            # :497 (+ the missing Adler byte)
    with pytest.raises(Z.error):
        Z.decompress(b"x\x9cKLJ\x06\x00\x02M\x01")
    assert Z.compress(b"", wbits=-15) == b"\x03\x00"                             # gzip_ng_threaded.py:333 terminator
    with pytest.raises(Z.error) as e:
        Z.decompress(zlib.compress(b"a" * 1000)[:-4])
    e.match("Error -5 while decompressing data: incomplete or truncated stream")  # :249-255
    with pytest.raises(Z.error) as e:
        Z.decompress(b"this is not zlib data")
    e.match("Error -3 while decompressing data")
    with pytest.raises(Z.error) as e:
        Z.compress(b"abc", 42)
    e.match("Bad compression level")
    with pytest.raises(Z.error) as e:
        Z.decompress(zlib.compress(b"x" * 100), 9)
    e.match("invalid window size")                                                # :868
    with pytest.raises(TypeError):
        Z.compress("a string")
    with pytest.raises(ValueError):
        Z.decompress(b"", 15, -1)


def test_parallel_compress_surface(Z, fastq):
    with pytest.raises(Z.error) as e:
        Z._ParallelCompress(1000, 42)
    e.match("Bad compression level")
    pc = Z._ParallelCompress(131072 + 13107, 6)
    with pytest.raises(TypeError):
        pc.compress_and_crc(b"only one")
    blk, zd = fastq[200000:300000], fastq[200000 - 32768:200000]
    out, crc = pc.compress_and_crc(blk, zd)
    assert out.endswith(b"\x00\x00\xff\xff") and crc == zlib.crc32(blk)
    assert zlib.decompressobj(-15, zdict=zd).decompress(out) == blk
    d = zlib.decompressobj(-15, zdict=zd)
    assert d.decompress(out + b"\x03\x00") == blk and d.eof
    small = Z._ParallelCompress(8192 + 819, 3)
    with pytest.raises(OverflowError) as e:
        small.compress_and_crc(os.urandom(65536), b"")
    e.match("Compressed output exceeds buffer size")


def test_gzip_ng_one_shot_and_files(G, Z, fastq, tmp_path):
    data = fastq[:300000]
    c = G.compress(data, 6, mtime=1)
    assert c[4:8] == b"\x01\x00\x00\x00" and c[9] == 255
    assert gzip.decompress(c) == data and G.decompress(c) == data
    assert G.decompress(gzip.compress(data, 1)) == data
    assert G.decompress(gzip.compress(data) + gzip.compress(data[:100]) + bytes(50)) == data + data[:100]
    for name in ("test.fastq.gz", "concatenated.fastq.gz", "test.fastq.bgzip.gz"):
        raw = open(os.path.join(GOLDEN, name), "rb").read()
        assert G.decompress(raw) == gzip.decompress(raw)
    # FEXTRA known answer, tests/test_gzip_compliance.py:622-628
    gz_extra = (b'\x1f\x8b\x08\x04\xb2\x17cQ\x02\xff\x09\x00XX\x05\x00Extra\x0bI-.\x01\x002\xd1Mx\x04\x00\x00\x00')
    assert G.decompress(gz_extra) == b"Test"
    # file objects
    p = tmp_path / "f.gz"
    with G.open(p, "wb", compresslevel=6) as f:
        for i in range(0, len(data), 70000):
            f.write(data[i:i + 70000])
    assert gzip.open(p).read() == data
    with G.open(p, "rb") as f:
        assert f.read() == data
    with G.open(p, "rb") as f:
        f.seek(1000)
        assert f.read(50) == data[1000:1050]
    with G.open(p, "rt") as f:
        assert f.readline() == data.split(b"\n")[0].decode() + "\n"


def test_gzip_reader_errors(G, Z, fastq):
    raw = gzip.compress(fastq[:30000], mtime=0)
    with pytest.raises(G.BadGzipFile) as e:
        G.decompress(raw[:-8] + bytes([raw[-8] ^ 1]) + raw[-7:])
    e.match("CRC check failed")
    with pytest.raises(G.BadGzipFile) as e:
        G.decompress(raw[:-4] + bytes([raw[-4] ^ 1]) + raw[-3:])
    e.match("Incorrect length of data produced")
    with pytest.raises(G.BadGzipFile) as e:
        G.decompress(b"\x1f\x8c" + raw[2:])
    e.match("Not a gzipped file")
    with pytest.raises(G.BadGzipFile) as e:
        G.decompress(raw[:2] + b"\x07" + raw[3:])
    e.match("Unknown compression method")
    with pytest.raises(EOFError) as e:
        G.decompress(raw[:-3])
    e.match("Compressed file ended before the end-of-stream marker was reached")
    with pytest.raises(G.BadGzipFile) as e:
        G.decompress(raw + b"garbage, more than a header")
    e.match("Not a gzipped file \\(b'ga'\\)")
    with pytest.raises(EOFError):                       # fewer than 10 stray bytes read as a truncated header (:2452)
        G.decompress(raw + b"garbage!")
    r = Z._GzipReader(raw + b"garbage, more than a header")
    assert r.read(30000) == fastq[:30000]          # good member is served, the error comes after it
    with pytest.raises(G.BadGzipFile):
        r.read(10)


# ---- incremental objects, after tests/test_zlib_compliance.py / test_compat.py:118-189 --------
def test_decompressobj_streaming(Z, fastq):
    data = fastq[:700000]
    for wbits, blob in ((15, zlib.compress(data, 6)), (-15, zlib.compressobj(9, 8, -15).compress(data) + zlib.compressobj(9, 8, -15).flush() if False else None),
                        (31, gzip.compress(data, 6)), (47, zlib.compress(data, 1))):
        if blob is None:
            co = zlib.compressobj(9, zlib.DEFLATED, -15)
            blob = co.compress(data) + co.flush()
        for step in (1000, 65536, len(blob)):
            d = Z.decompressobj(wbits)
            out = b"".join(d.decompress(blob[i:i + step]) for i in range(0, len(blob), step)) + d.flush()
            assert out == data and d.eof and d.unused_data == b"" and d.unconsumed_tail == b""
    # unused_data after the end of the stream (test_compat.py:179-189)
    d = Z.decompressobj()
    assert d.decompress(zlib.compress(b"abcdef" * 100) + b"trailing") == b"abcdef" * 100
    assert d.eof and d.unused_data == b"trailing"
    assert d.decompress(b"more") == b"" and d.unused_data == b"trailingmore"
    # max_length / unconsumed_tail loop (test_zlib_compliance.py:470-500)
    blob = zlib.compress(data[:200000], 6)
    d = Z.decompressobj()
    chunks = [d.decompress(blob, 1000)]
    assert len(chunks[0]) == 1000 and d.unconsumed_tail
    while d.unconsumed_tail:
        chunks.append(d.decompress(d.unconsumed_tail, 50000))
        assert len(chunks[-1]) <= 50000
    chunks.append(d.flush())
    assert b"".join(chunks) == data[:200000] and d.eof
    cdata = b"x\x9cKLJ\x06\x00\x02M\x01"            # "abc" without the last Adler byte, :497
    d = Z.decompressobj()
    got = d.decompress(cdata, 1)
    got += d.decompress(d.unconsumed_tail)
    assert got == b"abc" and d.unconsumed_tail == b""
    with pytest.raises(ValueError):
        Z.decompressobj().decompress(b"", -1)
    # sync-flushed pieces decode as they arrive (test_zlib_compliance.py:503-533)
    co = zlib.compressobj(6)
    p1 = co.compress(data[:5000]) + co.flush(zlib.Z_SYNC_FLUSH)
    p2 = co.compress(data[5000:9000]) + co.flush()
    d = Z.decompressobj()
    assert d.decompress(p1) == data[:5000] and not d.eof
    assert d.decompress(p2) == data[5000:9000] and d.eof
    # copy (test_zlib_compliance.py:750-779)
    d = Z.decompressobj()
    first = d.decompress(blob[:3000])
    d2 = d.copy()
    assert first + d.decompress(blob[3000:]) + d.flush() == data[:200000]
    assert first + d2.decompress(blob[3000:]) + d2.flush() == data[:200000]
    # zdict
    zd = data[:20000]
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, 0, zd)
    zblob = co.compress(data[10000:60000]) + co.flush()
    assert Z.decompressobj(15, zd).decompress(zblob) == data[10000:60000]
    with pytest.raises(Z.error):
        Z.decompressobj().decompress(zblob)
    with pytest.raises(Z.error):
        Z.decompressobj().decompress(b"not a zlib stream at all")


def test_zlib_decompressor(Z, fastq):
    data = fastq[:300000]
    blob = zlib.compress(data, 6) + b"tail"
    d = Z._ZlibDecompressor()
    out = []
    assert d.needs_input and not d.eof
    feed = [blob[:1000], blob[1000:]]
    for _ in range(100):
        if d.eof:
            break
        chunk = feed.pop(0) if (d.needs_input and feed) else b""
        piece = d.decompress(chunk, 70000)
        assert len(piece) <= 70000
        out.append(piece)
    assert b"".join(out) == data and d.eof and d.unused_data == b"tail"
    with pytest.raises(EOFError):
        d.decompress(b"x")
    d = Z._ZlibDecompressor(wbits=31)
    assert d.decompress(gzip.compress(data[:1000])) == data[:1000] and d.eof


def test_compressobj_and_cli(Z, G, fastq, tmp_path):
    data = fastq[:400000]
    for wbits in (15, -15, 31):
        co = Z.compressobj(6, Z.DEFLATED, wbits)
        blob = b"".join(co.compress(data[i:i + 50000]) for i in range(0, len(data), 50000))
        blob += co.flush(Z.Z_SYNC_FLUSH) + co.compress(b"tail") + co.flush()
        assert zlib.decompress(blob, wbits) == data + b"tail"
    src = tmp_path / "in.txt"
    src.write_bytes(data)
    G.main(["-6", "-n", "-f", str(src)])
    assert gzip.open(str(src) + ".gz").read() == data
    out = tmp_path / "back.txt"
    G.main(["-d", "-f", "-o", str(out), str(src) + ".gz"])
    assert out.read_bytes() == data


def test_large_one_shot_streams_decode_chunk_parallel():
    """zlib_ng.decompress of a large zlib / raw / gzip stream: the engine decodes it chunk-parallel, learns the
    output size from its count pass and asks for exactly that much room (no geometric regrowth)."""
    import zlib
    from zlib_ng_amd import _lib, corpus, zlib_ng
    ctx = _lib.default_context()
    data = corpus.text(24 << 20, seed=12).tobytes()
    for wbits, blob in ((15, zlib.compress(data, 6)), (-15, zlib.compressobj(9, zlib.DEFLATED, -15)),
                        (31, None), (15, zlib_ng.compress(data, 6))):
        if wbits == -15:
            blob = blob.compress(data) + blob.flush()
        if wbits == 31:
            co = zlib.compressobj(6, zlib.DEFLATED, 31); blob = co.compress(data) + co.flush()
        ctx.decode_paths(True)
        assert zlib_ng.decompress(blob, wbits) == data
        paths = ctx.decode_paths(True)
        assert paths["chunked"] >= 1, paths
    # an explicit buffer that is too small: needed size comes back, nothing is copied
    raw = zlib.compress(data, 6)[2:-4]
    code, out, used, _, _ = ctx.inflate_raw(raw, 1 << 20)
    assert code == _lib.BUF_ERROR and out == b"" and ctx.last_needed == len(data)
    code, out, used, crc, ad = ctx.inflate_raw(raw, len(data))
    assert code == _lib.STREAM_END and out == data and used == len(raw)
    assert crc == zlib.crc32(data) and ad == zlib.adler32(data)


def test_compress_copy():
    """After the reference's test_compresscopy / test_badcompresscopy (tests/test_zlib_compliance.py:700-740)."""
    import copy
    import zlib
    from zlib_ng_amd import zlib_ng
    data0 = b"To be, or not to be, that is the question:\n" * 4000
    data1 = data0.swapcase()
    for func in (lambda c: c.copy(), copy.copy, copy.deepcopy):
        c0 = zlib_ng.compressobj(zlib_ng.Z_BEST_COMPRESSION)
        bufs0 = [c0.compress(data0)]
        c1 = func(c0)
        bufs1 = bufs0[:]
        bufs0 += [c0.compress(data0), c0.flush()]
        bufs1 += [c1.compress(data1), c1.flush()]
        assert zlib.decompress(b"".join(bufs0)) == data0 + data0
        assert zlib.decompress(b"".join(bufs1)) == data0 + data1
    # a copy taken after a sync flush continues the same stream with its own dictionary tail
    c0 = zlib_ng.compressobj(6, zlib_ng.DEFLATED, 31)
    head = c0.compress(data0) + c0.flush(zlib_ng.Z_SYNC_FLUSH)
    c1 = c0.copy()
    a = head + c0.compress(data1) + c0.flush()
    b = head + c1.compress(data0) + c1.flush()
    assert zlib.decompress(a, 31) == data0 + data1 and zlib.decompress(b, 31) == data0 + data0
    c = zlib_ng.compressobj()
    c.compress(data0); c.flush()
    with pytest.raises(ValueError):
        c.copy()


def test_gzip_reader_streams_in_windows(monkeypatch, fastq):
    """_GzipReader reads the compressed file in windows (bounded memory): members complete inside a window are
    decoded, the incomplete tail is carried over; a member larger than the window makes the window grow.  Small
    windows force every case; results are the stdlib's."""
    import gzip
    import io
    from conftest import GOLDEN
    from zlib_ng_amd import _lib, corpus, gzip_ng, zlib_ng
    ctx = _lib.default_context()
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(128 << 10))
    text = corpus.text(6 << 20, seed=21).tobytes()
    blobs = {
        "bgzf fixture": open(os.path.join(GOLDEN, "test.fastq.bgzip.gz"), "rb").read(),
        "two members": open(os.path.join(GOLDEN, "concatenated.fastq.gz"), "rb").read(),
        "indexed members": ctx.gzip_members(text, 131072, 6),
        "one big member": gzip.compress(text, 6),
        "mixed": gzip.compress(text[:300000], 1) + bytes(7) + ctx.gzip_members(text[:1 << 20], 65536, 6) + gzip.compress(text[:70000], 9),
    }
    for name, blob in blobs.items():
        want = gzip.decompress(blob)
        r = zlib_ng._GzipReader(io.BytesIO(blob))
        got = bytearray()
        while True:
            piece = r.read(1000003)
            if not piece:
                break
            got += piece
        assert bytes(got) == want, name
        assert r.tell() == len(want)
        # seeking: backwards re-decodes from the start, whence=2 runs to the end
        assert r.seek(12345) == 12345 and r.read(100) == want[12345:12445]
        assert r.seek(-50, 2) == len(want) - 50 and r.read() == want[-50:]
        assert r.seek(len(want) // 2) == len(want) // 2 and r.read(64) == want[len(want) // 2:len(want) // 2 + 64]
        assert zlib_ng._GzipReader(io.BytesIO(blob)).readall() == want
        with gzip_ng.open(io.BytesIO(blob), "rb") as f:
            assert f.read() == want
    # corruption in a late member: everything before it is served, then the error
    blob = bytearray(blobs["indexed members"])
    blob[len(blob) - 20000] ^= 0x10
    r = zlib_ng._GzipReader(io.BytesIO(bytes(blob)))
    got = bytearray()
    with pytest.raises((gzip.BadGzipFile, zlib_ng.error, EOFError)):
        while True:
            piece = r.read(1 << 20)
            if not piece:
                break
            got += piece
    assert len(got) >= len(text) - (256 << 10) and bytes(got) == text[:len(got)]
    # truncated file: EOFError after the good bytes
    r = zlib_ng._GzipReader(io.BytesIO(blobs["one big member"][:-5000]))
    with pytest.raises(EOFError):
        r.readall()


def test_gzip_reader_continues_a_giant_member_across_windows(monkeypatch):
    """One ordinary gzip member much larger than the read window: its complete deflate blocks are decoded window by
    window (bit offset of the next block header + 32 KiB of history + CRC carried in zngamd_gz_state); the window does
    not grow to the size of the member."""
    import gzip
    import io
    from zlib_ng_amd import _lib, corpus, zlib_ng
    ctx = _lib.default_context()
    text = corpus.text(40 << 20, seed=33).tobytes()
    mixed = corpus.mixed(24 << 20, seed=34).tobytes()
    for data, level, window in ((text, 6, 1 << 20), (text, 1, 3 << 20), (mixed, 9, 1 << 20), (text[:3 << 20], 6, 128 << 10)):
        monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(window))
        blob = gzip.compress(data, level) + bytes(5) + gzip.compress(b"tail member", 6)
        want = data + b"tail member"
        ctx.decode_paths(True)
        r = zlib_ng._GzipReader(io.BytesIO(blob))
        got = bytearray()
        while True:
            piece = r.read(4 << 20)
            if not piece:
                break
            got += piece
        assert bytes(got) == want
        assert r._window <= 4 * window, (r._window, window)          # bounded: the member is several times larger
        assert r.seek(len(data) - 10) == len(data) - 10
        tail = b""
        while len(tail) < 30:                                        # a raw reader may return short at a member end
            piece = r.read(30 - len(tail))
            if not piece:
                break
            tail += piece
        assert tail == want[len(data) - 10:len(data) + 20]
    # damage in the middle of the giant member: the bytes before it arrive, then the error
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(1 << 20))
    blob = bytearray(gzip.compress(text, 6))
    blob[len(blob) // 2] ^= 0x55
    r = zlib_ng._GzipReader(io.BytesIO(bytes(blob)))
    got = bytearray()
    with pytest.raises((gzip.BadGzipFile, zlib_ng.error, EOFError)):
        while True:
            piece = r.read(4 << 20)
            if not piece:
                break
            got += piece
    # (like the streaming reference, bytes decoded from damaged data may be handed out before the trailer check fails)
    q = len(text) // 4
    assert len(got) > q and bytes(got[:q]) == text[:q]
    # cut short: EOFError at the end of what could be decoded
    r = zlib_ng._GzipReader(io.BytesIO(gzip.compress(text, 6)[:-100000]))
    got = bytearray()
    with pytest.raises(EOFError):
        while True:
            piece = r.read(4 << 20)
            if not piece:
                break
            got += piece
    assert len(got) > len(text) // 2 and bytes(got) == text[:len(got)]


def test_compressobj_zlib_container_with_preset_dictionary():
    """zdict with the default (zlib) container: FDICT + DICTID in the header, the data primed with the dictionary
    (test_zlib_compliance.py test_dictionary / test_dictionary_streaming)."""
    import zlib
    from zlib_ng_amd import zlib_ng
    words = b"the quick brown fox jumps over the lazy dog and runs away with the spoon ".split()
    zdict = b" ".join(words * 40)
    data = b" ".join(reversed(words * 300))
    co = zlib_ng.compressobj(6, zdict=zdict)
    blob = co.compress(data[:5000]) + co.compress(data[5000:]) + co.flush()
    assert blob[1] & 0x20 and int.from_bytes(blob[2:6], "big") == zlib.adler32(zdict)
    assert zlib.decompressobj(zdict=zdict).decompress(blob) == data
    with pytest.raises(zlib.error):
        zlib.decompress(blob)                                     # needs the dictionary
    do = zlib_ng.decompressobj(zdict=zdict)
    assert do.decompress(blob) + do.flush() == data
    # the dictionary pays: smaller than without it
    plain = zlib_ng.compress(data, 6)
    assert len(blob) < len(plain)
    # and the system's dictionary streams decode here
    c2 = zlib.compressobj(9, zdict=zdict)
    theirs = c2.compress(data) + c2.flush()
    do = zlib_ng.decompressobj(zdict=zdict)
    assert do.decompress(theirs) == data


def test_compressobj_respects_small_windows():
    """compressobj(wbits=9..14 / raw / gzip forms): no match may reach further back than the declared window -- the
    system zlib opened with the same wbits must decode it (test_zlib_compliance.py test_wbits)."""
    import zlib
    from zlib_ng_amd import corpus, zlib_ng
    data = corpus.text(3 << 20, seed=3).tobytes()
    for wb in (9, 10, 12, 14, 15, -9, -12, -15, 25, 28, 31):
        co = zlib_ng.compressobj(6, zlib_ng.DEFLATED, wb)
        blob = co.compress(data[:1000]) + co.compress(data[1000:]) + co.flush()
        assert zlib.decompress(blob, wb) == data, wb
        assert zlib.decompress(zlib_ng.compress(data, 6, wb), wb) == data, wb
        if 9 <= wb <= 15:
            assert (blob[0] >> 4) + 8 == wb                       # CINFO says what was used


def test_full_flush_is_a_restart_point():
    """Compress.flush(Z_FULL_FLUSH) (zlib_ngmodule.c:718-784 -> zng_deflate(Z_FULL_FLUSH)): decompression can restart behind
    the flush point, so the bytes behind it must decode with a FRESH raw decompressor that has seen nothing before -- no match
    may reach back across the point.  Z_SYNC_FLUSH gives no such promise (and with this repetitive input does reach back)."""
    import zlib
    from zlib_ng_amd import corpus, zlib_ng
    a = corpus.text(200000, seed=11).tobytes()
    b = a[-30000:] + corpus.text(100000, seed=12).tobytes()        # starts with what the history holds: tempting matches
    for wb in (-15, 15, 31):
        co = zlib_ng.compressobj(6, zlib_ng.DEFLATED, wb)
        first = co.compress(a) + co.flush(zlib_ng.Z_FULL_FLUSH)
        second = co.compress(b) + co.flush()
        assert zlib.decompress(first + second, wb) == a + b, wb
        assert first.endswith(b"\x00\x00\xff\xff")
        tail_len = {-15: 0, 15: 4, 31: 8}[wb]
        raw_second = second[:len(second) - tail_len] if tail_len else second
        fresh = zlib.decompressobj(-15)
        assert fresh.decompress(raw_second) == b, wb              # a fresh decoder: nothing in front of the flush point is needed
        ours = zlib_ng.decompressobj(-15)
        assert ours.decompress(raw_second) + ours.flush() == b, wb
    # several full flushes in a row, small pieces (the collected-input path)
    co = zlib_ng.compressobj(6, zlib_ng.DEFLATED, -15)
    pieces, blobs = [a[i * 5000:(i + 1) * 5000] for i in range(8)], []
    for p in pieces:
        blobs.append(co.compress(p) + co.flush(zlib_ng.Z_FULL_FLUSH))
    blobs.append(co.flush())
    for p, blob in zip(pieces, blobs):
        assert zlib.decompressobj(-15).decompress(blob) == p
    # the sync flush keeps the history (the contrast that shows the test can fail)
    co = zlib_ng.compressobj(6, zlib_ng.DEFLATED, -15)
    first = co.compress(a) + co.flush(zlib_ng.Z_SYNC_FLUSH)
    second = co.compress(b) + co.flush()
    assert zlib.decompress(first + second, -15) == a + b
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(second)


def test_gzip_compressobj_rejects_a_preset_dictionary():
    """zng_deflateSetDictionary returns Z_STREAM_ERROR on a gzip stream (no gzip decoder could supply the dictionary); the
    reference raises ValueError("Invalid dictionary") (zlib_ngmodule.c:401-416)."""
    from zlib_ng_amd import zlib_ng
    with pytest.raises(ValueError, match="Invalid dictionary"):
        zlib_ng.compressobj(6, zlib_ng.DEFLATED, 31, zdict=b"abcdefgh" * 100)
    zlib_ng.compressobj(6, zlib_ng.DEFLATED, 15, zdict=b"abcdefgh" * 100)       # zlib and raw containers take one
    zlib_ng.compressobj(6, zlib_ng.DEFLATED, -15, zdict=b"abcdefgh" * 100)


def test_stream_reset_entry_points():
    """zngamd_stream_deflate_reset / _inflate_reset (zng_deflateReset zlib_ngmodule.c:1725, zng_inflateReset :2525, :2715):
    a stream is reused for a second, independent stream; the first one's history and checksums are gone."""
    import ctypes as C
    import zlib
    from zlib_ng_amd import _lib, corpus, zlib_ng
    L = zlib_ng._slib()
    ctx = zlib_ng._ctx()
    a = corpus.text(150000, seed=21).tobytes()
    b = a[-20000:] + corpus.text(50000, seed=22).tobytes()

    def run(fn, zst, data, flush):
        out = bytearray()
        buf = C.create_string_buffer(1 << 20)
        keep = C.create_string_buffer(bytes(data), len(data)) if data else None
        zst.next_in = C.cast(keep, C.c_void_p).value if data else None
        zst.avail_in = len(data)
        while True:
            zst.next_out = C.cast(buf, C.c_void_p).value
            zst.avail_out = len(buf)
            err = fn(C.byref(zst), flush)
            out += buf.raw[:len(buf) - zst.avail_out]
            assert err in (_lib.OK, _lib.STREAM_END, _lib.BUF_ERROR), (err, zst.msg)
            if err == _lib.STREAM_END or (zst.avail_in == 0 and zst.avail_out != 0):
                return bytes(out), err
    zst = zlib_ng._ZStream()
    assert L.zngamd_stream_deflate_init(ctx.h, C.byref(zst), 6, 8, 31, 8, 0) == _lib.OK
    first, err = run(L.zngamd_stream_deflate, zst, a, 4)
    assert err == _lib.STREAM_END and zlib.decompress(first, 31) == a
    assert L.zngamd_stream_deflate_reset(C.byref(zst)) == _lib.OK
    assert zst.total_in == 0 and zst.total_out == 0
    second, err = run(L.zngamd_stream_deflate, zst, b, 4)
    assert err == _lib.STREAM_END and zlib.decompress(second, 31) == b          # a complete stream of its own: header, no reach into `a`
    assert L.zngamd_stream_deflate_end(C.byref(zst)) == _lib.OK
    # inflate: two gzip members through one stream with a reset in between (what GzipReader does between members)
    zst = zlib_ng._ZStream()
    assert L.zngamd_stream_inflate_init(ctx.h, C.byref(zst), 47) == _lib.OK     # auto-detect: the reset must go back to "auto"
    got, err = run(L.zngamd_stream_inflate, zst, first, 2)
    assert err == _lib.STREAM_END and got == a
    assert L.zngamd_stream_inflate_reset(C.byref(zst)) == _lib.OK
    got, err = run(L.zngamd_stream_inflate, zst, zlib.compress(b, 6), 2)        # a zlib stream this time
    assert err == _lib.STREAM_END and got == b
    assert L.zngamd_stream_inflate_end(C.byref(zst)) == _lib.OK
