"""gzip_ng_threaded on the GPU engine.  Cases follow the reference's tests/test_gzip_ng_threaded.py
(read == stdlib, threads in {1,3,-1} with 8 KiB blocks, incompressible data, injected oversized block,
bad level, threads=0, append, caller's stream stays open, flush makes a complete file each time)."""
import gzip
import io
import itertools
import os
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TEST_FILE = os.path.join(GOLDEN, "test.fastq.gz")


@pytest.fixture(scope="module")
def T():
    from zlib_ng_amd import gzip_ng_threaded
    return gzip_ng_threaded


@pytest.fixture(scope="module")
def Z():
    from zlib_ng_amd import zlib_ng
    return zlib_ng


def test_threaded_read(T):
    with T.open(TEST_FILE, "rb") as f:
        thread_data = f.read()
    with gzip.open(TEST_FILE, "rb") as f:
        assert thread_data == f.read()


@pytest.mark.parametrize(["mode", "threads"], itertools.product(["wb", "wt"], [1, 3, -1]))
def test_threaded_write(T, mode, threads):
    with tempfile.NamedTemporaryFile("wb", delete=False) as tmp:
        with T.open(tmp, mode, threads=threads, block_size=8 * 1024) as out_file:
            with gzip.open(TEST_FILE, "rb" if "b" in mode else "rt") as in_file:
                while True:
                    block = in_file.read(128 * 1024)
                    if not block:
                        break
                    out_file.write(block)
    with gzip.open(TEST_FILE, "rt") as a, gzip.open(tmp.name, "rt") as b:
        assert a.read() == b.read()
    os.unlink(tmp.name)


def test_threaded_write_framing_matches_reference_layout(T, fastq):
    """header 1f8b0800 00000000 ff xfl | blocks | 03 00 | crc isize | trailing empty member;
    byte-identical for 1 and 3 threads."""
    import struct
    import zlib
    data = fastq[:512 * 1024]
    outs = []
    for threads in (1, 3):
        bio = io.BytesIO()
        with T.open(bio, "wb", compresslevel=6, threads=threads, block_size=128 * 1024, exact_framing=True) as f:
            f.write(data)
        outs.append(bio.getvalue())
    assert outs[0] == outs[1]
    blob = outs[0]
    assert blob[:10] == bytes.fromhex("1f8b0800" "00000000" "ff00")
    empty = bytes.fromhex("1f8b0800" "00000000" "ff00" "0300" "00000000" "00000000")
    assert blob.endswith(empty)
    first = blob[:-len(empty)]
    assert first[-10:-8] == b"\x03\x00"
    assert struct.unpack("<II", first[-8:]) == (zlib.crc32(data), len(data))
    assert gzip.decompress(blob) == data


def test_threaded_open_no_threads(T):
    with tempfile.TemporaryFile("rb") as tmp:
        klass = T.open(tmp, "rb", threads=0)
        assert isinstance(klass, gzip.GzipFile)


def test_threaded_open_not_a_file_or_pathlike(T):
    with pytest.raises(TypeError) as error:
        T.open((1, 2, 3))
    error.match("str")
    error.match("bytes")
    error.match("file")


@pytest.mark.timeout(30)
def test_threaded_read_error(T):
    data = open(TEST_FILE, "rb").read()
    with T.open(io.BytesIO(data[:-8]), "rb") as tr_f:
        with pytest.raises(EOFError):
            tr_f.read()


@pytest.mark.timeout(30)
@pytest.mark.parametrize("threads", [1, 3])
def test_threaded_write_oversized_block_no_error(T, threads):
    data = os.urandom(1024 * 63)
    with tempfile.NamedTemporaryFile(mode="wb", delete=False) as tmp:
        with T.open(tmp, "wb", compresslevel=3, threads=threads, block_size=8 * 1024) as writer:
            writer.write(data)
    with gzip.open(tmp.name, "rb") as gzipped:
        assert data == gzipped.read()
    os.unlink(tmp.name)


@pytest.mark.timeout(30)
@pytest.mark.parametrize("threads", [1, 3])
def test_threaded_write_error(T, threads):
    f = T._ThreadedGzipWriter(io.BytesIO(), level=3, threads=threads, block_size=8 * 1024)
    f.input_queues[0].put((os.urandom(1024 * 64), b""))
    with pytest.raises(OverflowError) as error:
        f.close()
    error.match("Compressed output exceeds buffer size")


def test_close_reader_and_writer(T):
    f = T._ThreadedGzipReader(io.BytesIO(open(TEST_FILE, "rb").read()), "rb")
    f.close()
    assert f.closed
    f.close()
    for threads in (1, 3):
        w = T._ThreadedGzipWriter(io.BytesIO(), threads=threads)
        w.close()
        assert w.closed
        w.close()
        with pytest.raises(ValueError) as error:
            w.write(b"abc")
        error.match("closed")


def test_readable_writable(T):
    with T.open(TEST_FILE, "rb") as f:
        assert not f.writable()
    with T.open(io.BytesIO(), "wb") as f:
        assert not f.readable()


def test_writer_wrong_level(T, Z):
    with tempfile.NamedTemporaryFile("wb") as tmp:
        with pytest.raises(Z.error) as error:
            T.open(tmp.name, mode="wb", compresslevel=42)
        error.match("Bad compression level")


def test_writer_too_low_threads(T):
    with pytest.raises(ValueError) as error:
        T._ThreadedGzipWriter(io.BytesIO(), threads=0)
    error.match("threads")
    error.match("at least 1")


def test_reader_read_after_close(T):
    with open(TEST_FILE, "rb") as test_f:
        f = T._ThreadedGzipReader(test_f)
        f.close()
        with pytest.raises(ValueError) as error:
            f.read(1024)
        error.match("closed")


def test_append_binary_and_text(T, tmp_path):
    p = tmp_path / "test.txt.gz"
    with T.open(p, "wb") as f:
        f.write(b"AB")
    with T.open(p, mode="ab") as f:
        f.write(b"CD")
    assert gzip.open(p, "rb").read() == b"ABCD"
    q = tmp_path / "t2.txt.gz"
    with T.open(q, "wt") as f:
        f.write("AB")
    with T.open(q, mode="at") as f:
        f.write("CD")
    assert gzip.open(q, "rt").read() == "ABCD"


def test_streams_not_closed(T):
    s = io.BytesIO(gzip.compress(b"thisisatest"))
    with T.open(s, "rb") as f:
        assert f.read() == b"thisisatest"
    assert not s.closed
    s = io.BytesIO()
    with T.open(s, "wb") as f:
        f.write(b"thisisatest")
    assert not s.closed
    assert gzip.decompress(s.getvalue()) == b"thisisatest"


@pytest.mark.parametrize("threads", [1, 2])
def test_flush(T, tmp_path, threads):
    p = tmp_path / "output.gz"
    with T.open(p, "wb", threads=threads) as f:
        f.write(b"1")
        f.flush()
        assert gzip.decompress(p.read_bytes()) == b"1"
        f.write(b"2")
        f.flush()
        assert gzip.decompress(p.read_bytes()) == b"12"
        f.write(b"3")
        f.flush()
        assert gzip.decompress(p.read_bytes()) == b"123"
    assert gzip.decompress(p.read_bytes()) == b"123"


def test_large_blocks_default_size(T, fastq):
    """Default block_size is 1 MiB (gzip_ng_threaded.py:23-24): blocks are cut into 128 KiB units inside."""
    bio = io.BytesIO()
    with T.open(bio, "wb", threads=4) as f:
        f.write(fastq)
    assert gzip.decompress(bio.getvalue()) == fastq


@pytest.mark.parametrize("threads", [1, 3])
def test_writer_write_after_close(T, threads):
    # reference tests/test_gzip_ng_threaded.py:166-172
    f = T._ThreadedGzipWriter(io.BytesIO(), threads=threads)
    f.close()
    with pytest.raises(ValueError, match="closed"):
        f.write(b"abc")


@pytest.mark.parametrize("mode,threads", list(itertools.product(["rb", "wb"], [1, 2])))
def test_program_exits_when_it_fails_with_an_open_file(T, tmp_path, mode, threads):
    """A script that opens a threaded file without a context manager and then raises must still terminate: the worker
    threads may not keep the interpreter alive (reference tests/test_gzip_ng_threaded.py:216-232)."""
    import subprocess
    import sys
    from conftest import PKG_DIR
    target = tmp_path / "output.gz"
    target.write_bytes(gzip.compress(b"test" * (10 * 1024 * 1024)))
    program = tmp_path / "no_context_manager.py"
    program.write_text(
        "import sys\n"
        f"sys.path.insert(0, {PKG_DIR!r})\n"
        "from zlib_ng_amd import gzip_ng_threaded\n"
        f"f = gzip_ng_threaded.open({str(target)!r}, mode={mode!r}, threads={threads})\n"
        + ("f.read(100)\n" if mode == "rb" else "f.write(b'x' * 3000000)\n") +
        "raise Exception('Error')\n")
    done = subprocess.run([sys.executable, str(program)], capture_output=True, timeout=120)
    assert done.returncode == 1 and b"Exception: Error" in done.stderr


def test_threaded_reader_overlaps_windows(T, monkeypatch):
    """Many read windows: the pump hands whole decoded windows to the consumer and gets their buffers back (three are in
    rotation); odd read sizes cross window borders; a truncated file still raises after the good bytes."""
    from zlib_ng_amd import corpus
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(1 << 20))
    data = corpus.text(20 << 20, seed=5).tobytes() + corpus.mixed(6 << 20, seed=6).tobytes()
    blob = gzip.compress(data, 1) + gzip.compress(data[:3 << 20], 6)
    want = data + data[:3 << 20]
    for piece in (1 << 20, 777_777, 5 << 20):
        got = bytearray()
        with T.open(io.BytesIO(blob), "rb") as f:
            while True:
                b = f.read(piece)
                if not b:
                    break
                got += b
        assert bytes(got) == want, piece
    got = bytearray()
    with pytest.raises(EOFError):
        with T.open(io.BytesIO(blob[:-5]), "rb") as f:
            while True:
                b = f.read(1 << 20)
                if not b:
                    break
                got += b
    # (io.BufferedReader drops what a read() call had gathered when the raw reader raises inside it: up to one request)
    assert len(got) >= len(data) - (1 << 20) and want.startswith(bytes(got))


def test_writer_spreads_batches_over_contexts(monkeypatch):
    """threads > 1: every batch is cut into contiguous block ranges, one per GPU context (here two contexts on the one GPU of the
    test box), each range primed by the input in front of it -- the file is byte for byte what one context writes, and the
    stdlib reads it (reference fan-out / in-order drain: gzip_ng_threaded.py:233-246, :316-321, :382-398)."""
    import gzip
    import io
    from zlib_ng_amd import _lib, corpus, gzip_ng_threaded
    data = corpus.text(40 << 20, seed=9).tobytes() + b"tail" * 1000
    one = io.BytesIO()
    with gzip_ng_threaded.open(one, "wb", compresslevel=6, threads=1, block_size=128 * 1024) as f:
        f.write(data)
    monkeypatch.setenv("ZNGAMD_DEVICES", "0,0,0")
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.delenv("ZNGAMD_DEVICE", raising=False)
    assert len(_lib.contexts(8)) == 3 and len({id(c) for c in _lib.contexts(8)}) == 3 and len(_lib.contexts(2)) == 2
    many = io.BytesIO()
    with gzip_ng_threaded.open(many, "wb", compresslevel=6, threads=3, block_size=128 * 1024) as f:
        f.write(data)
        f.write(data[:3 << 20])
    ref = io.BytesIO()
    with gzip_ng_threaded.open(ref, "wb", compresslevel=6, threads=1, block_size=128 * 1024) as f:
        f.write(data)
        f.write(data[:3 << 20])
    assert many.getvalue() == ref.getvalue()
    assert gzip.decompress(many.getvalue()) == data + data[:3 << 20]
    assert gzip.decompress(one.getvalue()) == data
    # the ranges of one batch: contiguous, in order, each with its dictionary in front
    ctxs = _lib.contexts(3)
    blocks = [(o, min(131072, len(data) - o), min(32768, o), 0) for o in range(0, len(data), 131072)]
    a = _lib.deflate_blocks_multi(ctxs, data, blocks, 6, 131072 + 13107 + 500)
    b = ctxs[0].deflate_blocks(data, blocks, 6, 131072 + 13107 + 500, joined=True)
    assert a[0] == b[0] and list(a[1]) == list(b[1]) and a[2] == b[2] and list(a[3]) == list(b[3])


def test_writer_pieces_of_every_size_class(T):
    """What write() does with a piece depends on its size: below 32 KiB it is copied under the interpreter lock, from there on
    (every eighth one, while a batch runs) without it, a piece larger than the current batch limit ends the batch and raises the
    limit, one of 64 MiB or more goes to the engine as it is -- one stream whatever the mix, and the batches (8 MiB, then twice as
    much each time) leave no seam."""
    import zlib
    from zlib_ng_amd import corpus
    rng = np.random.default_rng(7)
    src = corpus.text(96 << 20, seed=3).tobytes()
    sizes = [1, 100, 32767, 32768, 32769, 131072, (8 << 20) + 1, 3 << 20, 20 << 20, 65 << 20] + [int(x) for x in rng.integers(1, 400000, 60)]
    rng.shuffle(sizes)
    out = io.BytesIO()
    pos, pieces = 0, []
    with T.open(out, "wb", compresslevel=4, threads=8, block_size=128 * 1024) as f:
        for sz in sizes:
            sz = min(sz, len(src) - pos)
            if sz <= 0:
                break
            piece = src[pos:pos + sz] if sz % 3 else memoryview(src)[pos:pos + sz]       # bytes and views alike
            assert f.write(piece) == sz
            pos += sz
    d = zlib.decompressobj(31)
    got = d.decompress(out.getvalue())
    assert got == src[:pos] and d.eof
    # behind the data: the empty members that carry the writer's segment index (FEXTRA), then the plain empty member of the
    # reference's close()
    assert d.unused_data[:4] == b"\x1f\x8b\x08\x04" and d.unused_data.endswith(b"\x1f\x8b\x08\x00" + bytes(4) + b"\xff\x00\x03\x00" + bytes(8))


def test_packed_batch_output_equals_the_per_block_form():
    """zngamd_deflate_blocks_packed (what the writer's batches use: the blocks' outputs back to back, copied straight into the
    caller's buffer) against zngamd_deflate_blocks (a slot per block): same bytes, lengths and CRCs; into a buffer of the
    caller's, with a block table made once; an overflowing block is reported the same way."""
    import zlib
    from zlib_ng_amd import _lib, corpus
    ctx = _lib.default_context()
    data = corpus.mixed(5 << 20, seed=4).tobytes()
    bs = 96 * 1024 + 7
    blocks = [(o, min(bs, len(data) - o), min(32768, o), 0) for o in range(0, len(data), bs)]
    cap = bs + bs // 10 + 500
    per_block, crcs_a, over_a = ctx.deflate_blocks(data, blocks, 5, cap)
    packed, crcs_b, over_b, lens = ctx.deflate_blocks(data, blocks, 5, cap, joined=True)
    assert not over_a and not over_b and crcs_a == crcs_b
    assert packed == b"".join(per_block) and lens == [len(x) for x in per_block]
    into = bytearray(len(blocks) * cap)
    view, crcs_c, over_c, lens_c = ctx.deflate_blocks(data, _lib.block_table(blocks), 5, cap, joined=True, into=into)
    assert bytes(view) == packed and crcs_c == crcs_a and lens_c == lens and not over_c
    assert zlib.decompress(packed + b"\x03\x00", -15) == data
    fold = 0
    for c, (_, ln, _, _) in zip(crcs_a, blocks):
        fold = ctx.crc32_combine(fold, c, ln)
    assert _lib.crc32_combine_many(0, crcs_a, [b[1] for b in blocks]) == fold == zlib.crc32(data)
    noise = np.random.default_rng(3).bytes(1 << 20)
    nb = [(o, 65536, 0, 0) for o in range(0, len(noise), 65536)]
    res, _, over, _ = ctx.deflate_blocks(noise, nb, 6, 65536, joined=True)          # incompressible: every block reaches its cap
    assert over and res is None


def test_reader_read_ahead_and_long_tails(T, Z, monkeypatch, tmp_path):
    """The next window's bytes are read from the file while the engine decodes this one, into a second buffer behind room for the
    tail this window leaves over; a tail longer than the room (here the room is made tiny) takes the copying path; seeking
    back waits for a read that is under way before it moves the file; buffers go back to the pool at close."""
    from zlib_ng_amd import _lib, corpus
    data = corpus.text(24 << 20, seed=11).tobytes()
    path = tmp_path / "a.gz"
    path.write_bytes(gzip.compress(data[:9 << 20], 6) + gzip.compress(data[9 << 20:], 1))
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(1 << 20))
    for front in (4 << 20, 4096):
        monkeypatch.setattr(Z._GzipReader, "_AHEAD_FRONT", front)
        with T.open(str(path), "rb") as f:
            assert f.read() == data
        with open(path, "rb") as fh:
            r = Z._GzipReader(fh, 1 << 16)
            assert r.read(1 << 20) == data[:1 << 20]
            r.seek(5 << 20)
            assert r.read(100) == data[5 << 20:(5 << 20) + 100]
            r.seek(10)                                                     # backwards: from the start again, the read-ahead joined first
            assert r.read(1000) == data[10:1010]
            assert r.readall() == data[1010:]
            r.close()
    before = len(_lib._buffer_pool)
    with T.open(str(path), "rb") as f:
        f.read(1 << 20)
    assert len(_lib._buffer_pool) >= min(before, 1)                        # (closed in the middle of the file: nothing leaks, buffers are kept)


def test_gzip_ng_open_pipelined_writer(tmp_path):
    """gzip_ng.open compresses through _PipelinedDeflate: batches of 8, 16, 32, 64 MiB on a thread beside the caller, each primed by the
    32 KiB in front of it -- one deflate stream over one window.  Pieces of every size class, flushes of the three kinds in
    between (each makes everything written so far readable; a full flush also cuts the history), the CRC folded from the
    batches, the stdlib as the judge."""
    import zlib
    from zlib_ng_amd import corpus, gzip_ng, zlib_ng
    src = corpus.text(100 << 20, seed=13).tobytes()
    rng = np.random.default_rng(5)
    path = tmp_path / "p.gz"
    pos = 0
    with gzip_ng.open(str(path), "wb", compresslevel=5) as f:
        assert isinstance(f.compress, gzip_ng._PipelinedDeflate)
        for sz in [1, 40000, 131072, (9 << 20) + 3, 131072, 70 << 20, 5, 131072] + [int(x) for x in rng.integers(1, 300000, 40)]:
            sz = min(sz, len(src) - pos)
            piece = src[pos:pos + sz] if sz % 2 else memoryview(src)[pos:pos + sz]
            assert f.write(piece) == sz
            pos += sz
            if sz == 5:
                f.flush()                                             # Z_SYNC_FLUSH: the file is readable up to here
                d = zlib.decompressobj(31)
                assert d.decompress(path.read_bytes()) == src[:pos]
            if sz == 40000:
                f.flush(zlib_ng.Z_FULL_FLUSH)
    with gzip.open(str(path), "rb") as g:
        assert g.read() == src[:pos]
    raw = path.read_bytes()
    assert int.from_bytes(raw[-8:-4], "little") == zlib.crc32(src[:pos]) and int.from_bytes(raw[-4:], "little") == pos & 0xFFFFFFFF
    # the compressor on its own: nothing comes out before a batch is full, everything at the latest with flush(Z_FINISH)
    c = gzip_ng._PipelinedDeflate(6)
    parts = [c.compress(src[o:o + 65536]) for o in range(0, 40 << 20, 65536)]      # (the 8 MiB batch comes out when the 16 MiB one goes in)
    assert parts[0] == b"" and any(parts)
    parts.append(c.flush())
    assert zlib.decompress(b"".join(parts), -15) == src[:40 << 20] and c._crc == zlib.crc32(src[:40 << 20])
    with pytest.raises(ValueError):
        c.compress(b"more")
    with pytest.raises(zlib_ng.error):
        gzip_ng._PipelinedDeflate(10)


def test_reader_rewinds_while_decoding_ahead(Z, monkeypatch, tmp_path):
    """_GzipReader decodes the window after the current one on a thread of its own; seeking back resets the decoder's state, and that
    thread (and the file read-ahead it started) has to be through before anything is reset or the file is moved: many rewinds at
    odd moments, every byte checked (a stale carry in front of the fresh stream reads as "Not a gzipped file")."""
    from zlib_ng_amd import corpus, gzip_ng
    data = corpus.text(12 << 20, seed=17).tobytes()
    path = tmp_path / "r.gz"
    path.write_bytes(gzip.compress(data[:5 << 20], 6) + gzip.compress(data[5 << 20:], 3))
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(1 << 18))
    rng = np.random.default_rng(11)
    with open(path, "rb") as fh:
        r = Z._GzipReader(fh, 1 << 16)
        assert r._decode_ahead
        for _ in range(40):
            pos = int(rng.integers(0, len(data) - 70000))
            r.seek(pos)
            n = int(rng.integers(1, 70000))
            got = b""
            while len(got) < n:                                  # (a raw reader may hand out less than asked for: the rest of its window)
                piece = r.read(n - len(got))
                assert piece
                got += piece
            assert got == data[pos:pos + n]
        r.seek(0)
        assert r.readall() == data
        r.close()
    with gzip_ng.open(str(path), "rb") as g:                    # the same through the file object, seeks included
        g.seek(3 << 20)
        assert g.read(1000) == data[3 << 20:(3 << 20) + 1000]
        g.seek(100)
        assert g.read(50) == data[100:150]
        assert g.read() == data[150:]


def test_reader_leaves_a_source_that_cannot_be_sought_alone(Z, monkeypatch):
    """Reading and decoding ahead run on threads of their own and ask the source for whole windows: a pipe, a socket or any
    object that cannot be sought gets neither (a consumer that stops early would wait in close() for a window that may never
    come, and the caller's file object would have a second reader) -- and a file that can be sought is decoded ahead only once
    the consumer has come back for a second window."""
    import io
    import threading
    from zlib_ng_amd import corpus

    class Pipe(io.RawIOBase):                                    # reads like a file, cannot be sought
        def __init__(self, b):
            self._b = io.BytesIO(b)

        def readable(self):
            return True

        def seekable(self):
            return False

        def readinto(self, buf):
            return self._b.readinto(buf)

    data = corpus.text(6 << 20, seed=23).tobytes()
    blob = gzip.compress(data, 6)
    monkeypatch.setenv("ZNGAMD_READ_WINDOW", str(1 << 18))
    names = lambda: {t.name for t in threading.enumerate() if t.name.startswith("zng-amd-")}
    r = Z._GzipReader(Pipe(blob), 1 << 16)
    assert not r._bulk_ok
    got = b""
    while len(got) < (3 << 20):
        got += r.read(70000)
        assert not names(), names()
    assert got == data[:len(got)]
    r.close()
    r = Z._GzipReader(io.BytesIO(blob), 1 << 16)
    assert r._bulk_ok
    first = r.read(1000)
    assert first == data[:1000] and "zng-amd-decode-ahead" not in names()        # the first window alone: nothing decoded ahead
    assert first + r.readall() == data
    r.close()
    assert not names()


def test_writer_of_indexed_members(T, tmp_path, monkeypatch):
    """indexed_members=True (or ZNGAMD_WRITER_MEMBERS=1): the writer leaves independent gzip members of at most 128 KiB carrying
    this engine's chunk index -- a gzip file for the standard library, and the members this engine's reader decodes with one
    wavefront each (the decoder's `indexed` path, not the chunk pipeline)."""
    import gzip as std_gzip
    from zlib_ng_amd import _lib, corpus
    data = corpus.text(5 * (1 << 20) + 4321, seed=21).tobytes()
    target = tmp_path / "members.gz"
    with T.open(target, "wb", compresslevel=6, threads=2, block_size=128 * 1024, indexed_members=True) as f:
        for o in range(0, 3 << 20, 100000):                     # small writes, collected
            f.write(data[o:min(o + 100000, 3 << 20)])
        f.flush()                                               # (a flush in the middle ends nothing but the batch)
        f.write(data[3 << 20:])
    raw = target.read_bytes()
    assert raw[:4] == b"\x1f\x8b\x08\x04"                       # FEXTRA: the index
    assert std_gzip.decompress(raw) == data
    n_members = -(-(3 << 20) // (128 * 1024)) + -(-(len(data) - (3 << 20)) // (128 * 1024))     # (no trailing empty member)
    ctx = _lib.default_context()
    ctx.decode_paths(reset=True)
    with T.open(target, "rb", threads=1) as f:
        assert f.read() == data
    counts = ctx.decode_paths(reset=True)
    assert counts["indexed"] == n_members and counts["chunked"] == 0 and counts["sequential"] == 0, counts
    # one big write, the environment switch, and an empty file
    monkeypatch.setenv("ZNGAMD_WRITER_MEMBERS", "1")
    with T.open(target, "wb", threads=1) as f:
        f.write(data)
    assert std_gzip.decompress(target.read_bytes()) == data and target.read_bytes()[3] == 4
    with T.open(target, "wb", threads=1) as f:
        pass
    assert std_gzip.decompress(target.read_bytes()) == b""
    monkeypatch.delenv("ZNGAMD_WRITER_MEMBERS")
    with T.open(target, "wb", threads=1) as f:                  # off: the reference's framing
        f.write(data[:100000])
    assert target.read_bytes()[3] == 0 and std_gzip.decompress(target.read_bytes()) == data[:100000]


def test_indexed_members_followed_by_padding_and_foreign_members(T, tmp_path):
    """A file of indexed members with NUL padding behind it -- and with a member of another writer behind the padding -- decodes:
    the reference's reader skips NUL bytes between and behind members (zlib_ngmodule.c:2604-2612; CPython's gzip does too)."""
    import gzip as std_gzip
    from zlib_ng_amd import corpus, gzip_ng, zlib_ng
    data = corpus.text(400000, seed=43).tobytes()
    target = tmp_path / "pad.gz"
    with T.open(target, "wb", compresslevel=6, threads=1, block_size=128 * 1024, indexed_members=True) as f:
        f.write(data)
    members = target.read_bytes()
    assert members[3] == 4
    tail = b"the tail of another writer\n" * 300
    for raw, want in ((members + bytes(50), data), (members + bytes(7), data),
                      (members + bytes(50) + std_gzip.compress(tail, mtime=0), data + tail),
                      (members + std_gzip.compress(tail, mtime=0) + bytes(13), data + tail)):
        assert std_gzip.decompress(raw) == want
        assert gzip_ng.decompress(raw) == want
        target.write_bytes(raw)
        with gzip_ng.open(target, "rb") as f:
            assert f.read() == want
        with T.open(target, "rb", threads=1) as f:
            assert f.read() == want


def test_gzip_ng_open_writes_members_when_the_environment_asks(tmp_path, monkeypatch):
    """ZNGAMD_WRITER_MEMBERS=1 also reaches gzip_ng.open(..., "w?"): the file is independent indexed members (binary and text
    mode), any gzip reader reads it; without the variable the writer is GzipNGFile as ever."""
    import gzip as std_gzip
    from zlib_ng_amd import corpus, gzip_ng
    data = corpus.text(700000, seed=41).tobytes()
    target = tmp_path / "m.gz"
    monkeypatch.setenv("ZNGAMD_WRITER_MEMBERS", "1")
    with gzip_ng.open(target, "wb", compresslevel=6) as f:
        assert not isinstance(f, gzip_ng.GzipNGFile)
        for o in range(0, len(data), 50000):
            f.write(data[o:o + 50000])
    raw = target.read_bytes()
    assert raw[3] == 4 and std_gzip.decompress(raw) == data
    with gzip_ng.open(target, "rb") as f:
        assert f.read() == data
    with gzip_ng.open(target, "wt", encoding="utf-8") as f:
        f.write("zeile eins\nzeile zwei\n")
    assert std_gzip.decompress(target.read_bytes()) == b"zeile eins\nzeile zwei\n"
    monkeypatch.delenv("ZNGAMD_WRITER_MEMBERS")
    with gzip_ng.open(target, "wb") as f:
        assert isinstance(f, gzip_ng.GzipNGFile)
        f.write(data[:1000])
    assert target.read_bytes()[3] != 4 and std_gzip.decompress(target.read_bytes()) == data[:1000]
