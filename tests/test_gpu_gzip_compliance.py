"""Behavioural compliance of zlib_ng_amd.gzip_ng with CPython's gzip module and with the behaviours the reference
pins for its own gzip_ng (tests/test_gzip_compliance.py, tests/test_gzip_ng.py; class / line given at each test).
Where CPython's gzip has the same entry point the scenario runs against both modules and the observations must
agree; files one module writes are read by the other.  Data and harness are this repo's own."""
import array
import gzip as CG
import io
import os
import pathlib
import random
import struct
import sys
import zlib as CZ

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from zlib_ng_amd import gzip_ng
    return gzip_ng


def _lines(seed, n):
    rnd = random.Random(seed)
    out = bytearray()
    while len(out) < n:
        out += b"  " + bytes(rnd.choice(b"etaoinshrdlu ") for _ in range(rnd.randint(10, 60))) + b"\n"
    return bytes(out[:n])


D1 = _lines(1, 700) + b"\n"
D2 = _lines(2, 1200) + b"\n"


class Unseekable(io.BytesIO):
    """reference: UnseekableIO, tests/test_gzip_compliance.py:50-58"""

    def seekable(self):
        return False

    def tell(self):
        raise io.UnsupportedOperation

    def seek(self, *args):
        raise io.UnsupportedOperation


def outcome(fn, mod):
    try:
        return ("ok", fn(mod))
    except Exception as e:   # noqa: BLE001
        return ("raise", "BadGzipFile" if isinstance(e, mod.BadGzipFile) else type(e).__name__)


def same(fn, G):
    a, b = outcome(fn, CG), outcome(fn, G)
    assert a == b, (repr(a)[:300], repr(b)[:300])
    return b


@pytest.fixture
def path(tmp_path):
    return str(tmp_path / "file.gz")


# ------------------------------------------------------------------------------------------- writing and reading

def test_write_flush_fileno_close_twice(G, path):
    # TestGzip.test_write (reference :85-97)
    with G.GzipFile(path, "wb") as f:
        f.write(D1 * 50)
        f.flush()
        os.fsync(f.fileno())
        f.close()
    f.close()
    assert CG.open(path).read() == D1 * 50


def test_pathlike_and_bytes_names(G, path):
    # test_write_read_with_pathlike_file / test_bytes_filename / test_1647484 / test_paddedfile_getattr (:99-112, :533-545)
    p = pathlib.Path(path)
    with G.GzipFile(p, "w") as f:
        f.write(D1 * 50)
    assert isinstance(f.name, str)
    with G.GzipFile(p, "a") as f:
        f.write(D1)
    with G.GzipFile(p) as f:
        assert f.read() == D1 * 51
    assert isinstance(f.name, str)
    b = path.encode("ascii") + b"2"
    with G.GzipFile(b, "wb") as f:
        f.write(D2)
    with G.GzipFile(b, "rb") as f:
        assert f.read() == D2
    with G.GzipFile(path + "2", "rb") as f:
        assert f.read() == D2 and f.name == path + "2" and f.fileobj.name == path + "2"


@pytest.mark.parametrize("kind", ["memoryview", "shaped", "bytearray", "array_I", "array_Q"])
def test_write_accepts_buffers(G, path, kind):
    # test_write_memoryview / bytearray / array / test_issue44439 (reference :114-124, :630-636)
    data = {"memoryview": memoryview(D1 * 50), "shaped": memoryview(bytes(range(256))).cast("B", shape=[8, 8, 4]),
            "bytearray": bytearray(D1 * 50), "array_I": array.array("I", (D1 * 40)[:2800]),
            "array_Q": array.array("Q", [1, 2, 3, 4, 5])}[kind]
    raw = bytes(data)
    with G.GzipFile(path, "wb") as f:
        assert f.write(data) == len(raw)
        assert f.tell() == len(raw)
    with G.GzipFile(path, "rb") as f:
        assert f.read() == raw
    assert CG.open(path).read() == raw


def test_write_rejects_non_buffers_without_damage(G, path):
    # test_write_incompatible_type (reference :126-135)
    with G.GzipFile(path, "wb") as f:
        for bad in ("", [], 7, None):
            with pytest.raises(TypeError):
                f.write(bad)
        f.write(D1)
    assert CG.open(path).read() == D1


def test_read_and_read1(G, path):
    # test_read / test_read1 (reference :139-158)
    with CG.GzipFile(path, "wb") as f:
        f.write(D1 * 50)
    with G.GzipFile(path, "r") as f:
        assert f.read() == D1 * 50
    blocks, nread = [], 0
    with G.GzipFile(path, "r") as f:
        while True:
            d = f.read1()
            if not d:
                break
            blocks.append(d)
            nread += len(d)
            assert f.tell() == nread
        assert f.read1(100) == b"" and f.read(100) == b""
    assert b"".join(blocks) == D1 * 50
    with G.GzipFile(path, "r") as f:
        assert f.read(2 ** 33) == D1 * 50          # test_read_large (:160-167): sizes beyond UINT_MAX are fine


def test_io_on_closed_objects(G, path):
    # test_io_on_closed_object / test_with_open (reference :169-195, :418-437)
    def run(m):
        seen = []
        with m.GzipFile(path, "wb") as f:
            f.write(b"xxx")
        f = m.GzipFile(path, "r")
        fileobj = f.fileobj
        seen.append(fileobj.closed)
        f.close()
        seen.append(fileobj.closed)
        for op in (lambda: f.read(1), lambda: f.seek(0), lambda: f.tell(), lambda: f.__enter__()):
            try:
                op()
                seen.append("ok")
            except ValueError:
                seen.append("ValueError")
        f = m.GzipFile(path, "w")
        fileobj = f.fileobj
        f.close()
        seen.append(fileobj.closed)
        for op in (lambda: f.write(b""), lambda: f.flush()):
            try:
                op()
                seen.append("ok")
            except ValueError:
                seen.append("ValueError")
        try:
            with m.GzipFile(path, "wb") as f:
                1 / 0
        except ZeroDivisionError:
            seen.append(f.closed)
        return seen
    same(run, G)


def test_append_members(G, path):
    # test_append / test_many_append (reference :197-225)
    with G.GzipFile(path, "wb") as f:
        f.write(D1 * 50)
    with G.GzipFile(path, "ab") as f:
        f.write(D2 * 15)
    for m in (CG, G):
        with m.GzipFile(path, "rb") as f:
            assert f.read() == D1 * 50 + D2 * 15
    with G.GzipFile(path, "wb", 9) as f:
        f.write(b"a")
    for i in range(120):
        with (G if i % 2 else CG).GzipFile(path, "ab", 9) as f:
            f.write(b"a")
    with G.GzipFile(path, "rb") as f:
        contents = b""
        while True:
            z = f.read(8192)
            if not z:
                break
            contents += z
    assert contents == b"a" * 121


@pytest.mark.filterwarnings("ignore:GzipFile was opened for writing:FutureWarning")
def test_exclusive_and_modes(G, path):
    # test_exclusive_write / test_mode / test_fileobj_mode (reference :227-233, :309-315, :508-531)
    with G.GzipFile(path, "xb") as f:
        f.write(D1 * 50)
    with G.GzipFile(path, "rb") as f:
        assert f.read() == D1 * 50 and f.myfileobj.mode == "rb"
    with pytest.raises(FileExistsError):
        G.GzipFile(path, "xb")
    os.unlink(path)
    with G.GzipFile(path, "x") as f:
        assert f.myfileobj.mode == "xb"
    with open(path, "r+b") as f:
        for mode, want in (("r", CG.READ), ("w", CG.WRITE), ("a", CG.WRITE), ("x", CG.WRITE)):
            with G.GzipFile(fileobj=f, mode=mode) as g:
                assert g.mode == want
        with pytest.raises(ValueError):
            G.GzipFile(fileobj=f, mode="z")
    for mode in ("rb", "r+b"):
        with open(path, mode) as f, G.GzipFile(fileobj=f) as g:
            assert g.mode == CG.READ
    for mode in ("wb", "ab", "xb"):
        if "x" in mode:
            os.unlink(path)
        with open(path, mode) as f:
            with G.GzipFile(fileobj=f) as g:
                assert g.mode == CG.WRITE


def test_buffered_reader_and_lines(G, path):
    # test_buffered_reader / test_readline / test_readlines / test_textio_readlines (reference :235-270, :492-498)
    with G.GzipFile(path, "wb") as f:
        f.write(D1 * 50)
    want = (D1 * 50).splitlines(keepends=True)
    with G.GzipFile(path, "rb") as f, io.BufferedReader(f) as r:
        assert r.readlines() == want
    with G.GzipFile(path, "rb") as f:
        line_length = 0
        while True:
            L = f.readline(line_length)
            if not L and line_length != 0:
                break
            assert len(L) <= line_length
            line_length = (line_length + 1) % 50
    with G.GzipFile(path, "rb") as f:
        assert f.readlines() == want
    with G.GzipFile(path, "rb") as f:
        while True:
            if f.readlines(150) == []:
                break
    with G.GzipFile(path, "r") as f, io.TextIOWrapper(f, encoding="ascii") as t:
        assert t.readlines() == (D1 * 50).decode("ascii").splitlines(keepends=True)


def test_seeking(G, path):
    # test_seek_read / test_seek_whence / test_seek_write (reference :272-307)
    with G.GzipFile(path, "wb") as f:
        f.write(D1 * 50)

    def walk(m):
        got = []
        with m.GzipFile(path) as f:
            while True:
                oldpos = f.tell()
                line1 = f.readline()
                if not line1:
                    break
                newpos = f.tell()
                f.seek(oldpos)
                amount = min(len(line1), 10)
                line2 = f.read(amount)
                got.append((oldpos, newpos, line1[:amount] == line2))
                f.seek(newpos)
            f.read(10)
            f.seek(10, whence=1)
            got.append(f.read(10))
            got.append(f.seek(0, 2) if sys.version_info >= (3, 7) else None)
            got.append(f.seek(-5, 1))
            got.append(f.read())
        return got
    same(walk, G)

    def sparse(m):
        buf = io.BytesIO()
        with m.GzipFile(fileobj=buf, mode="w") as f:
            for pos in range(0, 256, 16):
                f.seek(pos)
                f.write(b"GZ\n")
            try:
                f.seek(3)
            except OSError:
                pass
            else:
                raise AssertionError("negative seek in write mode accepted")
        return CG.decompress(buf.getvalue())
    same(sparse, G)


def test_seek_over_members(G, path):
    # reference tests/test_gzip_ng.py:428-463 (test_seek)
    with open(path, "wb") as f:
        for c in b"ABCD":
            f.write(CG.compress(b"X" * 500 + bytes([c]) + b"X" * 499))
    with G.open(path, "rb") as f:
        for pos, want in ((500, b"A"), (1500, b"B"), (500, b"A")):
            f.seek(pos)
            assert f.read(1) == want
            f.seek(pos, io.SEEK_SET)
            assert f.read(1) == want
        f.seek(500)
        f.seek(2000, io.SEEK_CUR)
        assert f.read(1) == b"C"
        f.seek(-1001, io.SEEK_CUR)
        assert f.read(1) == b"B"
        f.seek(200, io.SEEK_END)
        assert f.read(1) == b""
        f.seek(-1500, io.SEEK_END)
        assert f.read(1) == b"C"


# --------------------------------------------------------------------------------------------- header and trailer

def test_mtime_and_metadata(G, path):
    # test_mtime / test_metadata / test_compresslevel_metadata (reference :329-416)
    mtime = 123456789
    with G.GzipFile(path, "w", mtime=mtime) as f:
        f.write(D1)
    with G.GzipFile(path) as f:
        assert hasattr(f, "mtime") and f.mtime is None
        assert f.read() == D1
        assert f.mtime == mtime
    raw = open(path, "rb").read()
    name = os.path.basename(path)[:-3].encode("latin-1") + b"\x00"        # GzipFile stores the name without ".gz"
    assert raw[:4] == b"\x1f\x8b\x08\x08" and raw[4:8] == struct.pack("<i", mtime)
    assert raw[8:9] == b"\x02" and raw[9:10] == b"\xff" and raw[10:10 + len(name)] == name
    assert raw[-8:] == struct.pack("<II", CZ.crc32(D1), len(D1))
    for level, xfl in ((1, b"\x04"), (9, b"\x02"), (6, b"\x00")):
        with G.GzipFile(path, "w", compresslevel=level) as f:
            f.write(D1)
        assert open(path, "rb").read()[8:9] == xfl


def test_padding_junk_and_unseekable(G, path):
    # test_zero_padded_file / test_bad_gzip_file / test_non_seekable_file / BadGzipFile class (reference :439-468)
    with G.GzipFile(path, "wb") as f:
        f.write(D1 * 50)
    with open(path, "ab") as f:
        f.write(b"\x00" * 50)
    with G.GzipFile(path, "rb") as f:
        assert f.read() == D1 * 50
    assert issubclass(G.BadGzipFile, OSError) and G.BadGzipFile is CG.BadGzipFile
    with open(path, "wb") as f:
        f.write(D1 * 50)
    with G.GzipFile(path, "r") as f:
        with pytest.raises(G.BadGzipFile):
            f.readlines()
    buf = Unseekable()
    with G.GzipFile(fileobj=buf, mode="wb") as f:
        f.write(D1 * 50)
    z = buf.getvalue()
    assert CG.decompress(z) == D1 * 50
    for wr in (z, CG.compress(D1 * 50)):
        with G.GzipFile(fileobj=Unseekable(wr), mode="rb") as f:
            assert f.read() == D1 * 50


def test_peek(G, path):
    # test_peek (reference :470-490)
    data = D1 * 200
    with G.GzipFile(path, "wb") as f:
        f.write(data)

    def sizes():
        while True:
            yield from range(5, 50, 10)
    with G.GzipFile(path, "rb") as f:
        f.max_read_chunk = 33
        nread = 0
        for n in sizes():
            s = f.peek(n)
            if s == b"":
                break
            assert f.read(len(s)) == s
            nread += len(s)
        assert f.read(100) == b"" and nread == len(data)


def test_fileobj_from_fdopen(G, path):
    # test_fileobj_from_fdopen (reference :500-506)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT)
    with os.fdopen(fd, "wb") as f:
        with G.GzipFile(fileobj=f, mode="w"):
            pass
    assert CG.open(path).read() == b""


def test_decompression_is_bounded_per_read(G):
    # test_decompress_limited (reference :547-557)
    bomb = CG.compress(b"\0" * int(2e6), compresslevel=9)
    assert len(bomb) < io.DEFAULT_BUFFER_SIZE
    f = G.GzipFile(fileobj=io.BytesIO(bomb))
    assert f.read(1) == b"\0"
    assert f._buffer.raw.tell() <= 1 + io.DEFAULT_BUFFER_SIZE


def test_repr_and_read_only(G, path):
    # reference tests/test_gzip_ng.py:32-43
    with G.GzipNGFile(path, "wb") as f:
        assert "<gzip_ng _io.BufferedWriter name='" in repr(f)
    with G.GzipNGFile(path, "rb") as f:
        with pytest.raises(OSError, match=r"write\(\) on read-only GzipNGFile object"):
            f.write(b"bla")


# ---------------------------------------------------------------------------------------------------- shortcuts

def test_compress_shortcut(G):
    # test_compress / test_compress_mtime / test_compress_correct_level (reference :561-589)
    for data in (D1, D2):
        for args in ((), (1,), (6,), (9,)):
            z = G.compress(data, *args)
            assert type(z) is bytes
            with CG.GzipFile(fileobj=io.BytesIO(z), mode="rb") as f:
                assert f.read() == data
            z = G.compress(data, *args, mtime=123456789)
            with G.GzipFile(fileobj=io.BytesIO(z), mode="rb") as f:
                f.read(1)
                assert f.mtime == 123456789
    for mtime in (0, 42):
        assert D1 in G.compress(D1, compresslevel=0, mtime=mtime)
        assert D1 not in G.compress(D1, compresslevel=1, mtime=mtime)


def test_decompress_shortcut(G):
    # test_decompress / truncated / missing trailer (reference :591-607); gzip_ng :257-293, :320-345
    for data in (D1, D2):
        buf = io.BytesIO()
        with G.GzipFile(fileobj=buf, mode="wb") as f:
            f.write(data)
        assert G.decompress(buf.getvalue()) == data == CG.decompress(buf.getvalue())
        assert G.decompress(G.compress(data)) == data
    z = CG.compress(D1)
    assert G.decompress(z + z) == D1 + D1
    assert G.decompress(z + b"\x00\x00\x00" + z) == D1 + D1
    assert G.decompress(b"") == b""
    for cut in (4, 8):
        with pytest.raises(EOFError, match="Compressed file ended before the end-of-stream marker was reached"):
            G.decompress(z[:-cut])
    with pytest.raises(G.BadGzipFile, match="Incorrect length of data produced"):
        G.decompress(z[:-4] + (27890).to_bytes(4, "little"))
    with pytest.raises(G.BadGzipFile, match="CRC check failed"):
        G.decompress(z[:-8] + CZ.crc32(D1, 50).to_bytes(4, "little") + z[-4:])
    with pytest.raises(G.BadGzipFile, match=r"Not a gzipped file \(b'Th'\)"):
        G.decompress(b"This is not a gzip data stream.")
    with pytest.raises(G.BadGzipFile, match="Unknown compression method"):
        G.decompress(z[:2] + b"\x09" + z[3:])
    with pytest.raises(G.BadGzipFile):
        G.decompress(b"00")


def test_reading_truncated_files(G):
    # test_read_truncated (reference :609-620); gzip_ng test_GzipNGFile_read_truncated (:58-65)
    truncated = CG.compress(D1 * 50)[:-8]
    for _ in range(2):
        with G.GzipFile(fileobj=io.BytesIO(truncated)) as f:
            with pytest.raises(EOFError):
                f.read()
    for i in range(2, 10):
        with G.GzipFile(fileobj=io.BytesIO(truncated[:i])) as f:
            with pytest.raises(EOFError):
                f.read(1)
    with G.GzipFile(fileobj=io.BytesIO(CG.compress(b"short")[:-10]), mode="rb") as f:
        with pytest.raises(EOFError, match="Compressed file ended before the end-of-stream marker was reached"):
            f.read()


def _headers():
    start, end = b"\x1f\x8b\x08", b"\x00\x00\x00\x00\x00\xff"
    xtra, fname, fcomment = b"METADATA", b"my_data.tar", b"a header written by hand"
    yield start + bytes([CG.FEXTRA]) + end + len(xtra).to_bytes(2, "little") + xtra
    yield start + bytes([CG.FNAME]) + end + fname + b"\x00"
    yield start + bytes([CG.FCOMMENT]) + end + fcomment + b"\x00"
    h = start + bytes([CG.FHCRC]) + end
    yield h + (CZ.crc32(h) & 0xFFFF).to_bytes(2, "little")
    h = (start + bytes([CG.FTEXT | CG.FEXTRA | CG.FNAME | CG.FCOMMENT | CG.FHCRC]) + end +
         len(xtra).to_bytes(2, "little") + xtra + fname + b"\x00" + fcomment + b"\x00")
    yield h + (CZ.crc32(h) & 0xFFFF).to_bytes(2, "little")


def test_optional_header_fields(G):
    # reference tests/test_gzip_ng.py:348-377 (headers) and test_read_with_extra (compliance :622-628)
    co = CZ.compressobj(wbits=-15)
    body = co.compress(D1) + co.flush()
    trailer = struct.pack("<II", CZ.crc32(D1), len(D1))
    for h in _headers():
        member = h + body + trailer
        assert G.decompress(member) == D1
        with G.GzipFile(fileobj=io.BytesIO(member + member)) as f:
            assert f.read() == D1 + D1
    # a wrong header CRC is noticed
    h = list(_headers())[3]
    bad = h[:-2] + bytes([h[-2] ^ 1, h[-1]]) + body + trailer
    with pytest.raises(G.BadGzipFile):
        G.decompress(bad)


@pytest.mark.parametrize("trunc", [
    b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00",                   # no OS byte
    b"\x1f\x8b\x08\x02\x00\x00\x00\x00\x00\xff",               # FHCRC without the checksum
    b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff",               # FEXTRA without XLEN
    b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\xaa\x00",       # FEXTRA, XLEN, no field
    b"\x1f\x8b\x08\x08\x00\x00\x00\x00\x00\xff",               # FNAME without the name
    b"\x1f\x8b\x08\x10\x00\x00\x00\x00\x00\xff",               # FCOMMENT without the comment
])
def test_truncated_headers(G, trunc):
    # reference tests/test_gzip_ng.py:385-398
    with pytest.raises(EOFError):
        G.decompress(trunc)
    with G.GzipFile(fileobj=io.BytesIO(trunc)) as f:
        with pytest.raises(EOFError):
            f.read()


def test_very_long_header(G):
    # reference tests/test_gzip_ng.py:401-417
    co = CZ.compressobj(3, CZ.DEFLATED, -15)
    empty = co.compress(b"") + co.flush()
    for n in (256 * 1024, G.READ_BUFFER_SIZE * 2):
        member = b"\x1f\x8b\x08\x08\x00\x00\x00\x00\x00\xff" + n * b"A" + b"\x00" + empty + 8 * b"\x00"
        assert G.decompress(member) == b""
        with G.open(io.BytesIO(member)) as f:
            assert f.read() == b""


def test_reference_data_files(G, fastq):
    # reference tests/test_gzip_ng.py:420-426 (concatenated) and :466-473 (bgzip); files under tests/golden
    for name in sorted(os.listdir(GOLDEN)):
        if name.endswith(".gz"):
            p = os.path.join(GOLDEN, name)
            want = CG.decompress(open(p, "rb").read())
            with G.open(p, "rb") as f:
                assert f.read() == want, name


def test_stream_longer_than_isize_can_count(G):
    # reference tests/test_gzip_ng.py:295-317 (test_decompress_on_long_input), scaled: ISIZE is checked modulo 2**32, shown
    # here by a trailer that is only right modulo 2**32 being refused and the true one accepted
    data = bytes(1 << 20) + b"\x01" * 123
    z = G.compress(data)
    assert G.decompress(z) == data
    wrong = z[:-4] + struct.pack("<I", (len(data) + 1) & 0xFFFFFFFF)
    with pytest.raises(G.BadGzipFile):
        G.decompress(wrong)
    buf = io.BytesIO()
    with G.open(buf, "wb") as gz:
        for _ in range(64):
            gz.write(bytes(1 << 20))
        gz.write(b"\x01" * 123)
    buf.seek(0)
    with G.open(buf, "rb") as gz:
        for _ in range(64):
            assert gz.read(1 << 20) == bytes(1 << 20)
        assert gz.read() == b"\x01" * 123


# --------------------------------------------------------------------------------------------------------- open()

def test_open_binary_modes(G, path):
    # TestOpen.test_binary_modes / test_implicit_binary_modes / test_pathlike_file (reference :640-702)
    data = D1 * 50
    for explicit in (True, False):
        b = "b" if explicit else ""
        with G.open(path, "w" + b) as f:
            f.write(data)
        assert CG.decompress(open(path, "rb").read()) == data
        with G.open(path, "r" + b) as f:
            assert f.read() == data
        with G.open(path, "a" + b) as f:
            f.write(data)
        assert CG.decompress(open(path, "rb").read()) == data * 2
        with pytest.raises(FileExistsError):
            G.open(path, "x" + b)
        os.unlink(path)
        with G.open(path, "x" + b) as f:
            f.write(data)
        assert CG.decompress(open(path, "rb").read()) == data
    p = pathlib.Path(path)
    with G.open(p, "wb") as f:
        f.write(data)
    with G.open(p, "ab") as f:
        f.write(D1)
    with G.open(p) as f:
        assert f.read() == data + D1


def test_open_text_modes(G, path):
    # TestOpen.test_text_modes / test_encoding / test_encoding_error_handler / test_newline (reference :704-790)
    text = (D1 * 50).decode("ascii")
    native = text.replace("\n", os.linesep)
    with G.open(path, "wt", encoding="ascii") as f:
        f.write(text)
    assert CG.decompress(open(path, "rb").read()).decode("ascii") == native
    with G.open(path, "rt", encoding="ascii") as f:
        assert f.read() == text
    with G.open(path, "at", encoding="ascii") as f:
        f.write(text)
    assert CG.decompress(open(path, "rb").read()).decode("ascii") == native * 2
    with G.open(path, "wt", encoding="utf-16") as f:
        f.write(text)
    assert CG.decompress(open(path, "rb").read()).decode("utf-16") == native
    with G.open(path, "rt", encoding="utf-16") as f:
        assert f.read() == text
    with G.open(path, "wb") as f:
        f.write(b"foo\xffbar")
    with G.open(path, "rt", encoding="ascii", errors="ignore") as f:
        assert f.read() == "foobar"
    with G.open(path, "wt", encoding="ascii", newline="\n") as f:
        f.write(text)
    with G.open(path, "rt", encoding="ascii", newline="\r") as f:
        assert f.readlines() == [text]


def test_open_fileobj_and_bad_parameters(G, path):
    # TestOpen.test_fileobj / test_bad_params (reference :720-744)
    data = D1 * 50
    z = CG.compress(data)
    with G.open(io.BytesIO(z), "r") as f:
        assert f.read() == data
    with G.open(io.BytesIO(z), "rb") as f:
        assert f.read() == data
    with G.open(io.BytesIO(z), "rt", encoding="ascii") as f:
        assert f.read() == data.decode("ascii")
    for args, kw, exc in (((123.456,), {}, TypeError), ((path, "wbt"), {}, ValueError), ((path, "xbt"), {}, ValueError),
                          ((path, "rb"), {"encoding": "utf-8"}, ValueError), ((path, "rb"), {"errors": "ignore"}, ValueError),
                          ((path, "rb"), {"newline": "\n"}, ValueError)):
        with pytest.raises(exc):
            G.open(*args, **kw)
        with pytest.raises(exc):
            CG.open(*args, **kw)


# ------------------------------------------------------------------------------------------------- command line

def _run_main(G, argv, stdin=b"", monkeypatch=None):
    monkeypatch.setattr(sys, "argv", [""] + argv)
    monkeypatch.setattr(sys, "stdin", io.TextIOWrapper(io.BytesIO(stdin)))
    G.main()


DATA = b"A small payload for the command line of gzip_ng"


@pytest.mark.parametrize("level", range(1, 10))
def test_cli_stdin_stdout(G, capsysbinary, monkeypatch, level):
    # reference tests/test_gzip_ng.py:68-89
    _run_main(G, ["-d"], CG.compress(DATA, level), monkeypatch)
    out, err = capsysbinary.readouterr()
    assert (out, err) == (DATA, b"")
    _run_main(G, [f"-{level}"], DATA, monkeypatch)
    out, err = capsysbinary.readouterr()
    assert err == b"" and CG.decompress(out) == DATA


def test_cli_files(G, tmp_path, capsysbinary, monkeypatch):
    # reference tests/test_gzip_ng.py:92-178, :242-254
    plain, gz = tmp_path / "test", tmp_path / "test.gz"
    gz.write_bytes(CG.compress(DATA))
    _run_main(G, ["-d", str(gz)], b"", monkeypatch)
    assert capsysbinary.readouterr() == (b"", b"") and plain.read_bytes() == DATA
    gz.unlink()
    _run_main(G, [str(plain)], b"", monkeypatch)
    assert capsysbinary.readouterr() == (b"", b"") and CG.decompress(gz.read_bytes()) == DATA
    with pytest.raises(SystemExit, match="filename doesn't end"):
        _run_main(G, ["-d", "thisisatest.out"], b"", monkeypatch)
    assert capsysbinary.readouterr().out == b""
    nogz = tmp_path / "noext"
    nogz.write_bytes(CG.compress(DATA))
    _run_main(G, ["-cd", str(nogz)], b"", monkeypatch)
    assert capsysbinary.readouterr().out == DATA
    _run_main(G, ["-cd", str(gz)], b"", monkeypatch)
    assert capsysbinary.readouterr() == (DATA, b"")
    _run_main(G, ["-c", str(plain)], b"", monkeypatch)
    out, err = capsysbinary.readouterr()
    assert err == b"" and CG.decompress(out) == DATA
    other = tmp_path / "out"
    _run_main(G, ["-d", "-o", str(other), str(gz)], b"", monkeypatch)
    assert capsysbinary.readouterr() == (b"", b"") and other.read_bytes() == DATA
    comp = tmp_path / "compressed.gz"
    _run_main(G, ["-o", str(comp), str(plain)], b"", monkeypatch)
    assert capsysbinary.readouterr() == (b"", b"") and CG.decompress(comp.read_bytes()) == DATA
    _run_main(G, ["-n", "-f", "-o", str(comp), str(plain)], b"", monkeypatch)
    raw = comp.read_bytes()
    assert CG.decompress(raw) == DATA and raw[3] & CG.FNAME == 0 and raw[4:8] == b"\x00\x00\x00\x00"


def test_cli_overwrite_prompt(G, tmp_path, capsysbinary, monkeypatch):
    # reference tests/test_gzip_ng.py:181-240
    plain, comp, implicit = tmp_path / "test", tmp_path / "compressed.gz", tmp_path / "test.gz"
    plain.write_bytes(DATA)
    comp.touch()
    _run_main(G, ["-f", "-o", str(comp), str(plain)], b"", monkeypatch)
    assert capsysbinary.readouterr() == (b"", b"") and CG.decompress(comp.read_bytes()) == DATA
    with pytest.raises(EOFError):
        _run_main(G, ["-o", str(comp), str(plain)], b"", monkeypatch)
    assert b"compressed.gz already exists; do you wish to overwrite (y/n)?" in capsysbinary.readouterr().out
    implicit.touch()
    with pytest.raises(SystemExit, match="not overwritten"):
        _run_main(G, [str(plain)], b"n", monkeypatch)
    assert b"test.gz already exists; do you wish to overwrite (y/n)?" in capsysbinary.readouterr().out
    _run_main(G, [str(plain)], b"y", monkeypatch)
    out, err = capsysbinary.readouterr()
    assert b"already exists; do you wish to overwrite" in out and err == b""
    assert CG.decompress(implicit.read_bytes()) == DATA


def test_cli_flag_conflicts(G, monkeypatch, capsys):
    # TestCommandLine.test_compress_fast_best_are_exclusive / test_decompress_cannot_have_flags_compression (:878-893)
    for argv in (["--fast", "--best"], ["--fast", "-d"], ["-9", "-d"]):
        with pytest.raises(SystemExit):
            _run_main(G, argv, b"", monkeypatch)
        assert "not allowed with argument" in capsys.readouterr().err


# ------------------------------------------------------------------------------- every single-bit corruption of small files

def test_every_single_bit_flip_of_small_files(G):
    """One- and two-member files with every bit flipped once, through decompress() and through open().read(): the same
    verdict (bytes, or the class of the exception) as CPython's gzip.  One documented difference: with the FHCRC bit set the
    reference's reader verifies the header checksum (BadGzipFile, zlib_ngmodule.c:2496-2510); CPython 3.10 skips the field
    and fails later inside inflate."""
    from zlib_ng_amd import zlib_ng

    def verdict(fn, m, zerr):
        try:
            return ("ok", fn(m))
        except Exception as e:   # noqa: BLE001
            return ("raise", "zlib.error" if isinstance(e, zerr) else type(e).__name__)
    for payload in (b"", b"a", b"hello hello hello, hello?", D1[:300]):
        for second in (False, True):
            z = CG.compress(payload, 6, mtime=5)
            fhcrc_at = {3}
            if second:
                fhcrc_at.add(len(z) + 3)
                z = z + CG.compress(b"second member", 6, mtime=0)
            differ = []
            for pos in range(len(z)):
                for bit in range(8):
                    zz = bytearray(z)
                    zz[pos] ^= 1 << bit
                    zz = bytes(zz)

                    def rd(m):
                        with m.open(io.BytesIO(zz), "rb") as f:
                            return f.read()
                    for name, fn in (("decompress", lambda m: m.decompress(zz)), ("open", rd)):
                        a, b = verdict(fn, CG, CZ.error), verdict(fn, G, zlib_ng.error)
                        if a != b and not (pos in fhcrc_at and bit == 1 and b == ("raise", "BadGzipFile")):
                            differ.append((name, pos, bit, a, b))
            assert not differ, (len(differ), differ[:6])


# ------------------------------------------------------------------------------------- files of many small members

def _verdict(fn, m, zerr):
    try:
        return ("ok", fn(m))
    except Exception as e:   # noqa: BLE001
        return ("raise", "zlib.error" if isinstance(e, zerr) else type(e).__name__)


def test_many_small_members_in_one_launch(G, fastq):
    """Concatenated small members (logs, `cat *.gz`, files grown by appending) are sized and decoded in two launches, not member
    by member (csrc/zng_amd.hip: hop_plain_members).  Member starts are found by their magic bytes, so the payloads here
    contain the magic themselves (stored blocks: verbatim), headers carry names, padding sits between members, and a large
    member or a damaged one ends the run and is handled by the member loop -- always CPython gzip's verdict."""
    from zlib_ng_amd import _lib, zlib_ng
    ctx = _lib.default_context()
    rnd = random.Random(17)
    magic = b"\x1f\x8b\x08\x00" + bytes(6)
    parts, want = [], []
    for i in range(400):
        n = rnd.choice([0, 1, 300, 5000, 40000])
        o = rnd.randrange(0, len(fastq) - n)
        payload = fastq[o:o + n] + (magic * rnd.randrange(1, 4) if i % 3 == 0 else b"")
        level = rnd.choice([0, 1, 6, 9])
        if i % 5 == 0:
            buf = io.BytesIO()
            with CG.GzipFile(filename="name-%d.txt" % i, mode="wb", fileobj=buf, compresslevel=level, mtime=i) as f:
                f.write(payload)
            parts.append(buf.getvalue())
        else:
            parts.append(CG.compress(payload, level, mtime=0))
        if i % 7 == 0:
            parts.append(bytes(rnd.randrange(1, 9)))              # zero padding between members
        want.append(payload)
    blob, plain = b"".join(parts), b"".join(want)
    ctx.decode_paths()
    assert G.decompress(blob) == plain
    paths = ctx.decode_paths()
    assert paths["bgzf"] == 400 and paths["sequential"] == 0, paths          # "bgzf" counts members decoded one wavefront each in one launch
    with G.open(io.BytesIO(blob), "rb") as f:
        assert f.read() == plain
    # a large member in the middle, and small ones behind it
    big = fastq * 3
    mixed = b"".join(parts[:120]) + CG.compress(big, 6) + b"".join(parts[120:])
    head = len(b"".join(parts[:120]))
    expect = CG.decompress(mixed)
    assert G.decompress(mixed) == expect
    # damage: a flipped bit in member 57's payload, a cut in the last member, junk behind the last member
    starts = [0]
    for p in parts:
        starts.append(starts[-1] + len(p))
    for bad in (blob[:starts[57] + 14] + bytes([blob[starts[57] + 14] ^ 0x10]) + blob[starts[57] + 15:],
                blob[:-3], blob[:-9], blob + b"junk behind the members", blob[:starts[300] + 5]):
        a = _verdict(lambda m: m.decompress(bad), CG, CZ.error)
        b = _verdict(lambda m: m.decompress(bad), G, zlib_ng.error)
        assert a == b, (a[:1], b[:1], a[1] if a[0] == "raise" else len(a[1]), b[1] if b[0] == "raise" else len(b[1]))

        def rd(m):
            with m.open(io.BytesIO(bad), "rb") as f:
                return f.read()
        a, b = _verdict(rd, CG, CZ.error), _verdict(rd, G, zlib_ng.error)
        assert a == b, (a[:1], b[:1])
    assert head > 0


def test_line_sized_writes_are_collected(G, fastq, tmp_path):
    """The reference's benchmark_scripts/gzipwritelines.py pattern: thousands of writes of one line each, with flushes, a large
    write and a memoryview in between; the file is what one write of everything gives, for the stdlib and for our reader."""
    lines = fastq[:2_000_000].splitlines(keepends=True)
    p = tmp_path / "lines.gz"
    big = fastq[2_000_000:2_300_000]
    with G.open(p, "wb", compresslevel=6) as f:
        for i, line in enumerate(lines):
            assert f.write(line) == len(line)
            if i == 1000:
                f.flush()
            if i == 5000:
                assert f.write(memoryview(big)) == len(big)
            if i == 7000:
                assert f.write(bytearray(b"xyz")) == 3
        assert f.tell() == sum(map(len, lines)) + len(big) + 3
    want = b"".join(lines[:5001]) + big + b"".join(lines[5001:7001]) + b"xyz" + b"".join(lines[7001:])
    assert CG.decompress(p.read_bytes()) == want
    with G.open(p, "rb") as f:
        assert b"".join(f) == want
    # nothing written, and a single tiny write
    for payload in (b"", b"a"):
        q = tmp_path / "tiny.gz"
        with G.open(q, "wb") as f:
            f.write(payload)
        assert CG.decompress(q.read_bytes()) == payload
