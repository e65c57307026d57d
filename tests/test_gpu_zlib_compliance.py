"""Behavioural compliance of zlib_ng_amd.zlib_ng with CPython's zlib, run differentially: every scenario is a
function of the module under test, executed once against the stdlib `zlib` and once against the GPU-backed
module, and the observable results (decompressed bytes, attribute values, exception types, message patterns)
must agree.  Compressed bytes are never compared (the ZA codec picks its own matches); each stream one module
writes is read back by the other.

The scenarios follow the behaviours the reference pins in tests/test_zlib_compliance.py (class and line given
at each test); the data and the harness are this repo's own."""
import copy
import pickle
import random
import sys
import zlib as CZ

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Z():
    from zlib_ng_amd import zlib_ng
    return zlib_ng


def _prose(seed, n):
    rnd = random.Random(seed)
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 9))) for _ in range(300)]
    out = bytearray()
    while len(out) < n:
        out += rnd.choice(words) + (b" " if rnd.random() < 0.85 else b".\n")
    return bytes(out[:n])


TEXT = _prose(11, 2900)          # about the size of the scene the reference uses
TEXT128 = TEXT * 128


class _Index:
    """An integer-like argument (reference: CustomInt, test_zlib_compliance.py:1161)."""

    def __index__(self):
        return 100


def outcome(fn, mod):
    try:
        return ("ok", fn(mod))
    except Exception as e:   # noqa: BLE001 - the exception class is the observation
        name = type(e).__name__
        if isinstance(e, mod.error):
            name = "zlib.error"
        return ("raise", name)


def same(fn, Z):
    a, b = outcome(fn, CZ), outcome(fn, Z)
    assert a == b, (a if len(repr(a)) < 300 else repr(a)[:300], b if len(repr(b)) < 300 else repr(b)[:300])
    return b


# ------------------------------------------------------------------------------------------------ checksums

def test_checksum_start_values_and_empty(Z):
    # ChecksumTestCase.test_crc32start/empty, test_adler32start/empty (reference :76-92)
    for v in (0, 1, 432, 0xFFFFFFFF, 2 ** 32 + 7, -1):
        same(lambda m: m.crc32(b"", v), Z)
        same(lambda m: m.adler32(b"", v), Z)
        same(lambda m: m.crc32(b"xyzzy", v), Z)
        same(lambda m: m.adler32(b"xyzzy", v), Z)
    assert Z.crc32(b"") == Z.crc32(b"", 0) and Z.adler32(b"") == Z.adler32(b"", 1)


def test_checksum_argument_types(Z):
    # ExceptionTestCase.test_badargs (reference :147-156)
    for f in ("adler32", "crc32", "compress", "decompress"):
        same(lambda m: getattr(m, f)(), Z)
        for arg in (42, None, "", "abc", (), []):
            same(lambda m: getattr(m, f)(arg), Z)
    for ob in (bytearray(b"spam"), memoryview(b"spam"), memoryview(b"-spam-")[1:5]):
        assert Z.crc32(ob) == CZ.crc32(b"spam") and Z.adler32(ob) == CZ.adler32(b"spam")


def test_crc32_combine_identities(Z):
    # ChecksumTestCase.test_crc32_combine (reference :111-120); the stdlib has no crc32_combine to diff against
    assert Z.crc32_combine(0, 0, 0) == 0 and Z.crc32_combine(1, 0, 0) == 1 and Z.crc32_combine(432, 0, 0) == 432
    rnd = random.Random(5)
    for _ in range(20):
        a, b = TEXT128[:rnd.randrange(0, 70000)], TEXT128[:rnd.randrange(0, 70000)]
        assert Z.crc32_combine(CZ.crc32(a), CZ.crc32(b), len(b)) == CZ.crc32(a + b)


# ----------------------------------------------------------------------------------------------- bad arguments

def test_bad_level_and_constructor_arguments(Z):
    # ExceptionTestCase.test_badlevel/badcompressobj/baddecompressobj/decompressobj_badflush (reference :141-172)
    same(lambda m: m.compress(b"ERROR", 10), Z)
    same(lambda m: m.compress(b"ERROR", -2), Z)
    same(lambda m: m.compressobj(1, m.DEFLATED, 0), Z)
    same(lambda m: m.compressobj(1, m.DEFLATED, m.MAX_WBITS + 1), Z)
    same(lambda m: m.compressobj(10) and None, Z)
    same(lambda m: m.decompressobj(-1), Z)
    same(lambda m: m.decompressobj().flush(0), Z)
    same(lambda m: m.decompressobj().flush(-1), Z)
    same(lambda m: m.decompressobj().decompress(b"", -1), Z)


def test_keyword_forms(Z):
    # CompressTestCase.test_keywords, CompressObjectTestCase.test_keywords (reference :229-238, :312-331)
    same(lambda m: m.decompress(m.compress(TEXT, level=3)), Z)
    same(lambda m: m.compress(data=TEXT, level=3), Z)
    same(lambda m: m.decompress(m.compress(TEXT), wbits=m.MAX_WBITS, bufsize=m.DEF_BUF_SIZE), Z)
    same(lambda m: m.compressobj().compress(data=TEXT), Z)
    same(lambda m: m.decompressobj().decompress(data=CZ.compress(TEXT)), Z)

    def full_keywords(m):
        co = m.compressobj(level=2, method=m.DEFLATED, wbits=-12, memLevel=9, strategy=m.Z_FILTERED, zdict=b"")
        do = m.decompressobj(wbits=-12, zdict=b"")
        x = co.compress(TEXT) + co.flush()
        return do.decompress(x, max_length=len(TEXT)) + do.flush()
    assert same(full_keywords, Z) == ("ok", TEXT)


def test_overflowing_sizes(Z):
    # ExceptionTestCase.test_overflow (reference :174-181)
    same(lambda m: m.decompress(b"", 15, sys.maxsize + 1), Z)
    same(lambda m: m.decompressobj().decompress(b"", sys.maxsize + 1), Z)
    same(lambda m: m.decompressobj().flush(sys.maxsize + 1), Z)


# ------------------------------------------------------------------------------------------------- one shot

def test_one_shot_round_trip_and_input_kinds(Z):
    # CompressTestCase.test_speech/test_speech128 (reference :225-247)
    for data in (TEXT, TEXT128):
        x = Z.compress(data)
        assert Z.compress(bytearray(data)) == x and Z.compress(memoryview(data)) == x
        for ob in (x, bytearray(x), memoryview(x)):
            assert Z.decompress(ob) == data and CZ.decompress(ob) == data
        assert isinstance(Z.decompress(x), bytes)


def test_truncated_stream_message(Z):
    # CompressTestCase.test_incomplete_stream (reference :249-255)
    for mk in (CZ, Z):
        x = mk.compress(TEXT)
        for cut in (1, 4, 5, len(x) // 2):
            with pytest.raises(Z.error, match="Error -5 while decompressing data: incomplete or truncated stream"):
                Z.decompress(x[:-cut])
    with pytest.raises(Z.error, match="Error -5"):
        Z.decompress(b"")


def test_bufsize_forms(Z):
    # CompressTestCase.test_custom_bufsize / test_large_bufsize (reference :269-281)
    x = CZ.compress(TEXT * 10, 1)
    for bufsize in (1, 17, _Index(), 2 ** 20, 2 ** 32):
        assert Z.decompress(x, 15, bufsize) == TEXT * 10
    same(lambda m: m.decompress(x, 15, 0), Z)
    same(lambda m: m.decompress(x, 15, -1), Z)


# ------------------------------------------------------------------------------------------ compression objects

def test_pair_and_second_flush(Z):
    # CompressObjectTestCase.test_pair (reference :293-310)
    def run(m):
        res = []
        for data in (TEXT128, bytearray(TEXT128)):
            co = m.compressobj()
            x = co.compress(data) + co.flush()
            try:
                co.flush()
                res.append("second flush accepted")
            except m.error:
                res.append("second flush refused")
            for v in (x, bytearray(x)):
                dco = m.decompressobj()
                y = dco.decompress(v) + dco.flush()
                res.append((y == bytes(data), type(dco.unconsumed_tail).__name__, type(dco.unused_data).__name__))
        return res
    same(run, Z)


def test_compress_after_finish(Z):
    def run(m):
        co = m.compressobj()
        co.compress(TEXT)
        co.flush()
        return co.compress(b"more")
    same(run, Z)


def test_options(Z):
    # CompressObjectTestCase.test_compressoptions (reference :333-346)
    for level in (0, 1, 2, 6, 9):
        for wbits in (-9, -12, -15, 9, 15, 25, 31):
            for strategy in (CZ.Z_DEFAULT_STRATEGY, CZ.Z_FILTERED, CZ.Z_HUFFMAN_ONLY, CZ.Z_RLE, CZ.Z_FIXED):
                co = Z.compressobj(level, Z.DEFLATED, wbits, 9, strategy)
                x = co.compress(TEXT) + co.flush()
                dco = CZ.decompressobj(wbits)
                assert dco.decompress(x) + dco.flush() == TEXT
                assert dco.eof and dco.unused_data == b""


@pytest.mark.parametrize("step", [256, 1000, 70000])
def test_incremental_compress_one_shot_decompress(Z, step):
    # CompressObjectTestCase.test_compressincremental (reference :348-360)
    co = Z.compressobj()
    bufs = [co.compress(TEXT128[i:i + step]) for i in range(0, len(TEXT128), step)]
    bufs.append(co.flush())
    z = b"".join(bufs)
    assert CZ.decompress(z) == TEXT128
    dco = Z.decompressobj()
    assert dco.decompress(z) + dco.flush() == TEXT128


@pytest.mark.parametrize("flush", [False, True])
@pytest.mark.parametrize("maker", ["stdlib", "ours"])
def test_incremental_decompress(Z, flush, maker):
    # CompressObjectTestCase.test_decompinc / test_decompincflush (reference :362-403)
    z = (CZ if maker == "stdlib" else Z).compress(TEXT128)
    dco = Z.decompressobj()
    bufs = []
    for i in range(0, len(z), 64):
        bufs.append(dco.decompress(z[i:i + 64]))
        assert dco.unconsumed_tail == b"" and dco.unused_data == b""
    if flush:
        bufs.append(dco.flush())
    else:
        while True:
            chunk = dco.decompress(b"")
            if not chunk:
                break
            bufs.append(chunk)
    assert dco.unconsumed_tail == b"" and dco.unused_data == b""
    assert b"".join(bufs) == TEXT128


@pytest.mark.parametrize("flush", [False, True])
def test_decompress_with_max_length(Z, flush):
    # CompressObjectTestCase.test_decompimax / test_decompressmaxlen[flush] (reference :405-465)
    z = CZ.compress(TEXT128)
    for rule in (lambda cb: 64, lambda cb: 1 + len(cb) // 10):
        dco = Z.decompressobj()
        ref = CZ.decompressobj()
        bufs, cb, cr = [], z, z
        while cb:
            k = rule(cb)
            chunk = dco.decompress(cb, k)
            assert chunk == ref.decompress(cr, k)              # same bytes per call, not only in total
            assert len(chunk) <= k
            bufs.append(chunk)
            cb, cr = dco.unconsumed_tail, ref.unconsumed_tail
            # how far ahead of the output the decoder has read is its own business (zlib keeps whole bytes in its bit
            # buffer, the engine stops on a token): the tails agree up to a few bytes and both are suffixes of the input
            assert abs(len(cb) - len(cr)) <= 8 and z.endswith(cb)
        if flush:
            bufs.append(dco.flush())
        else:
            chunk = b"x"
            while chunk:
                chunk = dco.decompress(b"", 64)
                assert len(chunk) <= 64
                bufs.append(chunk)
        assert b"".join(bufs) == TEXT128


def test_max_length_misc(Z):
    # test_maxlenmisc / test_maxlen_large / test_maxlen_custom / test_clear_unconsumed_tail (reference :467-499)
    x = CZ.compress(TEXT * 10, 1)
    assert len(TEXT) * 10 > Z.DEF_BUF_SIZE
    assert Z.decompressobj().decompress(x, sys.maxsize) == TEXT * 10
    assert Z.decompressobj().decompress(x, _Index()) == (TEXT * 10)[:100]

    def clear(m):
        dco = m.decompressobj()
        d = dco.decompress(CZ.compress(b"abc"), 1)
        t1 = dco.unconsumed_tail
        d += dco.decompress(dco.unconsumed_tail)
        return d, len(t1) > 0, dco.unconsumed_tail
    assert same(clear, Z) == ("ok", (b"abc", True, b""))


@pytest.mark.parametrize("level", range(10))
def test_flush_modes(Z, level):
    # CompressObjectTestCase.test_flushes (reference :501-532)
    data = TEXT * 8
    for sync in ("Z_NO_FLUSH", "Z_SYNC_FLUSH", "Z_FULL_FLUSH", "Z_PARTIAL_FLUSH", "Z_BLOCK"):
        obj = Z.compressobj(level)
        a = obj.compress(data[:3000])
        b = obj.flush(getattr(Z, sync))
        c = obj.compress(data[3000:])
        d = obj.flush()
        assert CZ.decompress(a + b + c + d) == data, (sync, level)
        if sync in ("Z_SYNC_FLUSH", "Z_FULL_FLUSH"):
            # everything fed so far must be decodable from what has been returned so far
            assert CZ.decompressobj().decompress(a + b) == data[:3000], (sync, level)


def test_sync_flush_makes_input_visible(Z):
    # CompressObjectTestCase.test_odd_flush (reference :534-568): 17 KiB of random bytes
    data = random.Random(1).randbytes(17 * 1024)
    co = Z.compressobj(Z.Z_BEST_COMPRESSION)
    first = co.compress(data)
    second = co.flush(Z.Z_SYNC_FLUSH)
    assert Z.decompressobj().decompress(first + second) == data
    assert CZ.decompressobj().decompress(first + second) == data


def test_flush_of_unused_objects(Z):
    # CompressObjectTestCase.test_empty_flush (reference :570-578)
    z = Z.compressobj(Z.Z_BEST_COMPRESSION).flush()
    assert z and CZ.decompress(z) == b""
    assert Z.decompressobj().flush() == b""


def test_flush_mode_argument(Z):
    same(lambda m: m.compressobj().flush(99) and None, Z)
    same(lambda m: m.compressobj().flush(-1) and None, Z)
    same(lambda m: m.compressobj().flush("x"), Z)


# ---------------------------------------------------------------------------------------------- dictionaries

def test_dictionary(Z):
    # CompressObjectTestCase.test_dictionary (reference :580-596)
    words = TEXT.split()
    random.Random(3).shuffle(words)
    zdict = b"".join(words)
    for wr in (CZ, Z):
        co = wr.compressobj(zdict=zdict)
        cd = co.compress(TEXT) + co.flush()
        for rd in (CZ, Z):
            dco = rd.decompressobj(zdict=zdict)
            assert dco.decompress(cd) + dco.flush() == TEXT
        same(lambda m: m.decompressobj().decompress(cd), Z)
        same(lambda m: m.decompressobj(zdict=b"some other dictionary").decompress(cd), Z)
        same(lambda m: m.decompress(cd), Z)
    assert len(Z.compressobj(zdict=zdict).compress(b"") + cd) < len(CZ.compress(TEXT))   # the dictionary is used


def test_dictionary_across_sync_flushes(Z):
    # CompressObjectTestCase.test_dictionary_streaming (reference :598-610)
    piece = TEXT[1000:1500]
    for wr in (CZ, Z):
        co = wr.compressobj(zdict=TEXT)
        d = [co.compress(p) + co.flush(wr.Z_SYNC_FLUSH) for p in (piece, piece[100:], piece[:-100])]
        for rd in (CZ, Z):
            do = rd.decompressobj(zdict=TEXT)
            assert [do.decompress(x) for x in d] == [piece, piece[100:], piece[:-100]]


def test_raw_stream_with_dictionary(Z):
    # CompressObjectTestCase.test_decompress_raw_with_dictionary (reference :667-674)
    zdict = b"abcdefghijklmnopqrstuvwxyz"
    for wr in (CZ, Z):
        co = wr.compressobj(wbits=-wr.MAX_WBITS, zdict=zdict)
        comp = co.compress(zdict) + co.flush()
        for rd in (CZ, Z):
            dco = rd.decompressobj(wbits=-rd.MAX_WBITS, zdict=zdict)
            assert dco.decompress(comp) + dco.flush() == zdict


def test_dictionary_argument_types(Z):
    same(lambda m: m.compressobj(zdict="text"), Z)
    same(lambda m: m.decompressobj(zdict="text"), Z)
    same(lambda m: m.decompressobj(zdict=5), Z)
    assert Z.decompressobj(zdict=bytearray(b"abc")) is not None


# --------------------------------------------------------------------------------- ends of streams and leftovers

FOO = CZ.compress(b"foo")


def test_object_tolerates_missing_trailer(Z):
    # CompressObjectTestCase.test_decompress_incomplete_stream (reference :612-623)
    same(lambda m: m.decompress(FOO), Z)
    same(lambda m: m.decompress(FOO[:-5]), Z)

    def run(m):
        dco = m.decompressobj()
        y = dco.decompress(FOO[:-5])
        return y + dco.flush(), dco.eof
    assert same(run, Z) == ("ok", (b"foo", False))


def test_eof_attribute(Z):
    # test_decompress_eof / test_decompress_eof_incomplete_stream (reference :625-645)
    def run(m):
        seen = []
        dco = m.decompressobj()
        seen.append(dco.eof)
        dco.decompress(FOO[:-5]); seen.append(dco.eof)
        dco.decompress(FOO[-5:]); seen.append(dco.eof)
        dco.flush(); seen.append(dco.eof)
        d2 = m.decompressobj()
        d2.decompress(FOO[:-5]); d2.flush(); seen.append(d2.eof)
        return seen
    assert same(run, Z) == ("ok", [False, False, True, True, False])


def test_unused_data_accumulates(Z):
    # CompressObjectTestCase.test_decompress_unused_data (reference :647-665)
    source, remainder = b"abcdefghijklmnopqrstuvwxyz", b"0123456789"
    for wr in (CZ, Z):
        y = wr.compress(source)
        x = y + remainder
        for maxlen in (0, 1000):
            for step in (1, 2, len(y), len(x)):
                def run(m):
                    dco = m.decompressobj()
                    data, trace = b"", []
                    for i in range(0, len(x), step):
                        if i < len(y):
                            trace.append(dco.unused_data)
                        if maxlen == 0:
                            data += dco.decompress(x[i:i + step])
                            trace.append(dco.unconsumed_tail)
                        else:
                            data += dco.decompress(dco.unconsumed_tail + x[i:i + step], maxlen)
                    data += dco.flush()
                    return data, dco.eof, dco.unconsumed_tail, dco.unused_data, trace
                st, (data, eof, tail, unused, _) = same(run, Z)
                assert st == "ok" and data == source and eof and tail == b"" and unused == remainder


def test_feeding_after_eof(Z):
    def run(m):
        dco = m.decompressobj()
        out = dco.decompress(FOO + b"tail-1")
        out2 = dco.decompress(b"tail-2")
        return out, out2, dco.unused_data, dco.flush(), dco.eof
    assert same(run, Z) == ("ok", (b"foo", b"", b"tail-1tail-2", b"", True))


def test_flush_uses_retained_input(Z):
    # test_flush_with_freed_input / test_flush_custom_length (reference :676-709)
    def run(m):
        a = b"abcdefghijklmnopqrstuvwxyz"
        data = bytearray(CZ.compress(a))
        dco = m.decompressobj()
        first = dco.decompress(data, 1)
        data[:] = b"\0" * len(data)
        del data
        m.compress(b"QWERTYUIOPASDFGHJKLZXCVBNM")
        return first, dco.flush()
    assert same(run, Z) == ("ok", (b"a", b"bcdefghijklmnopqrstuvwxyz"))

    def custom(m):
        dco = m.decompressobj()
        dco.decompress(CZ.compress(TEXT * 10, 1), 1)
        return dco.flush(_Index())
    assert same(custom, Z) == ("ok", (TEXT * 10)[1:])


def test_large_leftovers(Z):
    # test_large_unused_data / test_large_unconsumed_tail (reference :805-833), at sizes that fit a test box
    unused = b"x" * (48 << 20)
    do = Z.decompressobj()
    assert do.decompress(CZ.compress(b"abcdefghijklmnop") + unused) + do.flush() == b"abcdefghijklmnop"
    assert do.unused_data == unused
    data = b"y" * (24 << 20)
    do = Z.decompressobj()
    comp = CZ.compress(data, 0)
    first = do.decompress(comp, 1)
    assert first == b"y" and len(do.unconsumed_tail) > (23 << 20)
    assert first + do.flush() == data and do.unconsumed_tail == b""


def test_corrupt_streams(Z):
    good = CZ.compress(TEXT128)
    rnd = random.Random(9)
    for _ in range(24):
        bad = bytearray(good)
        pos = rnd.randrange(2, len(bad) - 4)
        bad[pos] ^= 1 << rnd.randrange(8)
        a = outcome(lambda m: m.decompress(bytes(bad)), CZ)
        b = outcome(lambda m: m.decompress(bytes(bad)), Z)
        assert a[0] == b[0], (pos, a[:1], b[:1])       # both raise (any zlib.error) or both succeed
        if a[0] == "ok":
            assert a == b
    same(lambda m: m.decompress(b"Not a valid deflate block" * 30), Z)
    same(lambda m: m.decompressobj().decompress(b"Not a valid deflate block" * 30), Z)


# ----------------------------------------------------------------------------------------------------- copies

@pytest.mark.parametrize("how", ["method", "copy", "deepcopy"])
def test_compress_copy(Z, how):
    # CompressObjectTestCase.test_compresscopy (reference :711-733)
    func = {"method": lambda c: c.copy(), "copy": copy.copy, "deepcopy": copy.deepcopy}[how]
    data0, data1 = TEXT, TEXT.swapcase()
    c0 = Z.compressobj(Z.Z_BEST_COMPRESSION)
    bufs0 = [c0.compress(data0)]
    c1 = func(c0)
    bufs1 = bufs0[:]
    bufs0 += [c0.compress(data0), c0.flush()]
    bufs1 += [c1.compress(data1), c1.flush()]
    assert CZ.decompress(b"".join(bufs0)) == data0 + data0
    assert CZ.decompress(b"".join(bufs1)) == data0 + data1


@pytest.mark.parametrize("how", ["method", "copy", "deepcopy"])
def test_decompress_copy(Z, how):
    # CompressObjectTestCase.test_decompresscopy (reference :746-769)
    func = {"method": lambda c: c.copy(), "copy": copy.copy, "deepcopy": copy.deepcopy}[how]
    comp = CZ.compress(TEXT)
    d0 = Z.decompressobj()
    bufs0 = [d0.decompress(comp[:32])]
    d1 = func(d0)
    bufs1 = bufs0[:]
    bufs0.append(d0.decompress(comp[32:]))
    bufs1.append(d1.decompress(comp[32:]))
    assert b"".join(bufs0) == b"".join(bufs1) == TEXT


def test_copy_of_finished_objects(Z):
    # test_badcompresscopy / test_baddecompresscopy (reference :735-744, :771-780)
    def comp(m):
        c = m.compressobj()
        c.compress(TEXT)
        c.flush()
        return c
    def dec(m):
        d = m.decompressobj()
        d.decompress(CZ.compress(TEXT))
        d.flush()
        return d
    for mk in (comp, dec):
        for func in (lambda c: c.copy(), copy.copy, copy.deepcopy):
            same(lambda m: func(mk(m)) and None, Z)


def test_objects_do_not_pickle(Z):
    # test_compresspickle / test_decompresspickle / ZlibDecompressorTest.testPickle (reference :782-790, :1084)
    for proto in range(pickle.HIGHEST_PROTOCOL + 1):
        for mk in (lambda: Z.compressobj(9), Z.decompressobj, Z._ZlibDecompressor):
            with pytest.raises((TypeError, pickle.PicklingError)):
                pickle.dumps(mk(), proto)


# ------------------------------------------------------------------------------------------------ window bits

def test_wbits_matrix(Z):
    # CompressObjectTestCase.test_wbits (reference :835-912)
    def mk(m, wbits):
        co = m.compressobj(level=1, wbits=wbits)
        return co.compress(TEXT) + co.flush()
    for wr in (CZ, Z):
        zlib15, zlib9 = mk(wr, 15), mk(wr, 9)
        deflate15, deflate9, gz = mk(wr, -15), mk(wr, -9), mk(wr, 16 + 15)
        for stream, readers in ((zlib15, (15, 0, 32 + 15)), (zlib9, (9, 15, 0, 32 + 9)), (deflate15, (-15,)),
                                (deflate9, (-9, -15)), (gz, (16 + 15, 32 + 15))):
            for wbits in readers:
                assert same(lambda m: m.decompress(stream, wbits), Z) == ("ok", TEXT), (wr.__name__, wbits)
                assert same(lambda m: m.decompressobj(wbits).decompress(stream), Z) == ("ok", TEXT)
        for wbits in (14, 9):
            with pytest.raises(Z.error, match="invalid window size"):
                Z.decompress(zlib15, wbits)
            with pytest.raises(Z.error, match="invalid window size"):
                Z.decompressobj(wbits=wbits).decompress(zlib15)
        # streams of the wrong container
        same(lambda m: m.decompress(gz, 15), Z)
        same(lambda m: m.decompress(zlib15, 31), Z)
        same(lambda m: m.decompress(zlib15, -15), Z)
    for wbits in (-15, 15, 31):      # compress(wbits=) is in the reference (and CPython >= 3.11), not in every stdlib
        x = Z.compress(TEXT, wbits=wbits)
        assert Z.decompress(x, wbits=wbits) == TEXT and CZ.decompress(x, wbits) == TEXT
    for wbits in (8, -8, 16, 7, 40, 48, -16):
        same(lambda m: m.compressobj(wbits=wbits) and None, Z)
        same(lambda m: m.decompressobj(wbits=wbits).decompress(FOO), Z)


# ------------------------------------------------------------------------------------------- _ZlibDecompressor

DATA = CZ.compress(TEXT)
BIG_TEXT = DATA * ((128 * 1024 // len(DATA)) + 1)
BIG_DATA = CZ.compress(BIG_TEXT)


def test_zlibdecompressor_constructor_and_basic(Z):
    # ZlibDecompressorTest.test_Constructor / testDecompress / testDecompressChunks10 (reference :1031-1058)
    for args in (("bla",), (-15, "bla"), (-15, b"bla", "bla")):
        with pytest.raises(TypeError):
            Z._ZlibDecompressor(*args)
    d = Z._ZlibDecompressor()
    with pytest.raises(TypeError):
        d.decompress()
    assert d.decompress(DATA) == TEXT and d.eof and not d.needs_input
    d = Z._ZlibDecompressor()
    out = b"".join(d.decompress(DATA[i:i + 10]) for i in range(0, len(DATA), 10))
    assert out == TEXT and d.eof


def test_zlibdecompressor_unused_data_and_eof_error(Z):
    # testDecompressUnusedData / testEOFError (reference :1060-1072)
    d = Z._ZlibDecompressor()
    assert d.decompress(DATA + b"this is unused data") == TEXT
    assert d.unused_data == b"this is unused data"
    for late in (b"anything", b""):
        with pytest.raises(EOFError):
            d.decompress(late)


def test_zlibdecompressor_max_length_walk(Z):
    # testDecompressorChunksMaxsize (reference :1090-1117)
    d = Z._ZlibDecompressor()
    cut = len(BIG_DATA) - 64
    out = [d.decompress(BIG_DATA[:cut], max_length=100)]
    assert not d.needs_input and len(out[-1]) == 100
    out.append(d.decompress(b"", max_length=100))
    assert not d.needs_input and len(out[-1]) == 100
    out.append(d.decompress(BIG_DATA[cut:], max_length=100))
    assert len(out[-1]) <= 100
    while not d.eof:
        out.append(d.decompress(b"", max_length=100))
        assert len(out[-1]) <= 100
    assert b"".join(out) == BIG_TEXT and d.unused_data == b""


def test_zlibdecompressor_input_buffering(Z):
    # test_decompressor_inputbuf_1/2/3 (reference :1119-1167)
    for plan in ([(slice(0, 100), 0), (slice(0, 0), 2), (slice(100, 105), 15), (slice(105, None), -1)],
                 [(slice(0, 200), 0), (slice(0, 0), -1), (slice(200, 280), 2), (slice(280, 300), 2),
                  (slice(300, None), -1)],
                 [(slice(0, 200), 5), (slice(200, 300), 5), (slice(300, None), -1)]):
        d = Z._ZlibDecompressor()
        out = []
        for sl, k in plan:
            piece = d.decompress(DATA[sl], k)
            assert k < 0 or len(piece) <= k
            out.append(piece)
        assert b"".join(out) == TEXT


def test_zlibdecompressor_needs_input_protocol(Z):
    d = Z._ZlibDecompressor()
    assert d.needs_input and not d.eof and d.unused_data == b""
    out = bytearray()
    pos = 0
    while not d.eof:
        if d.needs_input:
            assert pos < len(BIG_DATA)
            piece, pos = BIG_DATA[pos:pos + 777], pos + 777
        else:
            piece = b""
        out += d.decompress(piece, 4096)
    assert bytes(out) == BIG_TEXT


def test_zlibdecompressor_failure_is_sticky_enough(Z):
    # ZlibDecompressorTest.test_failure (reference :1169-1173)
    d = Z._ZlibDecompressor()
    for _ in range(2):
        with pytest.raises(Exception):
            d.decompress(b"Not a valid deflate block" * 30)


def test_module_constants(Z):
    for name in ("DEFLATED", "DEF_BUF_SIZE", "DEF_MEM_LEVEL", "MAX_WBITS", "Z_BEST_COMPRESSION", "Z_BEST_SPEED",
                 "Z_BLOCK", "Z_DEFAULT_COMPRESSION", "Z_DEFAULT_STRATEGY", "Z_FILTERED", "Z_FINISH", "Z_FIXED",
                 "Z_FULL_FLUSH", "Z_HUFFMAN_ONLY", "Z_NO_COMPRESSION", "Z_NO_FLUSH", "Z_PARTIAL_FLUSH", "Z_RLE",
                 "Z_SYNC_FLUSH", "Z_TREES"):
        assert getattr(Z, name) == getattr(CZ, name), name
    # VersionTestCase.test_library_version (reference :65-71)
    assert Z.ZLIB_RUNTIME_VERSION[0] == Z.ZLIB_VERSION[0]
    assert Z.ZLIBNG_VERSION == Z.ZLIBNG_RUNTIME_VERSION


# ------------------------------------------------------------------------------- every single-bit corruption of a small stream

@pytest.mark.parametrize("wbits", [15, -15, 31])
def test_every_single_bit_flip_of_small_streams(Z, wbits):
    """Headers, every deflate bit and trailers: what inflate() refuses must be refused (reserved gzip flag bits included),
    what it still accepts must give the same bytes -- one-shot and through an object."""
    for payload in (b"", b"a", b"hello hello hello, hello?", TEXT[:300]):
        co = CZ.compressobj(6, CZ.DEFLATED, wbits)
        z = co.compress(payload) + co.flush()
        differ = []
        for pos in range(len(z)):
            for bit in range(8):
                zz = bytearray(z)
                zz[pos] ^= 1 << bit
                zz = bytes(zz)
                a, b = outcome(lambda m: m.decompress(zz, wbits), CZ), outcome(lambda m: m.decompress(zz, wbits), Z)
                if a != b:
                    differ.append(("one-shot", pos, bit, a, b))

                def through_object(m):
                    do = m.decompressobj(wbits)
                    out = do.decompress(zz)
                    return out + do.flush(), do.eof
                a, b = outcome(through_object, CZ), outcome(through_object, Z)
                if a != b:
                    differ.append(("object", pos, bit, a, b))
        assert not differ, (len(differ), differ[:6])


def test_preset_dictionary_stream_fed_at_every_cut(Z):
    """A zlib stream with a preset dictionary and bytes behind its end, fed to decompressobj in two pieces at every split point
    (the dictionary is asked for in the middle of a call: what follows the dictionary id must not be lost): output, eof,
    unused_data as CPython's zlib gives them."""
    import zlib
    src = bytes(range(256)) * 200
    zdict = src[100:20100]
    for n in (0, 1, 777):
        d = src[30000:30000 + n]
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, 0, zdict)
        blob = co.compress(d) + co.flush() + b"TAILBYTES"
        for cut in range(0, len(blob) + 1, 1 if n < 100 else 7):
            for limit in (0, 1000):
                do, ref = Z.decompressobj(15, zdict), zlib.decompressobj(15, zdict)
                got = want = b""
                for obj, acc in ((do, "got"), (ref, "want")):
                    out = obj.decompress(blob[:cut], limit)
                    out += obj.decompress(obj.unconsumed_tail + blob[cut:], limit)
                    while not obj.eof and obj.unconsumed_tail:
                        out += obj.decompress(obj.unconsumed_tail, 100000)
                    if acc == "got":
                        got = out
                    else:
                        want = out
                assert (got, do.eof, do.unused_data) == (want, ref.eof, ref.unused_data), (n, cut, limit)
