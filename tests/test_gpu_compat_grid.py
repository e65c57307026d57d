"""The reference's tests/test_compat.py parameter grid, in full, against CPython's zlib / gzip: every level x every
window-bits value of the three containers x every memLevel on 128 KiB of the reference's FASTQ payload, every strategy,
every power-of-two size from 8 B to 512 KiB at every level for the gzip shortcuts.  One test per reference test; the grid
is walked inside the test (thousands of ids would only slow collection down) and every failing cell is reported."""
import gzip
import itertools
import zlib

import pytest

pytestmark = pytest.mark.gpu

DATA_SIZES = [2 ** i for i in range(3, 20)]                                    # test_compat.py:25
WBITS_RANGE = list(range(9, 16)) + list(range(25, 32)) + list(range(-15, -8))  # test_compat.py:34
STRATEGIES = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED)


def grid():
    """limited_zlib_tests (test_compat.py:47-62): level x wbits x memLevel with the default strategy, then each strategy
    with default settings."""
    for level in range(-1, 10):
        for wbits in WBITS_RANGE:
            for mem in range(1, 10):
                yield 128 * 1024, level, wbits, mem, zlib.Z_DEFAULT_STRATEGY
    for strategy in STRATEGIES:
        yield 128 * 1024, -1, zlib.MAX_WBITS, zlib.DEF_MEM_LEVEL, strategy


@pytest.fixture(scope="module")
def Z():
    from zlib_ng_amd import zlib_ng
    return zlib_ng


@pytest.fixture(scope="module")
def G():
    from zlib_ng_amd import gzip_ng
    return gzip_ng


def _report(bad):
    assert not bad, f"{len(bad)} failing cells, first: {bad[:5]}"


def test_compress_grid(Z, fastq):
    # test_compress (test_compat.py:80-88): sizes x levels x wbits
    bad = []
    for size, level, wbits in itertools.product(DATA_SIZES, range(-1, 10), WBITS_RANGE):
        data = fastq[:size]
        if zlib.decompress(Z.compress(data, level=level, wbits=wbits), wbits) != data:
            bad.append((size, level, wbits))
    _report(bad)


def test_decompress_zlib_and_own_streams(Z, fastq):
    # test_decompress_zlib / test_decompress_zlib_ng (test_compat.py:90-97, :109-117)
    bad = []
    for size, level in itertools.product(DATA_SIZES, range(-1, 10)):
        data = fastq[:size]
        if Z.decompress(zlib.compress(data, level)) != data:
            bad.append(("zlib", size, level))
    for size, level, wbits in itertools.product(DATA_SIZES, (-1, 0, 1, 6, 9), WBITS_RANGE):
        data = fastq[:size]
        if Z.decompress(Z.compress(data, level=level, wbits=wbits), wbits=wbits) != data:
            bad.append(("own", size, level, wbits))
    _report(bad)


def test_decompress_wbits_grid(Z, fastq):
    # test_decompress_wbits (test_compat.py:99-107)
    bad = []
    for size, level, wbits, mem, strategy in grid():
        data = fastq[:size]
        co = zlib.compressobj(level=level, wbits=wbits, memLevel=mem, strategy=strategy)
        z = co.compress(data) + co.flush()
        if Z.decompress(z, wbits=wbits) != data:
            bad.append((level, wbits, mem, strategy))
    _report(bad)


def test_compressobj_grid(Z, fastq):
    # test_compress_compressobj (test_compat.py:119-130)
    bad = []
    for size, level, wbits, mem, strategy in grid():
        data = fastq[:size]
        co = Z.compressobj(level=level, wbits=wbits, memLevel=mem, strategy=strategy)
        z = co.compress(data) + co.flush()
        if zlib.decompress(z, wbits) != data:
            bad.append((level, wbits, mem, strategy))
    _report(bad)


def test_decompressobj_grid(Z, fastq):
    # test_decompress_decompressobj (test_compat.py:132-143)
    bad = []
    for size, level, wbits, mem, strategy in grid():
        data = fastq[:size]
        co = zlib.compressobj(level=level, wbits=wbits, memLevel=mem, strategy=strategy)
        z = co.compress(data) + co.flush()
        do = Z.decompressobj(wbits=wbits)
        out = do.decompress(z) + do.flush()
        if out != data or do.unused_data != b"" or do.unconsumed_tail != b"":
            bad.append((level, wbits, mem, strategy))
    _report(bad)


def test_unconsumed_tail_and_unused_data(Z, fastq):
    # test_decompressobj_unconsumed_tail / test_unused_data (test_compat.py:145-151, :179-189)
    do = Z.decompressobj()
    assert len(do.decompress(zlib.compress(fastq[:128 * 1024]), 2048)) == 2048
    unused = b"abcdefghijklmnopqrstuvwxyz"
    data = b"A meaningful sentence starts with a capital and ends with a."
    for wbits in (-15, 15, 31):
        co = zlib.compressobj(wbits=wbits)
        z = co.compress(data) + co.flush()
        do = Z.decompressobj(wbits=wbits)
        assert do.decompress(z + unused) == data and do.unused_data == unused


def test_gzip_shortcuts_grid(G, fastq):
    # test_gzip_ng_compress / test_decompress_gzip / test_decompress_gzip_ng (test_compat.py:153-177)
    bad = []
    for size, level in itertools.product(DATA_SIZES, range(10)):
        data = fastq[:size]
        own = G.compress(data, compresslevel=level)
        if gzip.decompress(own) != data:
            bad.append(("compress", size, level))
        if G.decompress(gzip.compress(data, compresslevel=level)) != data:
            bad.append(("decompress", size, level))
        if G.decompress(own) != data:
            bad.append(("round trip", size, level))
    _report(bad)
