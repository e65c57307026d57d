"""Held-out ratio gate (CPU, oracle): what a level compresses like on data the level table was not built on a generator for.

The reference's level is zlib-ng's (zlib_ngmodule.c:1645 zng_deflateInit2(level, ...), :1742 zng_deflate); its README promises
"other levels better than zlib" (README.rst:129-130).  zlib-ng is absent here, so the bar is the system zlib at the SAME level under
the bench's own protocol -- 128 KiB units, each primed with the previous 32 KiB of input, one sync flush per unit
(gzip_ng_threaded.py:317, zlib_ngmodule.c:1725-1742): compressed size within 2 % of zlib's at levels 1, 6 and 9 on every corpus
(Python sources, two ELF binaries, C headers -- assembled at run time by zlib_ng_amd.corpus.heldout), and the synthetic corpora
the earlier rounds calibrated on still at least as good as zlib.  tests/test_gpu_ratio_heldout.py is the twin through the HIP path.
"""
import zlib

import numpy as np
import pytest

UNIT = 131072
WIN = 32768
TOL = 1.02
# The bars below were set against zlib 1.2.x (1.2.11 in this image).  Another deflate behind the same module name (zlib-ng's compat
# build, a later zlib with other level tables) is another bar, and the corpora are whatever files this box has: the gate is then
# skipped, not failed.  (bench.py reports the same comparison, with the library's version beside it, on every box.)
pytestmark = pytest.mark.skipif(not zlib.ZLIB_RUNTIME_VERSION.startswith("1.2."),
                                reason="ratio bars were set against zlib 1.2.x, this box has " + zlib.ZLIB_RUNTIME_VERSION)


def zlib_units(data, level):
    tot = 0
    for off in range(0, len(data), UNIT):
        zd = data[max(0, off - WIN):off]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, zd) if zd else \
            zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY)
        tot += len(co.compress(data[off:off + UNIT]) + co.flush(zlib.Z_SYNC_FLUSH))
    return tot


def oracle_units(O, data, level, check=True):
    tot = 0
    for off in range(0, len(data), UNIT):
        zd, u = data[max(0, off - WIN):off], data[off:off + UNIT]
        c, crc = O.deflate_unit(u, zd, level)
        tot += len(c)
        if check and off % (5 * UNIT) == 0:          # any inflater reads it
            d = zlib.decompressobj(-15, zdict=zd) if zd else zlib.decompressobj(-15)
            assert d.decompress(c) == u and crc == zlib.crc32(u)
    return tot


@pytest.fixture(scope="module")
def corpora():
    from zlib_ng_amd import corpus
    c = corpus.heldout(4 << 20)
    if not c:
        pytest.skip("no held-out files on this box")
    return c


@pytest.mark.parametrize("level", [1, 6, 9])
def test_heldout_within_two_percent_of_zlib_at_the_same_level(corpora, level):
    from oracle import oracle as O
    for name, data in corpora.items():
        ours, ref = oracle_units(O, data, level), zlib_units(data, level)
        assert ours <= TOL * ref, f"{name} level {level}: {len(data) / ours:.4f} against zlib's {len(data) / ref:.4f}"


def test_levels_do_not_get_worse_upwards(corpora):
    """levels 4, 5, 6 and 7, 8, 9 search supersets of candidates; a small slack covers the parse's estimates"""
    from oracle import oracle as O
    for name, data in corpora.items():
        data = data[:1 << 20]
        sizes = [oracle_units(O, data, lv, check=False) for lv in range(1, 10)]
        for a, b in zip(sizes, sizes[1:]):
            assert b <= a * 1.005, f"{name}: {sizes}"


@pytest.mark.parametrize("level", [1, 6, 9])
def test_synthetic_corpora_still_at_least_zlib(level):
    from oracle import oracle as O
    from zlib_ng_amd import corpus
    for name, gen, seed in (("text", corpus.text, 1), ("fastq", corpus.fastq, 2)):
        data = gen(2 << 20, seed).tobytes()
        ours, ref = oracle_units(O, data, level), zlib_units(data, level)
        assert ours <= ref, f"{name} level {level}: {len(data) / ours:.4f} against zlib's {len(data) / ref:.4f}"
    data = corpus.mixed(2 << 20, 5).tobytes()
    ours, ref = oracle_units(O, data, level), zlib_units(data, level)
    assert ours <= 1.01 * ref, f"mixed level {level}: {len(data) / ours:.4f} against zlib's {len(data) / ref:.4f}"
