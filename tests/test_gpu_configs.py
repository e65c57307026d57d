"""The configurations BASELINE.json names, each exercised on the GPU against the oracle / the system zlib at a size the
check finishes in seconds (bench.py and bench_configs.py time them at full size):
  config 1  zlib_ng.compress / decompress, level 6, 1 MiB of os.urandom (stored blocks)
  config 2  level 1 on 1 GiB of text, device resident, round trip compared on the device
  config 5  levels 7-9 (deep chains, candidates compared in full) on the Silesia-like mix, byte-exact against the oracle
  and the reference's own >4 GiB case (tests/test_gzip_ng.py:295-317) at its true size."""
import ctypes as C
import io
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B = 131072


def test_config1_urandom_one_mib(ctx):
    """1 MiB of incompressible data, level 6: stored blocks, at most 0.03 % larger; the stdlib reads it and we read the stdlib's."""
    from zlib_ng_amd import zlib_ng
    for buf in (os.urandom(1 << 20), np.random.default_rng(0).bytes(1 << 20)):
        c = zlib_ng.compress(buf, 6)
        assert len(c) <= len(buf) * 1.0003 + 16, len(c)
        assert zlib.decompress(c) == buf
        assert zlib_ng.decompress(c) == buf
        assert zlib_ng.decompress(zlib.compress(buf, 6)) == buf


@pytest.mark.parametrize("level", [7, 8, 9])
def test_config5_deep_levels_on_the_mix(ctx, level):
    """20+ dictionary-chained units (about 3 of every class of corpus.mixed) through the HIP pipeline: the bytes and CRC of every unit
    are the oracle's, the stream inflates with the system zlib, and the ratio is not below zlib's at the same level."""
    from oracle import oracle as O
    from zlib_ng_amd import corpus
    data = corpus.mixed(7 * 3 * B + 7 * 1024, seed=5).tobytes()
    nb = len(data) // B
    assert nb >= 20
    data = data[:nb * B]
    blocks = [(b * B, B, 32768 if b else 0, 0) for b in range(nb)]
    outs, crcs, ovf = ctx.deflate_blocks(data, blocks, level, B + B // 8 + 600)
    assert not ovf
    for b, (out, crc) in enumerate(zip(outs, crcs)):
        lo = b * B
        exp, ecrc = O.deflate_unit(data[lo:lo + B], data[max(0, lo - 32768):lo] if b else b"", level, 0)
        assert out == exp, f"level {level}: unit {b} differs from the oracle"
        assert crc == ecrc == zlib.crc32(data[lo:lo + B])
    stream = b"".join(outs)
    assert zlib.decompressobj(-15).decompress(stream + b"\x03\x00") == data
    ztot = 0
    for b in range(nb):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, 0, data[b * B - 32768:b * B]) if b else zlib.compressobj(level, zlib.DEFLATED, -15)
        ztot += len(co.compress(data[b * B:(b + 1) * B]) + co.flush(zlib.Z_SYNC_FLUSH))
    assert len(stream) <= ztot, (len(stream), ztot)


def test_config2_level1_one_gib_device_resident(ctx):
    """1 GiB of text, level 1, 8192 dictionary-chained blocks: deflate + gather on the device, the whole stream inflated again on
    the device (chunk-parallel over its sync-flush points) and compared there with the input; a prefix through the system zlib."""
    from zlib_ng_amd import _lib, corpus
    L, h = ctx.L, ctx.h
    uniq = 32 << 20
    size = 1 << 30
    host = corpus.text(uniq, seed=2)
    nb = size // B

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert L.zngamd_dmalloc(h, nbytes, C.byref(p)) == 0, ctx.err()
        return p
    d_in = dmalloc(size + 64)
    for r in range(size // uniq):
        assert L.zngamd_h2d(h, C.c_void_p(d_in.value + r * uniq), host.ctypes.data_as(C.c_void_p), uniq) == 0
    assert L.zngamd_h2d(h, C.c_void_p(d_in.value + size), C.cast(C.c_char_p(bytes(64)), C.c_void_p), 64) == 0
    blocks = (_lib.Block * nb)()
    for b in range(nb):
        blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
    d_slots, d_len, d_crc = dmalloc(nb * _lib.SLOT_STRIDE), dmalloc(nb * 4), dmalloc(nb * 4)
    d_comp, d_out = dmalloc(size // 2 + (64 << 20)), dmalloc(size + 64)
    try:
        assert L.zngamd_deflate_blocks_dev(h, d_in, size, blocks, nb, 1, d_slots, d_len, d_crc, None) == 0, ctx.err()
        total = C.c_uint64(0)
        assert L.zngamd_gather_dev(h, d_slots, d_len, nb, d_comp, 0, size // 2 + (64 << 20) - 128, None, C.byref(total)) == 0, ctx.err()
        assert 0 < total.value < size // 2
        tail = b"\x03\x00" + bytes(64)
        assert L.zngamd_h2d(h, C.c_void_p(d_comp.value + total.value), C.cast(C.c_char_p(tail), C.c_void_p), len(tail)) == 0
        olen, used = C.c_uint64(0), C.c_uint64(0)
        ctx.decode_paths(True)
        rc = L.zngamd_inflate_raw_dev(h, d_comp, total.value + 2, d_out, size, C.byref(olen), C.byref(used))
        assert (rc, olen.value, used.value) == (_lib.STREAM_END, size, total.value + 2), ctx.err()
        assert ctx.decode_paths(True)["chunked"] == 1
        bad = C.c_uint64(1)
        assert L.zngamd_compare_dev(h, d_out, d_in, size, C.byref(bad)) == 0 and bad.value == 0
        # CRC-32 of every block as the reference's writer folds them (gzip_ng_threaded.py:394)
        crcs = np.empty(nb, np.uint32)
        assert L.zngamd_d2h(h, crcs.ctypes.data_as(C.c_void_p), d_crc, nb * 4) == 0
        whole = 0
        for b in range(uniq // B):
            whole = ctx.crc32_combine(whole, int(crcs[b]), B)
        assert whole == zlib.crc32(host)
        lens = np.empty(nb, np.uint32)
        assert L.zngamd_d2h(h, lens.ctypes.data_as(C.c_void_p), d_len, nb * 4) == 0
        k = int(lens[:64].sum())
        pref = np.empty(k, np.uint8)
        assert L.zngamd_d2h(h, pref.ctypes.data_as(C.c_void_p), d_comp, k) == 0
        assert zlib.decompressobj(-15).decompress(pref.tobytes()) == host[:64 * B].tobytes()
    finally:
        for p in (d_in, d_slots, d_len, d_crc, d_comp, d_out):
            L.zngamd_dfree(h, p)


def test_member_longer_than_4_gib_true_size():
    """reference tests/test_gzip_ng.py:295-317 (test_decompress_on_long_input) at its true size: 4 GiB of zeros + 123 bytes through
    gzip_ng.open, written and read in 1 MiB pieces; ISIZE has wrapped, lengths beyond 2**32 are in play everywhere."""
    from zlib_ng_amd import gzip_ng
    n = 20
    block_size = 2 ** n
    iterations = 2 ** (32 - n)
    zeros_block = bytes(block_size)
    buffered_stream = io.BytesIO()
    with gzip_ng.open(buffered_stream, "wb") as gz:
        for _ in range(iterations):
            gz.write(zeros_block)
        gz.write(b"\x01" * 123)
    assert buffered_stream.tell() < 64 << 20
    raw = buffered_stream.getvalue()
    assert int.from_bytes(raw[-4:], "little") == 123          # ISIZE = length mod 2**32
    buffered_stream.seek(0)
    with gzip_ng.open(buffered_stream, "rb") as gz:
        for _ in range(iterations):
            assert zeros_block == gz.read(block_size)
        assert gz.read() == b"\x01" * 123
