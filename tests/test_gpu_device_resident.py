"""Device-resident entry points of the C ABI (what bench.py drives): deflate_blocks_dev + gather_dev, gzip_members_dev,
gzip_scan_dev + gzip_inflate_members_dev, crc32_dev -- device memory through the ABI's own dmalloc / h2d / d2h."""
import ctypes as C
import gzip
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Dev:
    def __init__(self, ctx, nbytes):
        self.ctx, self.n = ctx, nbytes
        self.p = C.c_void_p()
        assert ctx.L.zngamd_dmalloc(ctx.h, max(nbytes, 1), C.byref(self.p)) == 0

    def put(self, data):
        buf = bytes(data)
        assert self.ctx.L.zngamd_h2d(self.ctx.h, self.p, C.cast(C.c_char_p(buf), C.c_void_p), len(buf)) == 0

    def get(self, nbytes=None, dtype=np.uint8):
        nbytes = self.n if nbytes is None else nbytes
        out = np.empty(nbytes, np.uint8)
        assert self.ctx.L.zngamd_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.p, nbytes) == 0
        return out.view(dtype)

    def free(self):
        self.ctx.L.zngamd_dfree(self.ctx.h, self.p)


def test_device_resident_round_trip(ctx):
    from oracle import oracle as O
    from zlib_ng_amd import _lib, corpus
    L, h = ctx.L, ctx.h
    B = 131072
    data = corpus.text(40 * B + 12345, seed=8).tobytes()           # 41 blocks, the last one short
    nb = (len(data) + B - 1) // B
    d_in = Dev(ctx, len(data) + 64); d_in.put(data + bytes(64))
    for level in (1, 6, 9):
        blocks = (_lib.Block * nb)()
        for b in range(nb):
            blocks[b] = _lib.Block(b * B, min(B, len(data) - b * B), 32768 if b else 0, 0, 0)
        nu = L.zngamd_count_units(blocks, nb)
        assert nu == nb
        d_slots, d_len, d_crc = Dev(ctx, nu * _lib.SLOT_STRIDE), Dev(ctx, nu * 4), Dev(ctx, nu * 4)
        assert L.zngamd_deflate_blocks_dev(h, d_in.p, len(data), blocks, nb, level, d_slots.p, d_len.p, d_crc.p, None) == 0, ctx.err()
        d_dst = Dev(ctx, len(data) + nb * 64)
        total = C.c_uint64(0)
        assert L.zngamd_gather_dev(h, d_slots.p, d_len.p, nu, d_dst.p, 0, d_dst.n, None, C.byref(total)) == 0, ctx.err()
        lens, crcs = d_len.get(dtype=np.uint32), d_crc.get(dtype=np.uint32)
        assert int(lens.sum()) == total.value
        stream = d_dst.get(total.value).tobytes()
        assert zlib.decompressobj(-15).decompress(stream) == data         # one dictionary-chained raw stream
        pos = 0
        for b in range(nb):                                                # every unit: oracle's bytes and CRC
            lo, hi = b * B, min((b + 1) * B, len(data))
            ref, rcrc = O.deflate_unit(data[lo:hi], data[max(0, lo - 32768):lo] if b else b"", level=level)
            assert stream[pos:pos + int(lens[b])] == ref and int(crcs[b]) == rcrc == zlib.crc32(data[lo:hi]), (level, b)
            pos += int(lens[b])
        # crc32 of a device buffer
        c = C.c_uint32(0)
        assert L.zngamd_crc32_dev(h, 0, d_in.p, len(data), C.byref(c)) == 0 and c.value == zlib.crc32(data)
        # indexed members, device resident; then the two-pass reader
        d_ms = Dev(ctx, len(data) + nb * 400 + 4096)
        ml, mn = C.c_uint64(0), C.c_uint32(0)
        assert L.zngamd_gzip_members_dev(h, d_in.p, len(data), B, level, d_ms.p, d_ms.n - 64, C.byref(ml), C.byref(mn)) == 0, ctx.err()
        assert mn.value == nb
        members = d_ms.get(ml.value).tobytes()
        assert gzip.decompress(members) == data                            # any gzip reader reads them
        d_tab, d_stat, d_out = Dev(ctx, nb * C.sizeof(_lib.Member)), Dev(ctx, nb * 4), Dev(ctx, len(data) + 64)
        nm, tot = C.c_uint32(0), C.c_uint64(0)
        assert L.zngamd_gzip_scan_dev(h, d_ms.p, ml.value, d_tab.p, nb, C.byref(nm), C.byref(tot)) == 0, ctx.err()
        assert (nm.value, tot.value) == (nb, len(data))
        assert L.zngamd_gzip_inflate_members_dev(h, d_ms.p, ml.value, d_tab.p, nm.value, d_out.p, len(data), d_stat.p) == 0
        assert not d_stat.get(dtype=np.int32).any()
        assert d_out.get(len(data)).tobytes() == data
        # a damaged member is reported in its own status slot, the others decode
        bad = bytearray(members)
        bad[len(bad) // 2] ^= 0x20
        d_ms.put(bytes(bad))
        if L.zngamd_gzip_scan_dev(h, d_ms.p, ml.value, d_tab.p, nb, C.byref(nm), C.byref(tot)) == 0:
            assert L.zngamd_gzip_inflate_members_dev(h, d_ms.p, ml.value, d_tab.p, nm.value, d_out.p, len(data), d_stat.p) == 0
            st = d_stat.get(dtype=np.int32)
            assert np.count_nonzero(st) == 1
        for d in (d_slots, d_len, d_crc, d_dst, d_ms, d_tab, d_stat, d_out):
            d.free()
    d_in.free()


def test_gunzip_partial_and_stream_entry_points(ctx):
    """zngamd_gunzip_partial (member-granular window) and zngamd_gunzip_stream (stateful) called directly."""
    from zlib_ng_amd import _lib, corpus
    text = corpus.text(3 << 20, seed=5).tobytes()
    a, b, c = gzip.compress(text[:1 << 20], 6), gzip.compress(text[1 << 20:2 << 20], 1), gzip.compress(text[2 << 20:], 9)
    blob = a + b + c
    # a window that ends inside the third member: two members come out, the third is left alone
    cut = len(a) + len(b) + len(c) // 2
    code, out, nm, used = ctx.gunzip_partial(blob[:cut], 4 << 20)
    assert (code, nm, used) == (0, 2, len(a) + len(b)) and out == text[:2 << 20]
    code, out, nm, used = ctx.gunzip_partial(blob[used:], 4 << 20)
    assert (code, nm, used) == (0, 1, len(c)) and out == text[2 << 20:]
    # a window that holds no complete member
    code, out, nm, used = ctx.gunzip_partial(blob[:len(a) // 2], 4 << 20)
    assert (code, nm, used, out) == (0, 0, 0, b"")
    # the stateful form walks through one member in 200 KB windows
    st = _lib.GzState()
    pos, got = 0, bytearray()
    big = gzip.compress(text, 6)
    win = 200000
    while pos < len(big):
        last = pos + win >= len(big)
        code, out, nm, used = ctx.gunzip_stream(st, big[pos:pos + win], 8 << 20, last)
        assert code == 0, (code, pos)
        got += out
        if used == 0:
            win *= 2
            continue
        pos += used
    assert bytes(got) == text and st.in_member == 0


def test_caller_stream_and_sync(ctx):
    """zngamd_set_stream(NULL) restores the context's own stream; zngamd_sync waits for it."""
    assert ctx.L.zngamd_set_stream(ctx.h, None) == 0
    assert ctx.crc32(b"penguin") == zlib.crc32(b"penguin")
    assert ctx.L.zngamd_sync(ctx.h) == 0
    assert ctx.L.zngamd_level_ok(6) == 1 and ctx.L.zngamd_level_ok(10) == 0 and ctx.L.zngamd_level_ok(-1) == 1
