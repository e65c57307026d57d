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
    for level in (1, 6, 8, 9):      # (8 and 9: the work-list search, eight and twelve chain steps in visits of four)
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


def test_chain_tables_carried_across_units_equal_the_oracle(monkeypatch):
    """The chain kernel walks RUNS of consecutive units and carries its tables over instead of inserting each unit's 32 KiB
    dictionary again (ZNGAMD_CHAIN_RUN forces short runs; large batches get them by themselves).  The bytes must not depend
    on where runs are cut: every unit against the oracle, which knows no runs -- mixed block sizes (units of 128 KiB and
    shorter, blocks of several units), a block without dictionary in the middle (a run must break there), a short block whose
    successor's dictionary spans two units (no carry), levels 1 / 6 / 9; and the chain links of carried units stage by stage."""
    from oracle import oracle as O
    from zlib_ng_amd import _lib, corpus
    B = 131072
    data = corpus.text(30 * B, seed=21).tobytes()
    sizes = [B, B, B, 3 * B, 100000, B, 20000, B, B, 2 * B + 777, B, B, 50000, B]
    for run in ("3", "5", "1", "64"):
        monkeypatch.setenv("ZNGAMD_CHAIN_RUN", run)
        ctx = _lib.Context(device=0)
        L, h = ctx.L, ctx.h
        blocks_py, off = [], 0
        for i, sz in enumerate(sizes):
            dl = 0 if i in (0, 6) else min(32768, off)          # block 6 starts afresh: no dictionary
            blocks_py.append((off, sz, dl))
            off += sz
        total = off
        d_in = Dev(ctx, total + 64); d_in.put(data[:total] + bytes(64))
        nb = len(blocks_py)
        blocks = (_lib.Block * nb)(*[_lib.Block(o, s, d, 0, 0) for o, s, d in blocks_py])
        nu = L.zngamd_count_units(blocks, nb)
        for level in (1, 6, 8, 9):      # (8 and 9: the work-list search, eight and twelve chain steps in visits of four)
            d_slots, d_len, d_crc = Dev(ctx, nu * _lib.SLOT_STRIDE), Dev(ctx, nu * 4), Dev(ctx, nu * 4)
            ublock = (C.c_uint32 * nu)()
            ctx.debug_keep(level == 6)          # (the link tables survive a call only on request: the token words take their memory)
            assert L.zngamd_deflate_blocks_dev(h, d_in.p, total, blocks, nb, level, d_slots.p, d_len.p, d_crc.p, ublock) == 0, ctx.err()
            lens, crcs = d_len.get(dtype=np.uint32), d_crc.get(dtype=np.uint32)
            slots = d_slots.get()
            u = 0
            for bi, (o, s, d) in enumerate(blocks_py):
                for k in range((s + B - 1) // B):
                    lo, hi = o + k * B, min(o + (k + 1) * B, o + s)
                    dl = min(32768, d + k * B)
                    ref, rcrc, dbg = O.deflate_unit(data[lo:hi], data[lo - dl:lo], level=level, debug=True)
                    got = slots[u * _lib.SLOT_STRIDE:u * _lib.SLOT_STRIDE + int(lens[u])].tobytes()
                    assert got == ref and int(crcs[u]) == rcrc, (run, level, bi, k)
                    if level == 6:                             # the links themselves, dictionary part included
                        links = np.frombuffer(ctx.debug_fetch(0, u, 2 * (dl + hi - lo)), dtype=np.uint16)
                        assert np.array_equal(links, dbg["prevdist"]), (run, bi, k, int(np.argmax(links != dbg["prevdist"])))
                    u += 1
            assert u == nu
            ctx.debug_keep(False)
            for dv in (d_slots, d_len, d_crc):
                dv.free()
        d_in.free()
        del ctx


def test_packed_deflate_equals_slots_and_gather(ctx):
    """zngamd_deflate_blocks_packed_dev (every unit's size known before it is packed, a prefix sum places it, the packer writes at
    any byte address) leaves exactly the stream that zngamd_deflate_blocks_dev + zngamd_gather_dev leave -- blocks of odd sizes,
    stored and fixed blocks, an empty block, a FINAL last block, every level; the system zlib decodes it."""
    from zlib_ng_amd import _lib, corpus
    L, h = ctx.L, ctx.h
    rng = np.random.default_rng(5)
    text = corpus.text(1 << 20, seed=9).tobytes()
    data = (text[:300001] + rng.bytes(70001) + bytes(50000) + text[300001:700003] + b"ab" + rng.bytes(5) + text[700003:900000])
    sizes = [131072, 1, 99999, 131071, 0, 7, 65536, 40000, 131072, 3, 70001, 131072]
    cuts, off = [], 0
    for sz in sizes:
        cuts.append((off, sz)); off += sz
    while off < len(data):
        sz = min(131072, len(data) - off); cuts.append((off, sz)); off += sz
    nb = len(cuts)
    d_in = Dev(ctx, len(data) + 64); d_in.put(data + bytes(64))
    for level in (0, 1, 4, 6, 9):
        for final_last in (False, True):
            blocks = (_lib.Block * nb)()
            for b, (o, n) in enumerate(cuts):
                blocks[b] = _lib.Block(o, n, min(o, 32768), _lib.FLAG_FINAL if (final_last and b == nb - 1) else 0, 0)
            nu = L.zngamd_count_units(blocks, nb)
            d_slots, d_len, d_crc = Dev(ctx, nu * _lib.SLOT_STRIDE), Dev(ctx, nu * 4), Dev(ctx, nu * 4)
            assert L.zngamd_deflate_blocks_dev(h, d_in.p, len(data), blocks, nb, level, d_slots.p, d_len.p, d_crc.p, None) == 0, ctx.err()
            d_dst = Dev(ctx, len(data) + nu * 64 + 64)
            total = C.c_uint64(0)
            assert L.zngamd_gather_dev(h, d_slots.p, d_len.p, nu, d_dst.p, 0, d_dst.n, None, C.byref(total)) == 0, ctx.err()
            want = d_dst.get(total.value).tobytes()
            want_len, want_crc = d_len.get(dtype=np.uint32).copy(), d_crc.get(dtype=np.uint32).copy()
            d_pk, d_len2, d_crc2, d_off = Dev(ctx, len(data) + nu * 64 + 64), Dev(ctx, nu * 4), Dev(ctx, nu * 4), Dev(ctx, nu * 8)
            total2 = C.c_uint64(0)
            assert L.zngamd_deflate_blocks_packed_dev(h, d_in.p, len(data), blocks, nb, level, d_pk.p, d_pk.n, d_len2.p, d_crc2.p, d_off.p,
                                                      C.byref(total2)) == 0, ctx.err()
            assert total2.value == total.value, (level, final_last)
            assert np.array_equal(d_len2.get(dtype=np.uint32), want_len) and np.array_equal(d_crc2.get(dtype=np.uint32), want_crc)
            offs = d_off.get(dtype=np.uint64)
            assert np.array_equal(offs, np.concatenate([[0], np.cumsum(want_len.astype(np.uint64))[:-1]]))
            got = d_pk.get(total2.value).tobytes()
            assert got == want, (level, final_last, next(i for i in range(len(want)) if got[i] != want[i]))
            tail = b"" if final_last else b"\x03\x00"
            assert zlib.decompress(got + tail, -15) == data
            # too small a destination: the size needed, nothing else
            small = C.c_uint64(0)
            assert L.zngamd_deflate_blocks_packed_dev(h, d_in.p, len(data), blocks, nb, level, d_pk.p, total.value - 1, d_len2.p, d_crc2.p, None,
                                                      C.byref(small)) == _lib.BUF_ERROR and small.value == total.value
            for d in (d_slots, d_len, d_crc, d_dst, d_pk, d_len2, d_crc2, d_off):
                d.free()
    d_in.free()
