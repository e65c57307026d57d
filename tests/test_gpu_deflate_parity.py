"""GPU parity: the HIP deflate pipeline against the oracle, stage by stage and byte for byte.
All compute calls go through the C ABI (libzng_amd.so)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEG = 2048


def _inputs(fastq):
    rng = np.random.default_rng(7)
    zeros = bytes(131072)
    rnd = rng.bytes(131072)
    mixed = fastq[:40000] + rng.bytes(30000) + bytes(20000) + fastq[50000:91072]
    period = (b"abcdefghijklmnopqrstuvwxyz0123456789" * 4000)[:131072]
    return {
        "fastq128k": fastq[:131072], "fastq_tail": fastq[131072:131072 + 100001],
        "zeros": zeros, "random": rnd, "mixed": mixed, "period36": period,
        "tiny5": b"hello", "len1": b"x", "len3": b"abc", "len4": b"abcd", "seg_edge": fastq[:2049],
        "seg_exact": fastq[:4096], "empty": b"",
        # small units take smaller segments (32 bytes .. 1 KiB: oracle.h za_o_seg_shift)
        "s33": fastq[:33], "s700": fastq[100:800], "s3000": fastq[:3000], "s10000": mixed[35000:45000], "s40000": fastq[7:40007],
        "s65536": fastq[:65536], "s65537": fastq[:65537],
    }


def _stage_compare(ctx, O, data, zdict, level, flags):
    from zlib_ng_amd import _lib
    buf = zdict + data
    exp, exp_crc, dbg = O.deflate_unit(data, zdict, level, flags, debug=True)
    ctx.debug_keep(True)
    got, crcs, ovf = ctx.deflate_blocks(buf, [(len(zdict), len(data), len(zdict), flags)], level, len(data) + 1024)
    ctx.debug_keep(False)
    n, dl = len(data), len(zdict)
    lv = 6 if level == -1 else level
    if n and lv > 0:
        prev = np.frombuffer(ctx.debug_fetch(0, 0, 2 * (dl + n)), np.uint16)
        assert np.array_equal(prev, dbg["prevdist"]), f"stage1 chains differ at {np.flatnonzero(prev != dbg['prevdist'])[:5]}"
        for what, key in ((9, "linkB"), (10, "linkC")) if lv >= 5 else ((9, "linkB"),):
            lk = np.frombuffer(ctx.debug_fetch(what, 0, 2 * (dl + n)), np.uint16)
            assert np.array_equal(lk, dbg[key]), f"stage1 {key} differs at {np.flatnonzero(lk != dbg[key])[:5]}"
        if lv >= 4:      # the dynamic programme rewrites the entries: the search's own results are the kept copy
            pre = np.frombuffer(ctx.debug_fetch(11, 0, 4 * n), np.uint32)
            assert np.array_equal(pre, dbg["best"]), f"stage2 search differs at {np.flatnonzero(pre != dbg['best'])[:5]}"
            cost = np.frombuffer(ctx.debug_fetch(12, 0, 4 * 258), np.uint32)
            assert np.array_equal(cost, dbg["dp_cost"]), f"stage3a cost table differs at {np.flatnonzero(cost != dbg['dp_cost'])[:5]}"
        best = np.frombuffer(ctx.debug_fetch(1, 0, 4 * n), np.uint32)
        assert np.array_equal(best, dbg["best_dp"]), f"stage2/3a entries differ at {np.flatnonzero(best != dbg['best_dp'])[:5]}"
        SEG = 1 << O.seg_shift(n, flags)              # the unit's segment size: 2 KiB for full units and indexed members, less for small units
        nseg = (n + SEG - 1) // SEG
        sn = np.frombuffer(ctx.debug_fetch(3, 0, 4 * 64), np.uint32)
        assert np.array_equal(sn, dbg["seg_ntok"]), "stage3 token counts differ"
        tok = np.frombuffer(ctx.debug_fetch(2, 0, 4 * nseg * SEG), np.uint32)
        for s in range(nseg):
            k = int(sn[s])
            assert np.array_equal(tok[s * SEG:s * SEG + k], dbg["tokens"][s * SEG:s * SEG + k]), f"stage3 tokens differ in segment {s}"
        hist = np.frombuffer(ctx.debug_fetch(4, 0, 4 * 320), np.uint32)
        assert np.array_equal(hist, dbg["hist"]), "stage3 histogram differs"
        plan = np.frombuffer(ctx.debug_fetch(7, 0, 16), np.uint32)
        assert int(plan[0]) == dbg["btype"], f"stage4 block type {plan[0]} != {dbg['btype']}"
        if dbg["btype"] != 0:
            codes = np.frombuffer(ctx.debug_fetch(5, 0, 4 * 320), np.uint32)
            assert np.array_equal((codes >> 16).astype(np.uint8), dbg["lens"]), "stage4 code lengths differ"
            sb = np.frombuffer(ctx.debug_fetch(6, 0, 4 * 65), np.uint32)
            assert np.array_equal(sb[:nseg], dbg["seg_bits"][:nseg]), "stage5 segment bit offsets differ"
            assert sb[64] == dbg["seg_bits"][nseg]
            nchunk = nseg                             # (the index's grain is the segment)
            ci = np.frombuffer(ctx.debug_fetch(8, 0, 4 * (nchunk + 1)), np.uint32)
            assert np.array_equal(ci, dbg["chunk_idx"][:nchunk + 1]), \
                f"stage5 chunk index differs at {np.flatnonzero(ci != dbg['chunk_idx'][:nchunk + 1])[:5]}"
    assert crcs[0] == exp_crc == zlib.crc32(data)
    assert not ovf and got[0] == exp, "final bytes differ"
    return got[0]


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5, 6, 7, 8, 9, 0])
def test_stagewise_parity(ctx, fastq, level):
    """Every level of the table (one, two and three unrolled chain steps, with and without table C and the dynamic programme; the
    work-list search of levels 7-9 with four, eight and twelve steps in visits of four), every stage against the oracle."""
    from oracle import oracle as O
    for name, data in _inputs(fastq).items():
        if level >= 7 and name not in ("fastq128k", "fastq_tail", "zeros", "period36", "tiny5", "empty", "s700", "s10000", "s40000"):
            continue
        c = _stage_compare(ctx, O, data, b"", level, 0)
        d = zlib.decompressobj(-15)
        assert d.decompress(c) == data, name


def test_flat_header_form(ctx, fastq):
    """Units of indexed members carry the dynamic header in its flat form (flag 2): same stages, any inflater reads it."""
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    for name, data in _inputs(fastq).items():
        for flags in (2, 3):
            c = _stage_compare(ctx, O, data, b"", 6, flags)
            d = zlib.decompressobj(-15)
            assert d.decompress(c) == data and d.eof == bool(flags & 1), name
    blk0, blk1 = fastq[:131072], fastq[131072:262144]
    c = _stage_compare(ctx, O, blk1, blk0[-32768:], 1, 2)
    assert zlib.decompressobj(-15, zdict=blk0[-32768:]).decompress(c) == blk1


def test_dictionary_and_final(ctx, fastq):
    from oracle import oracle as O
    blk0, blk1 = fastq[:131072], fastq[131072:262144]
    for zd in (blk0[-32768:], blk0[-100:], blk0[-32767:]):
        c = _stage_compare(ctx, O, blk1, zd, 6, 0)
        d = zlib.decompressobj(-15, zdict=zd)
        assert d.decompress(c) == blk1
    c = _stage_compare(ctx, O, blk1, b"", 6, 1)
    d = zlib.decompressobj(-15)
    assert d.decompress(c) == blk1 and d.eof
    assert _stage_compare(ctx, O, b"", b"", 6, 1) == b"\x03\x00"


def test_multi_block_batch_and_large_block(ctx, fastq):
    """Blocks larger than one unit are cut into dictionary-chained units; many blocks per call."""
    from oracle import oracle as O
    data = fastq[:1000000]
    bs = 300000
    blocks = []
    for off in range(0, len(data), bs):
        ln = min(bs, len(data) - off)
        blocks.append((off, ln, min(32768, off), 0))
    outs, crcs, ovf = ctx.deflate_blocks(data, blocks, 6, bs + bs // 10 + 500)
    assert not ovf
    stream = b"".join(outs) + b"\x03\x00"
    assert zlib.decompress(stream, -15) == data
    for (off, ln, dl, fl), crc, out in zip(blocks, crcs, outs):
        assert crc == zlib.crc32(data[off:off + ln])
        exp = b""
        for uo in range(0, ln, 131072):
            ul = min(131072, ln - uo)
            a = off + uo
            d = min(32768, dl + uo)
            exp += O.deflate_unit(data[a:a + ul], data[a - d:a], 6, 0)[0]
        assert out == exp


def test_one_shot_stream(ctx, fastq):
    from oracle import oracle as O
    for n in (0, 1, 5000, 131072, 131073, 400000):
        raw, crc, ad = ctx.deflate_stream(fastq[:n], 6)
        assert raw == O.deflate_stream(fastq[:n], 6)
        assert zlib.decompress(raw, -15) == fastq[:n]
        assert crc == zlib.crc32(fastq[:n]) and ad == zlib.adler32(fastq[:n])


def test_overflow_reported(ctx):
    rnd = np.random.default_rng(1).bytes(65536)
    outs, crcs, ovf = ctx.deflate_blocks(rnd, [(0, 65536, 0, 0)], 3, 8192 + 819)
    assert ovf and outs[0] is None


def test_checksums(ctx, fastq):
    for n in (0, 1, 7, 2048, 2049, 131072, 131073, 1000003):
        for seed in (0, 1, 0xFFFFFFFF, 123456789):
            assert ctx.crc32(fastq[:n], seed) == zlib.crc32(fastq[:n], seed)
            assert ctx.adler32(fastq[:n], seed) == zlib.adler32(fastq[:n], seed)
    assert ctx.crc32_combine(zlib.crc32(b"abc"), zlib.crc32(b"defgh"), 5) == zlib.crc32(b"abcdefgh")
