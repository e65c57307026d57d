"""The self-synchronising block decoder (za_par_sweep in csrc/za_inflate.hip) against CPython's zlib on the stream shapes
that steer it: every zlib strategy (fixed codes, Huffman only, run lengths), blocks that end inside a sweep, stored
blocks in between, preset dictionaries, long codes, hardly compressible and extremely compressible data, output limits
that cut a sweep short, truncated input and flipped bits.  What zlib accepts must decode to the same bytes; what zlib
refuses must be refused."""
import os
import random
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Z():
    from zlib_ng_amd import zlib_ng
    return zlib_ng


def _skewed(n, seed):
    """Bytes with a geometric distribution over 200 symbols: Huffman codes up to 15 bits, longer than the decoder's LUT."""
    rng = np.random.default_rng(seed)
    return np.minimum(rng.geometric(0.08, n) - 1, 199).astype(np.uint8).tobytes()


def _kinds(fastq):
    from zlib_ng_amd import corpus
    rnd = random.Random(4)
    text = corpus.text(600_000, seed=8).tobytes()
    return {
        "fastq": fastq[:700_000],
        "text": text,
        "skewed": _skewed(400_000, 1),
        "random": os.urandom(150_000),
        "zeros": bytes(500_000),
        "runs": b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 400) for _ in range(3000)),
        "mixed": text[:100_000] + os.urandom(70_000) + bytes(90_000) + fastq[:120_000] + _skewed(60_000, 2),
    }


def _deflate(data, level=6, wbits=15, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, zdict=None, flush_every=0):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy, *([zdict] if zdict is not None else []))
    if not flush_every:
        return co.compress(data) + co.flush()
    parts = []
    for i in range(0, len(data), flush_every):
        parts.append(co.compress(data[i:i + flush_every]))
        parts.append(co.flush(zlib.Z_FULL_FLUSH if (i // flush_every) % 2 else zlib.Z_SYNC_FLUSH))
    return b"".join(parts) + co.flush()


@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
def test_strategies_levels_and_sizes(Z, fastq, strategy):
    bad = []
    for name, data in _kinds(fastq).items():
        for level in (1, 6, 9):
            for size in (9_000, 70_000, len(data)):
                d = data[:size]
                z = _deflate(d, level, 15, strategy)
                if Z.decompress(z) != d:
                    bad.append((name, level, size))
    assert not bad, bad


def test_small_blocks_and_stored_blocks_between(Z, fastq):
    # memLevel 1 makes zlib close a block every ~1 K symbols (a sweep then holds several ends of block); sync and full
    # flushes put empty stored blocks between them; level 0 pieces are stored blocks proper
    kinds = _kinds(fastq)
    for name in ("fastq", "text", "mixed"):
        d = kinds[name][:400_000]
        for mem in (1, 2, 9):
            assert Z.decompress(_deflate(d, 6, 15, mem=mem)) == d, (name, mem)
        for every in (1_000, 17_000, 130_000):
            assert Z.decompress(_deflate(d, 6, -15, flush_every=every), -15) == d, (name, every)
    d = kinds["text"]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for i in range(0, len(d), 50_000):
        if (i // 50_000) % 3 == 2:                      # a stored stretch: switch the level for it
            parts.append(co.flush(zlib.Z_FULL_FLUSH))
            c0 = zlib.compressobj(0, zlib.DEFLATED, -15)
            raw = c0.compress(d[i:i + 50_000]) + c0.flush(zlib.Z_FULL_FLUSH)
            parts.append(raw)
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
        else:
            parts.append(co.compress(d[i:i + 50_000]))
    parts.append(co.flush())
    assert Z.decompress(b"".join(parts), -15) == d


def test_preset_dictionary_reaches_before_the_stream(Z, fastq):
    d = _kinds(fastq)["text"]
    zdict = d[200_000:232_768]
    body = d[220_000:520_000]                           # starts inside the dictionary's text: early matches reach back into it
    for wbits in (15, -15):
        z = _deflate(body, 9, wbits, zdict=zdict)
        do = Z.decompressobj(wbits, zdict=zdict)
        assert do.decompress(z) + do.flush() == body
        ref = zlib.decompressobj(wbits, zdict=zdict)
        assert ref.decompress(z) == body


def test_output_limits_cut_sweeps(Z, fastq):
    big = _kinds(fastq)["fastq"][:500_000]
    for limit in (1, 255, 256, 4_097, 70_001, 333_333):
        d = big[:2_000] if limit == 1 else big[:40_000] if limit < 4_000 else big       # every call is a launch: small limits on small data
        z = zlib.compress(d, 6)
        do = Z.decompressobj()
        out, tail = [], z
        while not do.eof:
            piece = do.decompress(tail, limit)
            assert len(piece) <= limit
            out.append(piece)
            tail = do.unconsumed_tail
            if not piece and not tail:
                break
        assert b"".join(out) == d, limit
    z = zlib.compress(big, 6)
    for cap_slack in (0, 1, 63, 64):                    # one-shot with a buffer that is exactly large enough, or nearly
        assert Z.decompress(z, 15, len(big) + cap_slack) == big


def test_truncated_input_gives_a_prefix(Z, fastq):
    d = _kinds(fastq)["text"][:300_000]
    z = zlib.compress(d, 6)
    rnd = random.Random(12)
    for cut in sorted(rnd.sample(range(10, len(z) - 1), 25)):
        do, ref = Z.decompressobj(), zlib.decompressobj()
        got, want = do.decompress(z[:cut]), ref.decompress(z[:cut])
        assert d.startswith(got) and not do.eof
        assert len(want) - 600 <= len(got) <= len(want), (cut, len(got), len(want))      # zlib also hands out the symbols of the cut token's predecessors
        with pytest.raises(Z.error, match="incomplete or truncated"):
            Z.decompress(z[:cut])
        # the rest arrives: the stream completes
        assert got + do.decompress(z[cut:]) + do.flush() == d and do.eof


@pytest.mark.parametrize("kind", ["fastq", "text", "skewed", "mixed"])
def test_flipped_bits_agree_with_zlib(Z, fastq, kind):
    d = _kinds(fastq)[kind][:400_000]
    z = bytearray(_deflate(d, 6, -15))
    rnd = random.Random(len(d))
    differ = []
    for _ in range(40):
        pos, bit = rnd.randrange(len(z)), 1 << rnd.randrange(8)
        z[pos] ^= bit
        try:
            want = zlib.decompress(bytes(z), -15)
        except zlib.error:
            want = None
        try:
            got = Z.decompress(bytes(z), -15)
        except Z.error:
            got = None
        if got != want:
            differ.append((pos, bit, None if want is None else len(want), None if got is None else len(got)))
        z[pos] ^= bit
    assert not differ, differ


def test_distance_before_the_start_is_refused(Z):
    # a hand-made fixed-Huffman block: 40 literals 'a'..., then a match of length 3 at distance 4096 (nothing that far back)
    def bits_of(value, n, out):
        for i in range(n):
            out.append((value >> i) & 1)

    def code_of(value, n, out):                         # Huffman codes go most significant bit first
        for i in reversed(range(n)):
            out.append((value >> i) & 1)
    for n_lit, dist_code, dist_extra_bits, dist_extra, ok in ((40, 23, 10, 0, False), (5000, 23, 10, 0, True),
                                                              (5000, 29, 13, 0, False), (30000, 29, 13, 0, True)):       # the last two inside a sweep
        out = []
        bits_of(1, 1, out); bits_of(1, 2, out)          # BFINAL, BTYPE = 01
        for i in range(n_lit):
            code_of(0x30 + 97 + (i % 7), 8, out)        # literal 'a'..'g': 8-bit codes 0x30 + value
        code_of(1, 7, out)                              # length code 257 (length 3): 7-bit code 0000001
        code_of(dist_code, 5, out); bits_of(dist_extra, dist_extra_bits, out)    # distance code 23: 2049..3072 + extra
        code_of(0, 7, out)                              # end of block
        while len(out) % 8:
            out.append(0)
        raw = bytes(sum(b << k for k, b in enumerate(out[i:i + 8])) for i in range(0, len(out), 8))
        try:
            want = zlib.decompress(raw, -15)
        except zlib.error:
            want = None
        assert (want is not None) == ok
        if ok:
            assert Z.decompress(raw, -15) == want
        else:
            with pytest.raises(Z.error):
                Z.decompress(raw, -15)


def test_many_chunks_take_the_small_kernel_size(Z, fastq):
    """From 1 536 chunks on the chunk kernels run with 512-bit sub-sequences and a small LDS footprint
    (ZNGAMD_CHUNKS_SMALL_FROM in csrc/zng_amd.hip); below, with 1 024 bits.  Both sizes must give the same bytes.  zlib -1
    closes a block about every 75 KB of this text, so 144 MiB make about 1 900 chunks."""
    from zlib_ng_amd import _lib, corpus
    ctx = _lib.default_context()
    text = corpus.text(16 << 20, seed=77).tobytes()
    mixed = corpus.mixed(8 << 20, seed=78).tobytes()
    data = text * 4 + mixed + fastq + text * 4 + mixed[:4 << 20]            # 144 MiB
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    z = co.compress(data) + co.flush()
    ctx.decode_paths()                                   # reading the counters clears them
    out = Z.decompress(z, -15, len(data))
    assert ctx.decode_paths()["chunked"] == 1
    assert len(out) == len(data) and out == data
    del out
    for level in (1, 6):                                 # same stream family, fewer chunks: the large size
        small = data[(60 << 20):(84 << 20)]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        z = co.compress(small) + co.flush()
        assert Z.decompress(z, -15) == small


def test_long_stream_goes_through_the_chunk_pipeline_in_batches():
    """A stream longer than one batch of the chunk pipeline (ZNGAMD_CHUNK_BATCH_MIB, 16 MiB here instead of 512) is decoded batch
    by batch, every batch resumed at a block header with the 32 KiB in front of it as its history: a block-parallel writer's
    stream (sync points: one decoding pass) and an ordinary gzip file (bit-level block finder, two passes), both against the
    system zlib.  In a child process: the batch size is read once per process."""
    import subprocess
    import sys
    code = r'''
import gzip, os, sys, zlib
sys.path.insert(0, os.path.join(%r, "python-zlib-ng_amd"))
from zlib_ng_amd import corpus, zlib_ng, _lib
data = corpus.text(96 << 20, seed=11).tobytes()
ours = zlib_ng.compress(data, 6, 31)                    # dict-chained 128 KiB blocks ending in sync flushes
assert len(ours) > (24 << 20)
ctx = _lib.default_context(); ctx.decode_paths(reset=True)
assert zlib_ng.decompress(ours, 31) == data
assert ctx.decode_paths()["chunked"] >= 1
plain = gzip.compress(data[:64 << 20], 6, mtime=0)      # one ordinary member
assert len(plain) > (20 << 20)
assert zlib_ng.decompress(plain, 31) == data[:64 << 20]
bad = bytearray(ours); bad[len(bad) // 2] ^= 0x10         # damage inside a later batch: an error, not a hang or wrong bytes
try:
    out = zlib_ng.decompress(bytes(bad), 31)
    assert out != data
    raise SystemExit("damaged stream decoded without an error")
except zlib_ng.error:
    pass
print("ok")
''' % (ROOT,)
    env = dict(os.environ, ZNGAMD_CHUNK_BATCH_MIB="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-2000:])
