"""One-off fuzz of the windowed readers on streams of block-parallel writers (this engine's threaded writer with random block sizes,
sync-flushed zlib streams): random read windows (64 KiB - 8 MiB, so that members are continued across many windows, with the
one-pass and the two-pass chunk decode, with and without history) and random read sizes, through gzip_ng.open and
gzip_ng_threaded.open; every byte compared.     python profiles/fuzz_reader_windows.py [seed] [cases]"""
import io, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import corpus, gzip_ng, gzip_ng_threaded
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
srcs = [corpus.text(24 << 20, seed=1).tobytes(), corpus.mixed(16 << 20, seed=5).tobytes(), corpus.fastq(12 << 20, seed=2).tobytes()]
bad = 0
for case in range(N):
    src = srcs[int(rng.integers(0, len(srcs)))]
    n = int(rng.integers(1 << 20, len(src)))
    data = src[:n]
    kind = int(rng.integers(0, 3))
    if kind == 0:                                            # this engine's threaded writer, random block size
        bs = int(rng.choice([8 << 10, 20000, 64 << 10, 128 << 10, 1 << 20]))
        bio = io.BytesIO()
        with gzip_ng_threaded.open(bio, "wb", compresslevel=int(rng.integers(1, 10)), threads=4, block_size=bs) as f:
            f.write(data)
        blob = bio.getvalue()
    elif kind == 1:                                          # zlib with sync flushes every `step` bytes
        step = int(rng.choice([30000, 100000, 400000]))
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, 31)
        parts = []
        for o in range(0, n, step):
            parts.append(co.compress(data[o:o + step])); parts.append(co.flush(zlib.Z_SYNC_FLUSH))
        blob = b"".join(parts) + co.flush()
    else:                                                    # several members of both kinds
        cut = n // 3
        bio = io.BytesIO()
        with gzip_ng_threaded.open(bio, "wb", compresslevel=6, threads=4, block_size=128 << 10) as f:
            f.write(data[:cut])
        import gzip as _gz
        blob = bio.getvalue() + _gz.compress(data[cut:2 * cut], 6) + _gz.compress(data[2 * cut:], 1)
    os.environ["ZNGAMD_READ_WINDOW"] = str(int(rng.choice([1 << 16, 300000, 1 << 20, 3 << 20, 8 << 20])))
    piece = int(rng.choice([1000, 65536, 131072, 1 << 20, 5 << 20]))
    for opener in (gzip_ng.open, lambda b, m: gzip_ng_threaded.open(b, m, threads=4)):
        got = bytearray()
        with opener(io.BytesIO(blob), "rb") as f:
            while True:
                b = f.read(piece)
                if not b:
                    break
                got += b
        if bytes(got) != data:
            bad += 1
            print("MISMATCH case", case, "kind", kind, "window", os.environ["ZNGAMD_READ_WINDOW"], "piece", piece, len(got), len(data))
print("cases", N, "mismatches", bad)
