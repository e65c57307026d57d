"""Files made of many small ordinary gzip members (concatenated logs, `cat *.gz`): one-shot and streamed."""
import gzip, io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, gzip_ng
ctx = _lib.default_context()
text = corpus.text(32 << 20, seed=3).tobytes() * 6
for msize, count in ((16 << 10, 2000), (256 << 10, 128), (2 << 20, 16), (8 << 20, 4), (16 << 20, 12)):
    blob = b"".join(gzip.compress(text[i * msize:(i + 1) * msize], 6) for i in range(count))
    want = text[:msize * count]
    gzip_ng.decompress(blob)
    ctx.decode_paths()
    t = time.perf_counter(); out = gzip_ng.decompress(blob); dt = time.perf_counter() - t
    paths = ctx.decode_paths()
    assert out == want
    t = time.perf_counter(); gzip.decompress(blob); dz = time.perf_counter() - t
    t = time.perf_counter()
    with gzip_ng.open(io.BytesIO(blob), "rb") as f:
        n = 0
        while True:
            b = f.read(1 << 20)
            if not b:
                break
            n += len(b)
    dr = time.perf_counter() - t
    print("%5d members of %4d KiB: decompress %.1f ms (%.0f MB/s), reader %.1f ms, system gzip %.1f ms; paths %s" % (
        count, msize >> 10, dt * 1e3, len(want) / dt / 1e6, dr * 1e3, dz * 1e3, paths))
