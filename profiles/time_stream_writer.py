"""Wall-clock throughput of the writers on 1 GiB of text (host buffers, PCIe and Python plumbing included)."""
import gzip, io, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, gzip_ng, gzip_ng_threaded
ctx = _lib.default_context()
base = corpus.text(64 << 20, seed=1).tobytes()
total = 16 * len(base)
for name, opener in (("gzip_ng_threaded.open(threads=8, block_size=128 KiB)", lambda p: gzip_ng_threaded.open(p, "wb", compresslevel=6, threads=8, block_size=128 * 1024)),
                     ("gzip_ng.open", lambda p: gzip_ng.open(p, "wb", compresslevel=6))):
    path = "/tmp/w.gz"
    for rep in range(2):
        t = time.perf_counter()
        with opener(path) as f:
            for _ in range(16):
                f.write(base)
        dt = time.perf_counter() - t
    size = os.path.getsize(path)
    t = time.perf_counter()
    n = 0
    with gzip_ng.open(path, "rb") as f:
        while True:
            piece = f.read(32 << 20)
            if not piece:
                break
            n += len(piece)
    dr = time.perf_counter() - t
    assert n == total
    print("%s: write %.0f MB/s (ratio %.3f), read back %.0f MB/s" % (name, total / dt / 1e6, total / size, n / dr / 1e6))
    os.remove(path)
t = time.perf_counter()
with gzip.open("/tmp/w.gz", "wb", compresslevel=6) as f:
    f.write(base)
print("system gzip module level 6: %.0f MB/s" % (len(base) / (time.perf_counter() - t) / 1e6))
os.remove("/tmp/w.gz")
