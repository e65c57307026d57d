"""Bounded-memory reading of a large ordinary gzip file through gzip_ng.open (64 MiB read windows, one giant member)."""
import gzip, os, resource, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, gzip_ng
ctx = _lib.default_context()
base = corpus.text(64 << 20, seed=1).tobytes()
path = "/tmp/big_single_member.gz"
t = time.perf_counter()
co = zlib.compressobj(1, zlib.DEFLATED, 31)
with open(path, "wb") as f:
    for _ in range(16):                               # 1 GiB, one member, system zlib level 1
        f.write(co.compress(base))
    f.write(co.flush())
print("built %d MiB gzip file in %.1f s" % (os.path.getsize(path) >> 20, time.perf_counter() - t))
for window in (64 << 20, 16 << 20):
    os.environ["ZNGAMD_READ_WINDOW"] = str(window)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t = time.perf_counter()
    n, crc = 0, 0
    with gzip_ng.open(path, "rb") as f:
        while True:
            piece = f.read(32 << 20)
            if not piece:
                break
            n += len(piece)
            crc = zlib.crc32(piece[:4096], crc)
    dt = time.perf_counter() - t
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print("window %3d MiB: %d MiB read in %.2f s = %.0f MB/s wall (PCIe + host copies included); max RSS %d -> %d MiB; paths %s" % (
        window >> 20, n >> 20, dt, n / dt / 1e6, rss0 >> 10, rss1 >> 10, ctx.decode_paths(True)))
os.environ["ZNGAMD_READ_WINDOW"] = str(64 << 20)
from zlib_ng_amd import gzip_ng_threaded
for piece_size in (32 << 20, 1 << 20, 128 << 10):
    t = time.perf_counter()
    n = 0
    with gzip_ng_threaded.open(path, "rb", threads=1) as f:
        while True:
            piece = f.read(piece_size)
            if not piece:
                break
            n += len(piece)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    m = 0
    with gzip_ng.open(path, "rb") as f:
        while True:
            piece = f.read(piece_size)
            if not piece:
                break
            m += len(piece)
    dp = time.perf_counter() - t
    print("reads of %5d KiB: gzip_ng_threaded.open %.0f MB/s, gzip_ng.open %.0f MB/s" % (piece_size >> 10, n / dt / 1e6, m / dp / 1e6))
t = time.perf_counter()
n = 0
with gzip.open(path, "rb") as f:
    while True:
        piece = f.read(32 << 20)
        if not piece:
            break
        n += len(piece)
print("system gzip module: %.0f MB/s" % (n / (time.perf_counter() - t) / 1e6))
os.remove(path)
