"""The reference's own streaming benchmarks (benchmark_scripts/gzipwrite128kblocks.py:6-12, gzipread128kblocks.py:5-9) on this
engine: a file is written and read back through the file API in 128 KiB calls -- gzip_ng.open and gzip_ng_threaded.open -- with
the system gzip module beside them.  Wall clock: PCIe, Python call overhead and file I/O included.

    python profiles/time_128k_calls.py [MiB of input, default 1024]
"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, gzip_ng, gzip_ng_threaded      # noqa: E402

CALL = 128 * 1024
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = corpus.text(64 << 20, seed=1).tobytes()
reps = max(1, (mib << 20) // len(base))
total = reps * len(base)
src = "/tmp/zng_in.bin"
with open(src, "wb") as f:
    for _ in range(reps):
        f.write(base)
_lib.default_context()                                               # context creation is not part of the timings


def write_128k(opener, dst):
    t = time.perf_counter()
    with open(src, "rb") as in_file, opener(dst) as out_gzip:
        while True:
            block = in_file.read(CALL)
            if block == b"":
                break
            out_gzip.write(block)
    return time.perf_counter() - t


def read_128k(opener, path):
    t = time.perf_counter()
    n = 0
    with opener(path) as gzip_file:
        while True:
            block = gzip_file.read(CALL)
            if not block:
                break
            n += len(block)
    assert n == total, (n, total)
    return time.perf_counter() - t


rows = [
    ("gzip_ng.open (level 6)", lambda p: gzip_ng.open(p, "wb", compresslevel=6), lambda p: gzip_ng.open(p, "rb")),
    ("gzip_ng_threaded.open (level 6, threads=8, block_size=128 KiB)",
     lambda p: gzip_ng_threaded.open(p, "wb", compresslevel=6, threads=8, block_size=CALL),
     lambda p: gzip_ng_threaded.open(p, "rb", threads=8, block_size=CALL)),
]
for name, wopen, ropen in rows:
    dst = "/tmp/zng_out.gz"
    tw = min(write_128k(wopen, dst) for _ in range(2))
    size = os.path.getsize(dst)
    tr = min(read_128k(ropen, dst) for _ in range(2))
    to_null = min(write_128k(wopen, os.devnull) for _ in range(1))
    print("%-64s write %7.0f MB/s (to /dev/null %7.0f), read %7.0f MB/s, ratio %.3f, %d MiB in 128 KiB calls" %
          (name, total / tw / 1e6, total / to_null / 1e6, total / tr / 1e6, total / size, total >> 20))
    os.remove(dst)
# the system gzip module on a slice (it is some 100x slower)
small = 64 << 20
t = time.perf_counter()
with gzip.open("/tmp/zng_sys.gz", "wb", compresslevel=6) as f:
    for o in range(0, small, CALL):
        f.write(base[o:o + CALL])
tw = time.perf_counter() - t
t = time.perf_counter()
with gzip.open("/tmp/zng_sys.gz", "rb") as f:
    while f.read(CALL):
        pass
tr = time.perf_counter() - t
print("%-64s write %7.0f MB/s, read %7.0f MB/s (64 MiB)" % ("system gzip module (level 6)", small / tw / 1e6, small / tr / 1e6))
os.remove("/tmp/zng_sys.gz")
os.remove(src)
