"""The reference's line benchmarks (benchmark_scripts/gzipwritelines.py, gzipreadlines.py) on this engine: a FASTQ file written line
by line through gzip_ng.open and read back by iterating over its lines; the system gzip module beside it.  Wall clock.

    python profiles/time_lines.py [repeats of the 3.5 MB fixture, default 20]
"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, gzip_ng, gzip_ng_threaded      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with gzip.open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz"), "rb") as f:
    lines = f.read().splitlines(keepends=True) * reps
total = sum(len(x) for x in lines)
_lib.default_context()


def write_lines(opener, dst):
    t = time.perf_counter()
    with opener(dst) as out_gzip:
        for line in lines:
            out_gzip.write(line)
    return time.perf_counter() - t


def read_lines(opener, path):
    t = time.perf_counter()
    n = 0
    with opener(path) as gzip_file:
        for line in gzip_file:
            n += len(line)
    assert n == total, (n, total)
    return time.perf_counter() - t


dst = "/tmp/zng_lines.gz"
for name, wopen, ropen in (
        ("gzip_ng.open (level 6)", lambda p: gzip_ng.open(p, "wb", compresslevel=6), lambda p: gzip_ng.open(p, "rb")),
        ("gzip_ng_threaded.open (level 6, threads=8)", lambda p: gzip_ng_threaded.open(p, "wb", compresslevel=6, threads=8),
         lambda p: gzip_ng_threaded.open(p, "rb", threads=8)),
        ("system gzip module (level 6)", lambda p: gzip.open(p, "wb", compresslevel=6), lambda p: gzip.open(p, "rb"))):
    tw = min(write_lines(wopen, dst) for _ in range(2))
    size = os.path.getsize(dst)
    tr = min(read_lines(ropen, dst) for _ in range(2))
    print("%-48s write %7.1f MB/s, read %7.1f MB/s, ratio %.3f (%d lines, %.0f MB)" %
          (name, total / tw / 1e6, total / tr / 1e6, total / size, len(lines), total / 1e6))
os.remove(dst)
