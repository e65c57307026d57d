"""Latency of one-shot zlib_ng.decompress / compress on small single streams (the sequential wavefront decoder's
territory), with the engine's kernel times beside the wall time."""
import gzip, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, zlib_ng
ctx = _lib.default_context()
data = gzip.open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz")).read()
for size in (1 << 10, 16 << 10, 128 << 10, 1 << 20, 3 << 20):
    d = data[:size]
    z = zlib.compress(d, 6)
    zlib_ng.decompress(z); zlib_ng.compress(d)
    ctx.profiling(True); ctx.kernel_times(True)
    n = 5
    t = time.perf_counter()
    for _ in range(n):
        out = zlib_ng.decompress(z)
    dt = (time.perf_counter() - t) / n
    kt = ctx.kernel_times(True)
    assert out == d
    t = time.perf_counter()
    for _ in range(n):
        zlib.decompress(z)
    dz = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        zlib_ng.compress(d)
    dc = (time.perf_counter() - t) / n
    ctx.profiling(False)
    print("%8d B: decompress %.2f ms (%.1f MB/s; system zlib %.3f ms), compress %.2f ms; kernel ms per call: %s" % (
        size, dt * 1e3, size / dt / 1e6, dz * 1e3, dc * 1e3, {k: round(v[0] / n, 2) for k, v in kt.items() if v[1]}))
