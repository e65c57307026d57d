"""Where the time of a 1 MiB one-shot compress goes (BASELINE config 1: os.urandom, and text beside it): wall and kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, zlib_ng
ctx = _lib.default_context()
for name, buf in (("urandom", os.urandom(1 << 20)), ("text", corpus.text(1 << 20, seed=1).tobytes())):
    zlib_ng.compress(buf, 6)
    ctx.profiling(True); ctx.kernel_times(True)
    t = time.perf_counter(); c = zlib_ng.compress(buf, 6); dt = time.perf_counter() - t
    kt = {k: round(v[0], 3) for k, v in ctx.kernel_times(True).items() if v[1]}
    t = time.perf_counter(); d = zlib_ng.decompress(c); dd = time.perf_counter() - t
    kd = {k: round(v[0], 3) for k, v in ctx.kernel_times(True).items() if v[1]}
    ctx.profiling(False)
    assert d == buf
    print(f"{name}: compress {dt*1e3:.2f} ms kernels {kt} | decompress {dd*1e3:.2f} ms kernels {kd}")
