import os, sys, time, zlib, gzip
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, zlib_ng
ctx = _lib.default_context()
data = gzip.open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz")).read()
for size in (1 << 10, 16 << 10, 64 << 10):
    d = data[:size]; z = zlib.compress(d, 6)
    for _ in range(3): zlib_ng.decompress(z); zlib_ng.compress(d)
    n = 50
    t = time.perf_counter()
    for _ in range(n): zlib_ng.compress(d)
    tc = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n): zlib_ng.decompress(z)
    td = (time.perf_counter() - t) / n
    ctx.profiling(True); ctx.kernel_times(True)
    for _ in range(10): zlib_ng.compress(d)
    kc = ctx.kernel_times(True)
    for _ in range(10): zlib_ng.decompress(z)
    kd = ctx.kernel_times(True); ctx.profiling(False)
    print(f"{size:6d} B: compress {tc*1e6:7.1f} us (kernels {sum(v[0] for v in kc.values())/10*1e3:6.1f} us: { {k: round(v[0]/10*1e3,1) for k,v in kc.items() if v[1]} }), decompress {td*1e6:7.1f} us (kernels {sum(v[0] for v in kd.values())/10*1e3:6.1f} us)")
os.environ["ZNGAMD_TRACE"] = "1"
