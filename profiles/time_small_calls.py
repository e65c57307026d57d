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
    def lap(f, x):
        ts = []
        for _ in range(n):
            t = time.perf_counter(); f(x); ts.append(time.perf_counter() - t)
        slow = max(range(n), key=lambda i: ts[i])
        if ts[slow] > 4 * sorted(ts)[n // 2]: print(f"   (call {slow} of {n} took {ts[slow]*1e6:.0f} us; the three before it {[round(x*1e6) for x in ts[max(0,slow-3):slow]]})")
        ts.sort()
        return sum(ts) / n, ts[n // 2], ts[-1]
    tc, tc_med, tc_max = lap(zlib_ng.compress, d)
    td, td_med, td_max = lap(zlib_ng.decompress, z)
    body = z[2:]
    tr, tr_med, tr_max = lap(lambda b: ctx.inflate_raw(b, 65536), body)
    ctx.profiling(True); ctx.kernel_times(True)
    for _ in range(10): zlib_ng.compress(d)
    kc = ctx.kernel_times(True)
    for _ in range(10): zlib_ng.decompress(z)
    kd = ctx.kernel_times(True); ctx.profiling(False)
    if hasattr(ctx.L, "zngamd_debug_plan_stats"):
        import ctypes as C
        o = (C.c_ulonglong * 16)(); ctx.L.zngamd_debug_plan_stats(o)
        names = ["load", "sort L", "lengths L", "tree D", "canon+costs", "rle", "cl tree", "costs", "header", "codes out"]
        print("   plan kernel, clocks per launch: " + ", ".join(f"{names[i]} {o[i] / max(1, o[15]):.0f}" for i in range(10)))
    print(f"{size:6d} B: compress {tc*1e6:7.1f} us (kernels {sum(v[0] for v in kc.values())/10*1e3:6.1f} us: { {k: round(v[0]/10*1e3,1) for k,v in kc.items() if v[1]} }), decompress {td*1e6:7.1f} us (kernels {sum(v[0] for v in kd.values())/10*1e3:6.1f} us); medians {tc_med*1e6:.1f} / {td_med*1e6:.1f} us, slowest {tc_max*1e6:.0f} / {td_max*1e6:.0f} us; the engine's inflate call alone {tr_med*1e6:.1f} us")
os.environ["ZNGAMD_TRACE"] = "1"
