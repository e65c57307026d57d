"""Where the wall time of the one-shot host-buffer calls goes (256 MiB of text, level 6): the C entry points into a buffer that
exists already against the Python calls that make a fresh result object; kernel time from the engine's events."""
import ctypes as C, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import _lib, corpus, zlib_ng, gzip_ng
n = int(os.environ.get("MIB", "256")) << 20
data = corpus.text(n, seed=1)
ctx = _lib.default_context(); L, h = ctx.L, ctx.h
out = np.zeros(n // 2 + (1 << 20), dtype=np.uint8)
olen, crc, adl = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
L.zngamd_deflate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
def kt():
    k = ctx.kernel_times(True); return {a: round(b[0], 2) for a, b in k.items() if b[1]}
for it in range(3):
    ctx.profiling(True); ctx.kernel_times(True)
    t0 = time.perf_counter()
    r = L.zngamd_deflate_stream(h, data.ctypes.data, n, 6, 15, out.ctypes.data, out.size, C.byref(olen), C.byref(crc), None)
    dt = time.perf_counter() - t0
    print("C zngamd_deflate_stream (existing buffer): rc %d %.1f ms = %.1f GB/s  kernels %s" % (r, dt * 1e3, n / dt / 1e9, kt()))
comp = bytes(out[:olen.value])
b = data.tobytes()
for wb in (-15, 31):
    for it in range(4):
        ctx.kernel_times(True)
        t0 = time.perf_counter(); c2 = zlib_ng.compress(b, 6, wb); dt = time.perf_counter() - t0
        print("zlib_ng.compress(wbits=%d): %.1f ms = %.1f GB/s  kernels %s" % (wb, dt * 1e3, n / dt / 1e9, kt()))
    assert zlib.decompress(c2, wb) == b
c2 = zlib_ng.compress(b, 6, -15)
back = np.zeros(n + 64, dtype=np.uint8)
used = C.c_uint64(0)
for it in range(3):
    ctx.kernel_times(True)
    t0 = time.perf_counter()
    r = L.zngamd_inflate_raw(h, C.cast(C.c_char_p(c2), C.c_void_p), C.c_uint64(len(c2)), None, C.c_uint32(0), C.c_void_p(back.ctypes.data), C.c_uint64(n), C.byref(olen), C.byref(used), C.byref(crc), None)
    dt = time.perf_counter() - t0
    print("C zngamd_inflate_raw (existing buffer): rc %d %.1f ms = %.1f GB/s  kernels %s" % (r, dt * 1e3, n / dt / 1e9, kt()))
for it in range(4):
    ctx.kernel_times(True)
    d2 = None
    t0 = time.perf_counter(); d2 = zlib_ng.decompress(c2, -15); dt = time.perf_counter() - t0
    print("zlib_ng.decompress: %.1f ms = %.1f GB/s  kernels %s" % (dt * 1e3, n / dt / 1e9, kt()))
assert d2 == b
ctx.profiling(False)
