"""The caller's side of the 128 KiB-call benchmark on the box's CPU, no GPU work: what one thread copies per second in 128 KiB pieces out
of a 1 GiB object (cold source) into a 64 MiB buffer, through the idioms the writer could use, and the whole write() path of
gzip_ng_threaded with the engine call replaced by nothing.  Says what the writer's caller can reach at best."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, gzip_ng_threaded, zlib_ng      # noqa: E402

n, CALL = 1 << 30, 128 * 1024
blob = bytes(os.urandom(1 << 20)) * (n >> 20)
mvb = memoryview(blob)
buf = bytearray(64 << 20)
mv = memoryview(buf)
a = ctypes.c_char.from_buffer(buf); base = ctypes.addressof(a); del a


def t(fn, name):
    best = 9
    for _ in range(3):
        s = time.perf_counter(); fn(); best = min(best, time.perf_counter() - s)
    print(f"{name:60s} {best * 1e3:7.1f} ms  {n / best / 1e9:5.1f} GB/s  {best / (n / CALL) * 1e6:5.1f} us/call", flush=True)


def assign():
    p = 0
    for o in range(0, n, CALL):
        if p + CALL > len(buf):
            p = 0
        mv[p:p + CALL] = mvb[o:o + CALL]; p += CALL


def move():
    p = 0
    for o in range(0, n, CALL):
        if p + CALL > len(buf):
            p = 0
        src, keep = _lib._addr(mvb[o:o + CALL]); ctypes.memmove(base + p, src, CALL); p += CALL


def one_big():
    ctypes.memmove(base, blob, 64 << 20)


t(assign, "memoryview slice assignment, 128 KiB pieces")
t(move, "_addr + ctypes.memmove, 128 KiB pieces")
s = time.perf_counter()
for o in range(0, n, 64 << 20):
    ctypes.memmove(base, ctypes.c_char_p(blob[o:o + 1]) and (ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p).value + o), 64 << 20)
dt = time.perf_counter() - s
print(f"{'memmove of 64 MiB pieces':60s} {dt * 1e3:7.1f} ms  {n / dt / 1e9:5.1f} GB/s")

_lib.contexts = lambda k: [None]


def nothing(ctxs, b, blocks, level, cap, into=None, table=None):
    return memoryview(into)[:len(b) // 3], [0] * len(blocks), False, None


_lib.deflate_blocks_multi = nothing
zlib_ng.crc32_combine = lambda a, b, c: 0


def writer():
    with gzip_ng_threaded.open("/dev/null", "wb", compresslevel=6, threads=8, block_size=CALL) as f:
        for o in range(0, n, CALL):
            f.write(mvb[o:o + CALL])


def raw_writer(drop, piece=None):
    def f():
        w = gzip_ng_threaded._ThreadedGzipWriter("/dev/null", "wb", block_size=CALL, level=6, threads=8)
        orig = w._flush_small
        if drop:
            def fl(wait=True):
                w._small_n = 0
            w._flush_small = fl
        for o in range(0, n, CALL):
            w.write(piece if piece is not None else mvb[o:o + CALL])
        w._flush_small = orig
        w.close()
    return f


t(writer, "gzip_ng_threaded write() path, engine call replaced by nothing")
t(raw_writer(False), "the raw writer alone (no io.BufferedWriter in front)")
t(raw_writer(True), "the raw writer alone, batches dropped")
t(raw_writer(True, mvb[:CALL]), "the raw writer alone, batches dropped, one warm piece")
w = gzip_ng_threaded._ThreadedGzipWriter("/dev/null", "wb", block_size=CALL, level=6, threads=8)
w._flush_small = lambda wait=True: setattr(w, "_small_n", 0)


def reuse():
    for o in range(0, n, CALL):
        w.write(mvb[o:o + CALL])


t(reuse, "one writer object kept (warm buffers), batches dropped")
