"""Where the time of the reference's 128 KiB-call benchmark goes on this engine (gzip_ng_threaded.open, threads=8, block_size=128 KiB):
the writer's engine batches and file writes, the caller's waits; the reader's pump (file read, engine call) and the consumer's waits.
Timers are wrapped around the product's methods from outside.  Wall clock.

    python profiles/time_threaded_rw.py [MiB, default 1024] [call KiB, default 128]
"""
import collections
import os
import sys
import threading
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
sys.path.insert(0, ROOT)
from zlib_ng_amd import _lib, gzip_ng_threaded, zlib_ng, corpus      # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
CALL = (int(sys.argv[2]) if len(sys.argv) > 2 else 128) * 1024
uniq = corpus.text(64 << 20, seed=5)
blob = bytes(uniq) * (mib // 64)
n = len(blob)
mvb = memoryview(blob)
_lib.default_context()

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
lock = threading.Lock()


def timed(owner, name, label):
    fn = getattr(owner, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t
            with lock:
                acc[label] += dt
                cnt[label] += 1
    setattr(owner, name, wrap)


W = gzip_ng_threaded._ThreadedGzipWriter
timed(_lib, "deflate_blocks_multi", "writer: engine batch")
timed(W, "_write_later", "writer: file write")
if os.environ.get("TIME_WRITE_CALLS"):
    timed(W, "write", "writer: inside write() itself (all calls)")
    timed(W, "_flush_small", "writer: _flush_small")
    timed(W, "_end_gzip_stream", "writer: _end_gzip_stream")
    timed(W, "stop", "writer: stop")
    timed(W, "_release_buffers", "writer: _release_buffers")
    timed(W, "close", "writer: raw close()")
    timed(W, "flush", "writer: raw flush()")
timed(W, "_join_batch", "writer: caller waits for the batch in front")
timed(W, "_settle_write", "writer: batch waits for the file write in front")
R = zlib_ng._GzipReader
timed(R, "_read_window", "reader pump: file read")
timed(_lib.Context, "gunzip_stream", "reader pump: engine call")
timed(gzip_ng_threaded._ThreadedGzipReader, "_next_piece", "reader consumer: waits for a window")

path = "/tmp/zng_rw.gz"


def report(title, dt):
    print(f"{title}: {dt * 1e3:.1f} ms = {n / dt / 1e9:.2f} GB/s")
    with lock:
        for k in sorted(acc):
            print(f"    {k:55s} {acc[k] * 1e3:8.1f} ms in {cnt[k]} calls")
        acc.clear(); cnt.clear()


for rep in range(3):
    t = time.perf_counter()
    f = gzip_ng_threaded.open(path, "wb", compresslevel=6, threads=8, block_size=CALL)
    t_open = time.perf_counter()
    for o in range(0, n, CALL):
        f.write(mvb[o:o + CALL])
    t_loop = time.perf_counter()
    f.close()
    t_end = time.perf_counter()
    report(f"write {mib} MiB in {CALL >> 10} KiB calls (run {rep}; open {(t_open - t) * 1e3:.1f} ms, the calls {(t_loop - t_open) * 1e3:.1f} ms, close {(t_end - t_loop) * 1e3:.1f} ms)", t_end - t)
if mib <= 1024:
    assert zlib.decompress(open(path, "rb").read(), 31) == blob
for rep in range(3):                # as the reference's own benchmark does it (benchmark_scripts/gzipwrite128kblocks.py): to os.devnull
    t = time.perf_counter()
    with gzip_ng_threaded.open(os.devnull, "wb", compresslevel=6, threads=8, block_size=CALL) as f:
        for o in range(0, n, CALL):
            f.write(mvb[o:o + CALL])
    report(f"write {mib} MiB to os.devnull in {CALL >> 10} KiB calls (run {rep})", time.perf_counter() - t)
for rep in range(3):
    t = time.perf_counter()
    got = 0
    with gzip_ng_threaded.open(path, "rb", threads=8, block_size=CALL) as f:
        while True:
            b = f.read(CALL)
            if not b:
                break
            got += len(b)
    assert got == n
    report(f"read back in {CALL >> 10} KiB calls (run {rep})", time.perf_counter() - t)
for win in (8, 16, 32):
    os.environ["ZNGAMD_READ_WINDOW"] = str(win << 20)
    t = time.perf_counter()
    got = 0
    with gzip_ng_threaded.open(path, "rb", threads=8, block_size=CALL) as f:
        while True:
            b = f.read(CALL)
            if not b:
                break
            got += len(b)
    report(f"read back, windows of {win} MiB", time.perf_counter() - t)
os.remove(path)
