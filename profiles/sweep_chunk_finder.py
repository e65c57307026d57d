"""Chunk-parallel inflate of ONE deflate stream: block boundaries from sync markers alone against sync markers + the bit-level header
finder (ZNGAMD_DENSE_MIN = sync hits from which the finder is skipped), and the chunk size (ZNGAMD_CHUNK_DIV: chunks of at least
1 / DIV of the output).  Each setting runs in a child process (the knobs are read once).  Wall clock and kernel classes.

    python profiles/sweep_chunk_finder.py            # parent: all settings
"""
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import io
    from zlib_ng_amd import _lib, gzip_ng, gzip_ng_threaded, zlib_ng, corpus
    ctx = _lib.default_context()
    ctx.profiling(True)
    uniq = bytes(corpus.text(64 << 20, seed=5))
    cases = []
    for mib in [int(x) for x in os.environ.get("SWEEP_MIB", "64,256,1024").split(",")]:
        blob = uniq * (mib // 64)
        bio = io.BytesIO()
        with gzip_ng_threaded.open(bio, "wb", compresslevel=6, threads=8, block_size=128 * 1024) as f:
            f.write(blob)
        cases.append((f"threaded writer, 128 KiB blocks, {mib} MiB", bio.getvalue(), len(blob)))
        if mib == 256:
            cases.append((f"one-shot compress (gzip), {mib} MiB", zlib_ng.compress(blob, 6, 31), len(blob)))
    z = zlib.compressobj(6, zlib.DEFLATED, 31)
    cases.append(("zlib level 6, 128 MiB", z.compress(uniq * 2) + z.flush(), 128 << 20))
    for name, gz, n in cases:
        best = 1e9
        for _ in range(3):
            ctx.kernel_times(True)
            t = time.perf_counter(); out = gzip_ng.decompress(gz); dt = time.perf_counter() - t
            kt = ctx.kernel_times(True)
            if dt < best:
                best, bk = dt, kt
        assert len(out) == n
        print(f"    {name:48s} {best * 1e3:8.1f} ms  scan {bk['scan'][0]:6.2f}  inflate {bk['inflate'][0]:6.2f} ms  paths {ctx.decode_paths()}")
        del out
    sys.exit(0)

for dense_min, div in ((8, 4096), (1 << 30, 4096), (1 << 30, 8192), (1 << 30, 16384), (4096, 8192), (8, 16384)):
    env = dict(os.environ, ZNGAMD_DENSE_MIN=str(dense_min), ZNGAMD_CHUNK_DIV=str(div))
    print(f"ZNGAMD_DENSE_MIN={dense_min} ZNGAMD_CHUNK_DIV={div}", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
