"""gzip_ng_threaded.open(..., "wb") in 128 KiB calls (1 GiB to os.devnull, the reference's write benchmark) with one, two and three
contexts on the one GPU (ZNGAMD_DEVICES=0 / 0,0 / 0,0,0): the engine calls of a batch's ranges run side by side, so one range's
upload and download overlap another's kernels."""
import os, sys, time, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
    import numpy as np
    from zlib_ng_amd import corpus, gzip_ng_threaded, zlib_ng
    blob = bytes(np.tile(corpus.text(64 << 20, 1), 4)); mv = memoryview(blob); CALL = 131072
    def w():
        with gzip_ng_threaded.open(os.devnull, "wb", compresslevel=6, threads=8, block_size=CALL) as f:
            for _ in range(4):
                for o in range(0, len(blob), CALL): f.write(mv[o:o + CALL])
    w()
    best = min((lambda t: (w(), time.perf_counter() - t)[1])(time.perf_counter()) for _ in range(3))
    tc = min((lambda t: (zlib_ng.compress(blob, 6, 31), time.perf_counter() - t)[1])(time.perf_counter()) for _ in range(3))
    print(f"ZNGAMD_DEVICES={os.environ.get('ZNGAMD_DEVICES')}: threaded write {4 * len(blob) / best / 1e9:.2f} GB/s; zlib_ng.compress 256 MiB {len(blob) / tc / 1e9:.2f} GB/s")
else:
    for spec in ("0", "0,0", "0,0,0"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, ZNGAMD_DEVICES=spec))
