#!/usr/bin/env python3
"""bench_configs.py -- the other BASELINE.json configurations (bench.py is the headline metric).

  config 0  zlib_ng.compress / decompress level 6 on a 1 MiB os.urandom buffer (API plumbing, PCIe included)
  config 1  single-GPU deflate level 1 (and 6) on 1 GiB synthetic text, 128 KiB blocks (device resident)
  config 2  single-GPU inflate of a 4 GiB multi-member stream: the inflate leg of bench.py
  config 4  level 9 on a Silesia-like mixed corpus: ratio + MB/s next to the system zlib at level 9
Prints one JSON line per configuration.  Needs a GPU.
"""
import ctypes as C
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))

BLOCK = 131072


def tiled_dev(ctx, devmem, host, size):
    """`size` bytes of device memory holding `host` (a numpy uint8 array) tiled, + 64 readable bytes of slack: one upload, then
    device-to-device doubling (no tensor library: zlib_ng_amd.devmem over the C ABI)"""
    d = devmem.empty(ctx, size + 64)
    n = min(host.size, size)
    d[0:n] = host[:n]
    while n < size:
        k = min(n, size - n)
        d[n:n + k] = d[0:k]
        n += k
    d[size:size + 64] = 0
    return d


def deflate_dev(ctx, _lib, devmem, d_in, size, level, chained=True, steps=2):
    import numpy as np
    L, h = ctx.L, ctx.h
    nb = size // BLOCK
    blocks = (_lib.Block * nb)()
    for b in range(nb):
        blocks[b] = _lib.Block(b * BLOCK, BLOCK, 32768 if (b and chained) else 0, 0, 0)
    slots = devmem.empty(ctx, nb * _lib.SLOT_STRIDE)
    ul = devmem.empty(ctx, 4 * nb)
    uc = devmem.empty(ctx, 4 * nb)
    best = None
    for _ in range(steps + 1):
        ctx.sync()
        t0 = time.perf_counter()
        r = L.zngamd_deflate_blocks_dev(h, d_in.vp(), size, blocks, nb, level, slots.vp(), ul.vp(), uc.vp(), None)
        dt = time.perf_counter() - t0
        assert r == 0, ctx.err()
        best = dt if best is None else min(best, dt)
    lens = ul.cpu(np.int32)
    return best, int(lens.astype(np.int64).sum()), slots, lens


def main():
    from zlib_ng_amd import _lib, corpus, devmem, zlib_ng
    ctx = _lib.default_context()
    out = []

    # ---- config 0: API plumbing on incompressible data (host buffers, PCIe in the loop)
    buf = os.urandom(1 << 20)
    c = zlib_ng.compress(buf, 6)                   # first call creates the context and sizes the workspaces for 8 units
    zlib_ng.decompress(c)
    tc = td = None
    for _ in range(3):                             # steady state: best of three
        t0 = time.perf_counter(); c = zlib_ng.compress(buf, 6); t1 = time.perf_counter()
        d = zlib_ng.decompress(c); t2 = time.perf_counter()
        tc = t1 - t0 if tc is None else min(tc, t1 - t0)
        td = t2 - t1 if td is None else min(td, t2 - t1)
    assert d == buf and zlib.decompress(c) == buf
    out.append({"config": 0, "workload": "zlib_ng.compress/decompress level 6, 1 MiB os.urandom, host API, steady state",
                "compressed_bytes": len(c), "compress_ms": round(tc * 1e3, 2), "decompress_ms": round(td * 1e3, 2)})

    # ---- host-buffer API on a large input: PCIe, staging and Python buffers inside the timed region
    big = corpus.text(64 << 20, seed=1).tobytes() * 4
    zlib_ng.compress(big[:1 << 20], 6)
    first_c = first_d = best_c = best_d = None
    for it in range(3):                            # the first call of a size also sizes the workspaces and the pinned staging buffers
        c = d = None
        t0 = time.perf_counter(); c = zlib_ng.compress(big, 6); t1 = time.perf_counter()
        d = zlib_ng.decompress(c, bufsize=len(big)); t2 = time.perf_counter()
        if it == 0:
            first_c, first_d = t1 - t0, t2 - t1
        best_c = t1 - t0 if best_c is None else min(best_c, t1 - t0)
        best_d = t2 - t1 if best_d is None else min(best_d, t2 - t1)
    assert d == big
    out.append({"config": "0b", "workload": "zlib_ng.compress/decompress level 6, 256 MiB text, host API (PCIe inclusive), best of 3",
                "compress_MBps": round(len(big) / best_c / 1e6, 1), "decompress_MBps": round(len(big) / best_d / 1e6, 1),
                "first_call_compress_MBps": round(len(big) / first_c / 1e6, 1), "first_call_decompress_MBps": round(len(big) / first_d / 1e6, 1),
                "ratio": round(len(big) / len(c), 4)})
    del big, c, d

    # ---- config 1: level 1 (and 6 for comparison), 1 GiB text
    size = 1 << 30
    host = corpus.text(64 << 20, seed=1)
    d_in = tiled_dev(ctx, devmem, host, size)
    for level in (1, 6):
        dt, comp, slots, ul = deflate_dev(ctx, _lib, devmem, d_in, size, level)
        parts = []
        for b in range(64):                        # first 64 blocks through the system zlib
            n = int(ul[b])
            parts.append(slots[b * _lib.SLOT_STRIDE:b * _lib.SLOT_STRIDE + n].cpu().tobytes())
        assert zlib.decompressobj(-15).decompress(b"".join(parts)) == host[:64 * BLOCK].tobytes()
        out.append({"config": 1 if level == 1 else "1b",
                    "workload": f"deflate level {level}, 1 GiB Zipf text, 128 KiB dict-chained blocks, device resident",
                    "MBps": round(size / dt / 1e6, 1), "ratio": round(size / comp, 4),
                    "zlib_ratio_same_level": round((4 << 20) / len(zlib.compress(host[:4 << 20].tobytes(), level)), 4)})
        del slots, ul
    del d_in

    # ---- config 4: level 9 on the mixed corpus
    mixed = corpus.mixed(200 << 20, seed=5)
    size = (mixed.size // BLOCK) * BLOCK
    d_in = tiled_dev(ctx, devmem, mixed[:size], size)
    dt, comp, slots, ul = deflate_dev(ctx, _lib, devmem, d_in, size, 9, steps=1)
    import numpy as np
    # 16 x 1 MiB taken evenly across the corpus (every data class is represented)
    sample = np.concatenate([mixed[o:o + (1 << 20)] for o in range(0, size - (1 << 20), size // 16)][:16]).tobytes()
    t0 = time.perf_counter(); zc = zlib.compress(sample, 9); tz = time.perf_counter() - t0
    out.append({"config": 4, "workload": f"deflate level 9, {size >> 20} MiB Silesia-like mix, 128 KiB dict-chained blocks, device resident",
                "MBps": round(size / dt / 1e6, 1), "ratio": round(size / comp, 4),
                "zlib9_ratio_16x1MiB_strided_sample": round(len(sample) / len(zc), 4), "zlib9_MBps_1core": round(len(sample) / tz / 1e6, 1)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
