"""Kernel times of the deflate pipeline per CLASS of the Silesia-like mix (corpus.mixed), level LEVEL (default 9), 28 MiB each."""
import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import _lib, corpus, devmem
ctx = _lib.default_context(); L, h = ctx.L, ctx.h
B = 131072
level = int(os.environ.get("LEVEL", "9"))
a = corpus.mixed(7 * (28 << 20), seed=5)
part = a.size // 7
for ci, name in enumerate(["text", "xml", "fastq", "walk", "sparse", "soup", "random"]):
    x = a[ci * part:(ci + 1) * part]
    size = (x.size // B) * B; nb = size // B
    d = devmem.empty(ctx, size + 64); d[:size] = x[:size].copy(); d[size:] = 0
    blocks = (_lib.Block * nb)()
    for b in range(nb): blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
    out = devmem.empty(ctx, size + nb * 64); ul = devmem.empty(ctx, 4 * nb); uc = devmem.empty(ctx, 4 * nb)
    tot = C.c_uint64(0)
    for it in range(2):
        ctx.profiling(True); ctx.kernel_times(True)
        r = L.zngamd_deflate_blocks_packed_dev(h, d.vp(), size, blocks, nb, level, out.vp(), out.nbytes, ul.vp(), uc.vp(), None, C.byref(tot))
        kt = ctx.kernel_times(True)
    assert r == 0, ctx.err()
    print(f"{name:7s} level {level} ratio {size / tot.value:7.3f}  " + "  ".join(f"{k} {v[0]:.2f}" for k, v in kt.items() if v[1]))
