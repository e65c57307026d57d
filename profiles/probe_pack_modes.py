"""Probe: does the time of za_k_pack depend on where the output slots lie?  One process, the same 1 GiB deflate with the slot
array placed at different offsets inside one large allocation."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import torch
from zlib_ng_amd import _lib, corpus
ctx = _lib.Context(0); L, h = ctx.L, ctx.h
n = 1 << 30; B = 131072
host = corpus.text(32 << 20)
d = torch.from_numpy(host).cuda().repeat(n // (32 << 20)); d = torch.cat([d, torch.zeros(64, dtype=torch.uint8, device="cuda")])
nb = n // B
blocks = (_lib.Block * nb)()
for b in range(nb): blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
big = torch.empty(nb * _lib.SLOT_STRIDE + (64 << 20), dtype=torch.uint8, device="cuda")
ul = torch.empty(nb, dtype=torch.int32, device="cuda"); uc = torch.empty(nb, dtype=torch.int32, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
for off in (0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096, 3 << 20, 17 << 20, 0):
    slots = big[off:off + nb * _lib.SLOT_STRIDE]
    res = []
    for it in range(3):
        ctx.profiling(True); ctx.kernel_times(True)
        r = L.zngamd_deflate_blocks_dev(h, p(d), n, blocks, nb, 6, p(slots), p(ul), p(uc), None)
        kt = ctx.kernel_times(True)
        res.append(round(kt["pack"][0], 3))
    print("offset %9d  ptr %% 2MiB = %8d  pack ms %s  parse %.3f" % (off, slots.data_ptr() % (2 << 20), res, kt["parse"][0]))
