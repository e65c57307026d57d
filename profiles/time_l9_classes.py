"""Level-9 (and 6) deflate of every data class of the Silesia-like mix on its own: which class costs what (kernel times per GiB)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd")); sys.path.insert(0, ROOT)
import torch
from zlib_ng_amd import _lib, corpus
ctx = _lib.default_context(); L, h = ctx.L, ctx.h
B = 131072
mixed = corpus.mixed(200 << 20, seed=5)
part = (200 << 20) // 7
names = ["text", "xml", "fastq", "walk", "sparse", "soup", "random"]
p = lambda t: C.c_void_p(t.data_ptr())
for level in (9, 6):
    for i, nm in enumerate(names + ["all"]):
        a = mixed[i * part:(i + 1) * part] if nm != "all" else mixed
        size = (a.size // B) * B
        nb = size // B
        d = torch.cat([torch.from_numpy(a[:size].copy()).cuda(), torch.zeros(64, dtype=torch.uint8, device="cuda")])
        blocks = (_lib.Block * nb)()
        for b in range(nb): blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
        slots = torch.empty(nb * _lib.SLOT_STRIDE, dtype=torch.uint8, device="cuda"); ul = torch.empty(nb, dtype=torch.int32, device="cuda"); uc = torch.empty(nb, dtype=torch.int32, device="cuda")
        for it in range(2):
            torch.cuda.synchronize(); ctx.profiling(True); ctx.kernel_times(True)
            t = time.perf_counter()
            r = L.zngamd_deflate_blocks_dev(h, p(d), size, blocks, nb, level, p(slots), p(ul), p(uc), None)
            dt = time.perf_counter() - t
            kt = ctx.kernel_times(True); ctx.profiling(False)
        comp = int(ul.to(torch.int64).sum().item())
        print(f"L{level} {nm:7s} {size >> 20:4d} MiB ratio {size / comp:7.3f}  {size / dt / 1e9:6.1f} GB/s  ms per GiB:", {k: round(v[0] * (1 << 30) / size, 2) for k, v in kt.items() if v[1]})
        del d, slots, ul, uc
