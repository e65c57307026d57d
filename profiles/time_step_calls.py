import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo/python-zlib-ng_amd")
import torch
from zlib_ng_amd import _lib, corpus
ctx = _lib.Context(0); L, h = ctx.L, ctx.h
n = 4096 << 20; B = 131072; nb = n // B
host = corpus.text(64 << 20, seed=1)
d = torch.cat([torch.from_numpy(host).cuda().repeat(n // host.size), torch.zeros(64, dtype=torch.uint8, device="cuda")])
p = lambda t: C.c_void_p(t.data_ptr())
ms = torch.empty(n // 2 + nb * 2800 + (64 << 20), dtype=torch.uint8, device="cuda")
ml, mn = C.c_uint64(0), C.c_uint32(0)
assert L.zngamd_gzip_members_dev(h, p(d), n, B, 6, p(ms), ms.numel() - 64, C.byref(ml), C.byref(mn)) == 0
mtab = torch.empty(nb * C.sizeof(_lib.Member), dtype=torch.uint8, device="cuda")
nm, tot = C.c_uint32(0), C.c_uint64(0)
torch.cuda.synchronize()
for it in range(4):
    ctx.profiling(True); ctx.kernel_times(True)
    t0 = time.perf_counter()
    assert L.zngamd_gzip_scan_dev(h, p(ms), ml.value, p(mtab), nb, C.byref(nm), C.byref(tot)) == 0
    dt = time.perf_counter() - t0
    kt = ctx.kernel_times(True)
    print("scan call %.3f ms wall, kernel %.3f ms" % (dt * 1e3, kt["scan"][0]))
blocks = (_lib.Block * nb)()
for b in range(nb): blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
slots = torch.empty(nb * _lib.SLOT_STRIDE, dtype=torch.uint8, device="cuda"); ul = torch.empty(nb, dtype=torch.int32, device="cuda"); uc = torch.empty(nb, dtype=torch.int32, device="cuda")
comp = torch.empty(n // 2 + (64 << 20), dtype=torch.uint8, device="cuda"); ct = C.c_uint64(0)
for it in range(3):
    ctx.kernel_times(True)
    t0 = time.perf_counter(); r = L.zngamd_deflate_blocks_dev(h, p(d), n, blocks, nb, 6, p(slots), p(ul), p(uc), None); t1 = time.perf_counter()
    r2 = L.zngamd_gather_dev(h, p(slots), p(ul), nb, p(comp), 0, comp.numel(), None, C.byref(ct)); t2 = time.perf_counter()
    kt = ctx.kernel_times(True)
    print("deflate call %.3f ms wall, kernels %.3f; gather call %.3f wall, kernels %.3f" % ((t1 - t0) * 1e3, sum(kt[k][0] for k in ("chains", "search", "parse", "plan", "pack")), (t2 - t1) * 1e3, kt["gather"][0]))
