"""Kernel time of the two-pass member inflate on MIB (default 1024) MiB of level-6 members; with ZNGAMD_LIB pointing at an
ablation build (profiles/abl_inflate.sh) the output check is only reported, not required."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import torch
from zlib_ng_amd import _lib, corpus
ctx = _lib.Context(0); L, h = ctx.L, ctx.h
n = int(os.environ.get("MIB", "1024")) << 20; B = 131072; nb = n // B
host = corpus.text(64 << 20, seed=1)
d = torch.cat([torch.from_numpy(host).cuda().repeat(n // host.size), torch.zeros(64, dtype=torch.uint8, device="cuda")])
p = lambda t: C.c_void_p(t.data_ptr())
ms = torch.empty(n // 2 + nb * 2800 + (64 << 20), dtype=torch.uint8, device="cuda")
ml, mn = C.c_uint64(0), C.c_uint32(0)
assert L.zngamd_gzip_members_dev(h, p(d), n, B, 6, p(ms), ms.numel() - 64, C.byref(ml), C.byref(mn)) == 0
mtab = torch.empty(nb * C.sizeof(_lib.Member), dtype=torch.uint8, device="cuda")
mstat = torch.empty(nb, dtype=torch.int32, device="cuda")
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
nm, tot = C.c_uint32(0), C.c_uint64(0)
best = None
for it in range(4):
    ctx.profiling(True); ctx.kernel_times(True)
    assert L.zngamd_gzip_scan_dev(h, p(ms), ml.value, p(mtab), nb, C.byref(nm), C.byref(tot)) == 0
    r = L.zngamd_gzip_inflate_members_dev(h, p(ms), ml.value, p(mtab), nm.value, p(out), n, p(mstat))
    kt = ctx.kernel_times(True)
    best = kt["inflate"][0] if best is None else min(best, kt["inflate"][0])
ok = bool((out[:n] == d[:n]).all().item()) and int((mstat != 0).sum().item()) == 0
print("%-40s rc %d output ok %-5s inflate %.3f ms scan %.3f ms (member stream %d bytes)" %
      (os.environ.get("ABL", "product build"), r, ok, best, kt["scan"][0], ml.value))
