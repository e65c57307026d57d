"""Level 9 on the 199 MiB Silesia-like mix, three passes, for `rocprofv3 --kernel-trace --stats` (BASELINE config 5):
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 profiles/prof_level9.py"""
import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd")); sys.path.insert(0, ROOT)
import torch
from zlib_ng_amd import _lib, corpus
ctx = _lib.default_context(); L, h = ctx.L, ctx.h
B = 131072
a = corpus.mixed(200 << 20, seed=5)
size = (a.size // B) * B
nb = size // B
d = torch.cat([torch.from_numpy(a[:size].copy()).cuda(), torch.zeros(64, dtype=torch.uint8, device="cuda")])
blocks = (_lib.Block * nb)()
for b in range(nb): blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
out = torch.empty(size + nb * 64, dtype=torch.uint8, device="cuda"); ul = torch.empty(nb, dtype=torch.int32, device="cuda"); uc = torch.empty(nb, dtype=torch.int32, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
tot = C.c_uint64(0)
for it in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = L.zngamd_deflate_blocks_packed_dev(h, p(d), size, blocks, nb, 9, p(out), out.numel(), p(ul), p(uc), None, C.byref(tot))
    dt = time.perf_counter() - t
    assert r == 0, ctx.err()
    print(f"pass {it}: {size >> 20} MiB level 9 ratio {size / tot.value:.4f} {size / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms)")
