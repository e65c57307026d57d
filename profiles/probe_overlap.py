"""Does the deflate pipeline gain from running two batches side by side (kernels of different stages sharing the CUs)?
Two contexts (own streams and workspaces) on one GPU, each compressing its half of the input from its own thread, against one
context compressing everything; STAGGER: the second thread starts that many ms later."""
import ctypes as C, os, sys, threading, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, devmem
B = 131072
n = int(os.environ.get("MIB", "2048")) << 20
host = corpus.text(64 << 20, seed=1)
ctxs = [_lib.Context(0), _lib.Context(0)]
def setup(ctx, size):
    d = devmem.empty(ctx, size + 64)
    for o in range(0, size, host.size): d[o:o + min(host.size, size - o)] = host[:min(host.size, size - o)]
    d[size:] = 0
    nb = size // B
    blocks = (_lib.Block * nb)()
    for b in range(nb): blocks[b] = _lib.Block(b * B, B, 32768 if b else 0, 0, 0)
    return dict(ctx=ctx, d=d, nb=nb, size=size, blocks=blocks, out=devmem.empty(ctx, size // 2 + (64 << 20)), ul=devmem.empty(ctx, 4 * nb), uc=devmem.empty(ctx, 4 * nb))
def run(j):
    tot = C.c_uint64(0)
    r = j["ctx"].L.zngamd_deflate_blocks_packed_dev(j["ctx"].h, j["d"].vp(), j["size"], j["blocks"], j["nb"], 6, j["out"].vp(), j["out"].nbytes - 64, j["ul"].vp(), j["uc"].vp(), None, C.byref(tot))
    assert r == 0, j["ctx"].err()
full = setup(ctxs[0], n)
for _ in range(2): run(full)
t = time.perf_counter(); run(full); t_full = time.perf_counter() - t
del full
halves = [setup(ctxs[0], n // 2), setup(ctxs[1], n // 2)]
for h in halves: run(h); run(h)
t = time.perf_counter(); run(halves[0]); t_half = time.perf_counter() - t
for stagger in (0.0, 0.004, 0.008):
    def second():
        time.sleep(stagger); run(halves[1])
    best = 1e9
    for _ in range(3):
        th = threading.Thread(target=second)
        t = time.perf_counter(); th.start(); run(halves[0]); th.join(); best = min(best, time.perf_counter() - t)
    print(f"{n >> 20} MiB: one context {t_full * 1e3:.1f} ms; one half alone {t_half * 1e3:.1f} ms; two halves side by side (second {stagger * 1e3:.0f} ms later) {best * 1e3:.1f} ms")
