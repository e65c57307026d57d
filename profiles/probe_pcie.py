"""Host <-> device copy rates on this box (pageable / pinned / registered, single and chunked), and the host's own memcpy rate:
what the host-buffer entry points can reach (profiles/r03_pcie.txt)."""
import ctypes as C, time, numpy as np, threading
hip = C.CDLL("libamdhip64.so")
def chk(r):
    assert r == 0, r
N = 256 << 20
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), N))
pin = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(pin), N, 0))
page = np.ones(N, dtype=np.uint8)
pinarr = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), (N,))
pinarr[:] = 1
def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); hip.hipDeviceSynchronize(); best = min(best, time.perf_counter() - t0)
    return N / best / 1e9
print("H2D pageable hipMemcpy      %.1f GB/s" % t(lambda: chk(hip.hipMemcpy(d, page.ctypes.data_as(C.c_void_p), N, 1))))
print("H2D pinned   hipMemcpy      %.1f GB/s" % t(lambda: chk(hip.hipMemcpy(d, pin, N, 1))))
print("D2H pageable hipMemcpy      %.1f GB/s" % t(lambda: chk(hip.hipMemcpy(page.ctypes.data_as(C.c_void_p), d, N, 2))))
print("D2H pinned   hipMemcpy      %.1f GB/s" % t(lambda: chk(hip.hipMemcpy(pin, d, N, 2))))
fresh = lambda: np.empty(N, dtype=np.uint8)
def d2h_fresh():
    a = fresh(); chk(hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), d, N, 2))
print("D2H fresh pageable (first touch) %.1f GB/s" % t(d2h_fresh))
def reg():
    chk(hip.hipHostRegister(page.ctypes.data_as(C.c_void_p), N, 0)); chk(hip.hipMemcpy(d, page.ctypes.data_as(C.c_void_p), N, 1)); chk(hip.hipHostUnregister(page.ctypes.data_as(C.c_void_p)))
print("H2D register+copy+unregister %.1f GB/s" % t(reg))
libc = C.CDLL("libc.so.6")
libc.memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
print("memcpy pageable->pinned 1 thread %.1f GB/s" % t(lambda: libc.memcpy(pin, page.ctypes.data_as(C.c_void_p), N)))
def mt(k):
    def run():
        th = []
        step = N // k
        for i in range(k):
            th.append(threading.Thread(target=libc.memcpy, args=(pin.value + i * step, page.ctypes.data + i * step, step)))
        for x in th: x.start()
        for x in th: x.join()
    return run
for k in (2, 4, 8):
    print("memcpy pageable->pinned %d threads %.1f GB/s" % (k, t(mt(k))))
# chunked overlap: memcpy chunk k+1 to pinned while chunk k travels
s0 = C.c_void_p(); chk(hip.hipStreamCreate(C.byref(s0)))
def piped(chunk, threads):
    def run():
        nchunk = N // chunk
        pend = None
        for c in range(nchunk):
            off = c * chunk
            step = chunk // threads
            th = [threading.Thread(target=libc.memcpy, args=(pin.value + off + i * step, page.ctypes.data + off + i * step, step)) for i in range(threads)]
            for x in th: x.start()
            for x in th: x.join()
            chk(hip.hipMemcpyAsync(C.c_void_p(d.value + off), C.c_void_p(pin.value + off), chunk, 1, s0))
        chk(hip.hipStreamSynchronize(s0))
    return run
for chunk, th in ((16 << 20, 1), (16 << 20, 4), (32 << 20, 4), (8 << 20, 4)):
    print("H2D staged through pinned, %d MiB chunks, %d copy threads: %.1f GB/s" % (chunk >> 20, th, t(piped(chunk, th))))
