"""Device buffers over the C ABI (zngamd_dmalloc / _h2d / _d2h / _d2d / _dmemset / _compare_dev): what a harness or a
device-resident caller needs to hold inputs and outputs in HBM without a tensor library.  No torch, no numpy arithmetic on
payload bytes: copies, fills and the engine's own device compare.  All operations run on the context's stream, in order with
the engine's kernels."""
import ctypes as C

import numpy as np

from . import _lib


class DeviceBuffer:
    """`nbytes` of device memory of a context, or a view into one (`buf[a:b]`).  Indices are BYTE offsets."""

    def __init__(self, ctx, nbytes=0, _ptr=None, _base=None):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self._base = _base
        if _ptr is None:
            p = C.c_void_p()
            ctx._chk(ctx.L.zngamd_dmalloc(ctx.h, max(1, self.nbytes), C.byref(p)))
            self.ptr = p.value
            self._own = True
        else:
            self.ptr = int(_ptr)
            self._own = False

    # ---- shape
    def numel(self):
        return self.nbytes

    def data_ptr(self):
        return self.ptr

    def vp(self, off=0):
        return C.c_void_p(self.ptr + int(off))

    def _span(self, key):
        if isinstance(key, slice):
            a, b, st = key.indices(self.nbytes)
            if st != 1:
                raise ValueError("device buffers are sliced contiguously")
            return a, max(a, b)
        k = int(key)
        if k < 0:
            k += self.nbytes
        return k, k + 1

    def __getitem__(self, key):
        a, b = self._span(key)
        return DeviceBuffer(self.ctx, b - a, _ptr=self.ptr + a, _base=self._base or self)

    def __setitem__(self, key, value):
        a, b = self._span(key)
        n = b - a
        L, h = self.ctx.L, self.ctx.h
        if isinstance(value, DeviceBuffer):
            if value.nbytes != n:
                raise ValueError(f"size mismatch: {value.nbytes} into {n}")
            self.ctx._chk(L.zngamd_d2d(h, self.vp(a), value.vp(), n))
        elif isinstance(value, int):
            self.ctx._chk(L.zngamd_dmemset(h, self.vp(a), value & 0xFF, n))
        else:
            arr = np.ascontiguousarray(np.frombuffer(value, dtype=np.uint8) if not isinstance(value, np.ndarray) else value.view(np.uint8).reshape(-1))
            if arr.size != n:
                raise ValueError(f"size mismatch: {arr.size} into {n}")
            if n:
                self.ctx._chk(L.zngamd_h2d(h, self.vp(a), arr.ctypes.data_as(C.c_void_p), n))

    def zero_(self):
        self[:] = 0
        return self

    # ---- to the host
    def cpu(self, dtype=np.uint8):
        out = np.empty(self.nbytes, dtype=np.uint8)
        if self.nbytes:
            self.ctx._chk(self.ctx.L.zngamd_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.vp(), self.nbytes))
        return out.view(dtype)

    def equal(self, other):
        """byte-for-byte equal to `other` (same size), compared on the device"""
        if other.nbytes != self.nbytes:
            return False
        mm = C.c_uint64(0)
        self.ctx._chk(self.ctx.L.zngamd_compare_dev(self.ctx.h, self.vp(), other.vp(), self.nbytes, C.byref(mm)))
        return mm.value == 0

    def free(self):
        if self._own and self.ptr:
            self.ctx.L.zngamd_dfree(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr, self._own = 0, False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def empty(ctx, nbytes):
    return DeviceBuffer(ctx, nbytes)


def from_host(ctx, data):
    arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
    buf = DeviceBuffer(ctx, arr.size)
    buf[:] = arr
    return buf


def mem_info(ctx):
    f, t = C.c_uint64(0), C.c_uint64(0)
    ctx._chk(ctx.L.zngamd_mem_info(ctx.h, C.byref(f), C.byref(t)))
    return f.value, t.value
