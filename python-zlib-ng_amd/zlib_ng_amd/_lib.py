"""ctypes binding of libzng_amd.so (C ABI declared in include/zng_amd.h).

There is no fallback: if the library is missing, or no MI355X-class GPU is usable, importing
works but the first call that needs the engine raises ``RuntimeError``.
"""
import ctypes as C

import numpy as np
import mmap as _mmap
import os
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZNGAMD_LIB: measurement builds of the same library kept elsewhere (profiles/*.sh compile their variants to a scratch path and
# point this at them -- the product library in the tree is never overwritten)
LIB_PATH = os.environ.get("ZNGAMD_LIB") or os.path.join(_HERE, "libzng_amd.so")

OK, STREAM_END, NEED_DICT = 0, 1, 2
STREAM_ERROR, DATA_ERROR, MEM_ERROR, BUF_ERROR = -2, -3, -4, -5
E_GZ_MAGIC, E_GZ_METHOD, E_GZ_HCRC, E_GZ_CRC, E_GZ_LENGTH, E_GZ_TRUNC = -101, -102, -103, -104, -105, -106
E_HIP, E_ARG, E_OVERFLOW = -201, -202, -203

FLAG_FINAL = 1
FLAG_FLATHDR = 2
CHUNK_SHIFT = 11
UNIT_MAX = 131072
SLOT_STRIDE = 131136
SEG = 2048
INDEX_STRIDE = 68        # ZNGAMD_INDEX_STRIDE
FLAG_FINAL, FLAG_FLATHDR, FLAG_SEG2K, FLAG_UNITS16K = 1, 2, 16, 32
# The writer's segment index in a FILE (r06): behind the data member, EMPTY gzip members (header with FEXTRA, `03 00`, zero CRC and
# ISIZE) whose 'Z','A' subfield holds: version 3, kind 1, the number of records (u16), the first record's unit number (u32), then
# records of 138 bytes -- a unit's compressed bytes (u32, sync marker included), its output bytes (u32), 65 x u16: the bit offset of
# its first segment's first token, then each segment's length in bits (the last one ends at the end-of-block code; zeros behind
# it; all zeros = stored blocks).  The LAST of these members is a locator of 54 bytes (kind 2): members, units, bytes of the index
# members, bytes of the data member in front of them -- a reader that can seek finds the index from the file's end.
INDEX_REC = None
INDEX_PER_MEMBER = 470
INDEX_LOCATOR_BYTES = 54
E_INDEX = -204
K_NAMES = ["chains", "search", "parse", "plan", "pack", "gather", "scan", "inflate", "other", "optparse"]

# every symbol include/zng_amd.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "zngamd_device_count", "zngamd_ctx_create", "zngamd_ctx_destroy", "zngamd_last_error", "zngamd_version",
    "zngamd_set_stream", "zngamd_sync", "zngamd_dmalloc", "zngamd_dfree", "zngamd_h2d", "zngamd_d2h",
    "zngamd_crc32", "zngamd_adler32", "zngamd_crc32_dev", "zngamd_crc32_combine", "zngamd_crc32_combine_many", "zngamd_level_ok",
    "zngamd_deflate_blocks", "zngamd_deflate_blocks_packed", "zngamd_count_units", "zngamd_deflate_blocks_dev", "zngamd_deflate_blocks_packed_dev", "zngamd_gather_dev",
    "zngamd_deflate_stream", "zngamd_inflate_raw", "zngamd_inflate_resume", "zngamd_gzip_scan_dev", "zngamd_gzip_inflate_members_dev",
    "zngamd_gzip_inflate_plain_members_dev", "zngamd_inflate_raw_dev", "zngamd_compare_dev", "zngamd_crc32_fold_dev",
    "zngamd_stream_deflate_init", "zngamd_stream_deflate", "zngamd_stream_deflate_set_dictionary", "zngamd_stream_deflate_copy", "zngamd_stream_pending", "zngamd_stream_inflate_ahead",
    "zngamd_stream_deflate_end", "zngamd_stream_deflate_reset", "zngamd_stream_inflate_reset", "zngamd_stream_inflate_init", "zngamd_stream_inflate", "zngamd_stream_inflate_set_dictionary",
    "zngamd_stream_inflate_copy", "zngamd_stream_inflate_end",
    "zngamd_comm_unique_id", "zngamd_comm_create", "zngamd_comm_destroy", "zngamd_comm_last_error", "zngamd_comm_count", "zngamd_comm_layout",
    "zngamd_comm_allgather_stream", "zngamd_comm_offsets", "zngamd_comm_wait", "zngamd_comm_barrier", "zngamd_comm_max_f64",
    "zngamd_gunzip", "zngamd_gunzip_partial", "zngamd_gunzip_stream", "zngamd_gzip_members", "zngamd_gzip_members_dev", "zngamd_profiling",
    "zngamd_kernel_times", "zngamd_kernel_class_count", "zngamd_abi", "zngamd_decode_paths", "zngamd_deflate_index_dev", "zngamd_inflate_units_indexed_dev", "zngamd_index_create", "zngamd_index_destroy", "zngamd_deflate_index", "zngamd_deflate_blocks_packed_indexed", "zngamd_indexed_units", "zngamd_debug_fetch", "zngamd_debug_keep", "zngamd_d2d", "zngamd_dmemset", "zngamd_mem_info",
]


class GzState(C.Structure):                    # zngamd_gz_state
    _fields_ = [("in_member", C.c_uint32), ("start_bit", C.c_uint32), ("crc", C.c_uint32), ("window_len", C.c_uint32),
                ("out_total", C.c_uint64), ("window", C.c_uint8 * 32768), ("index", C.c_void_p)]


class Block(C.Structure):
    _fields_ = [("off", C.c_uint64), ("len", C.c_uint32), ("dict_len", C.c_uint32),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class Member(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint64), ("out_off", C.c_uint64),
                ("out_len", C.c_uint32), ("crc", C.c_uint32), ("index_off", C.c_uint32), ("nseg", C.c_uint32)]


_lib = None
_lib_lock = threading.Lock()


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python python-zlib-ng_amd/build.py` "
                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        vp, u8p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        L.zngamd_device_count.restype = C.c_int
        L.zngamd_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.zngamd_ctx_destroy.argtypes = [vp]
        L.zngamd_ctx_destroy.restype = None
        L.zngamd_last_error.argtypes = [vp]
        L.zngamd_last_error.restype = C.c_char_p
        L.zngamd_version.restype = C.c_char_p
        L.zngamd_set_stream.argtypes = [vp, vp]
        L.zngamd_sync.argtypes = [vp]
        L.zngamd_dmalloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.zngamd_dfree.argtypes = [vp, vp]
        L.zngamd_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.zngamd_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.zngamd_crc32.argtypes = [vp, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.zngamd_adler32.argtypes = [vp, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.zngamd_crc32_dev.argtypes = [vp, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_uint32)]
        L.zngamd_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.zngamd_crc32_combine.restype = C.c_uint32
        L.zngamd_crc32_combine_many.argtypes = [C.c_uint32, u32p, C.POINTER(C.c_uint64), C.c_uint32]
        L.zngamd_crc32_combine_many.restype = C.c_uint32
        L.zngamd_level_ok.argtypes = [C.c_int]
        L.zngamd_deflate_blocks.argtypes = [vp, u8p, C.c_uint64, C.POINTER(Block), C.c_uint32, C.c_int,
                                            u8p, C.c_uint64, u32p, u32p]
        L.zngamd_deflate_blocks_packed.argtypes = [vp, u8p, C.c_uint64, C.POINTER(Block), C.c_uint32, C.c_int,
                                                   u8p, C.c_uint64, C.c_uint64, u32p, u32p, C.POINTER(C.c_uint64)]
        L.zngamd_deflate_blocks_packed_indexed.argtypes = [vp, u8p, C.c_uint64, C.POINTER(Block), C.c_uint32, C.c_int,
                                                           u8p, C.c_uint64, C.c_uint64, u32p, u32p, C.POINTER(C.c_uint64),
                                                           C.c_uint32, vp, vp, vp]
        L.zngamd_count_units.argtypes = [C.POINTER(Block), C.c_uint32]
        L.zngamd_count_units.restype = C.c_uint32
        L.zngamd_deflate_blocks_dev.argtypes = [vp, vp, C.c_uint64, C.POINTER(Block), C.c_uint32, C.c_int,
                                                vp, vp, vp, u32p]
        L.zngamd_deflate_blocks_packed_dev.argtypes = [vp, vp, C.c_uint64, C.POINTER(Block), C.c_uint32, C.c_int, vp, C.c_uint64, vp, vp, vp,
                                                       C.POINTER(C.c_uint64)]
        L.zngamd_gather_dev.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint64, C.c_uint64, vp,
                                        C.POINTER(C.c_uint64)]
        L.zngamd_deflate_stream.argtypes = [vp, u8p, C.c_uint64, C.c_int, C.c_int, u8p, C.c_uint64,
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.zngamd_inflate_raw.argtypes = [vp, u8p, C.c_uint64, u8p, C.c_uint32, u8p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.zngamd_inflate_resume.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint32, u8p, C.c_uint64,
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]
        L.zngamd_gzip_scan_dev.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint64)]
        L.zngamd_gzip_inflate_members_dev.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint64, vp]
        L.zngamd_gzip_inflate_plain_members_dev.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint64, vp]
        L.zngamd_inflate_raw_dev.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.zngamd_compare_dev.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.zngamd_crc32_fold_dev.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        L.zngamd_gunzip.argtypes = [vp, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint32)]
        L.zngamd_gunzip_partial.argtypes = [vp, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.zngamd_gunzip_stream.argtypes = [vp, C.POINTER(GzState), u8p, C.c_uint64, C.c_int, u8p, C.c_uint64,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.zngamd_gzip_members.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, C.c_int, u8p, C.c_uint64,
                                          C.POINTER(C.c_uint64)]
        L.zngamd_gzip_members_dev.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp, C.c_uint64,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.zngamd_profiling.argtypes = [vp, C.c_int]
        L.zngamd_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.zngamd_decode_paths.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
        L.zngamd_debug_fetch.argtypes = [vp, C.c_int, C.c_uint32, vp, C.c_size_t]
        L.zngamd_deflate_index_dev.argtypes = [vp, vp, C.c_uint32]
        L.zngamd_index_create.argtypes = [vp, C.c_uint32, vp, vp, vp, C.POINTER(vp)]
        L.zngamd_index_destroy.argtypes = [vp]
        L.zngamd_index_destroy.restype = None
        L.zngamd_deflate_index.argtypes = [vp, C.c_uint32, vp, vp, vp]
        L.zngamd_indexed_units.argtypes = [vp, C.c_int]
        L.zngamd_indexed_units.restype = C.c_uint64
        L.zngamd_inflate_units_indexed_dev.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, vp, vp, C.c_uint32,
                                                       vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.zngamd_debug_keep.argtypes = [vp, C.c_int]
        L.zngamd_d2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.zngamd_dmemset.argtypes = [vp, vp, C.c_int, C.c_size_t]
        L.zngamd_mem_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.zngamd_comm_offsets.restype = C.c_uint64
        L.zngamd_comm_offsets.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
        _lib = L
        return L


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zng_amd error {code}: {msg}")
        self.code = code
        self.msg = msg


class _PyBuffer(C.Structure):                  # Py_buffer (CPython's buffer protocol view)
    _fields_ = [("buf", C.c_void_p), ("obj", C.c_void_p), ("len", C.c_ssize_t), ("itemsize", C.c_ssize_t),
                ("readonly", C.c_int), ("ndim", C.c_int), ("format", C.c_char_p), ("shape", C.c_void_p),
                ("strides", C.c_void_p), ("suboffsets", C.c_void_p), ("internal", C.c_void_p)]


C.pythonapi.PyObject_GetBuffer.argtypes = [C.py_object, C.POINTER(_PyBuffer), C.c_int]
C.pythonapi.PyObject_GetBuffer.restype = C.c_int
C.pythonapi.PyBuffer_Release.argtypes = [C.POINTER(_PyBuffer)]
C.pythonapi.PyBuffer_Release.restype = None


class _Pinned:
    """Holds a buffer-protocol view of an object (read-only ones included) for the duration of an engine call."""

    def __init__(self, obj):
        self.view = _PyBuffer()
        self.ok = C.pythonapi.PyObject_GetBuffer(obj, C.byref(self.view), 0) == 0      # PyBUF_SIMPLE

    def __del__(self):
        if getattr(self, "ok", False):
            C.pythonapi.PyBuffer_Release(C.byref(self.view))
            self.ok = False


def _addr(buf):
    """Address + keep-alive object of a bytes-like, without copying (contiguous buffers of any kind)."""
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p), buf
    try:
        pin = _Pinned(buf)                       # raises BufferError for non-contiguous exporters
    except Exception:
        pin = None
    if pin is not None and pin.ok:
        return C.c_void_p(pin.view.buf), (pin, buf)
    b = memoryview(buf).tobytes()
    return C.cast(C.c_char_p(b), C.c_void_p), b


_py = C.pythonapi
_py.PyBytes_FromStringAndSize.restype = C.py_object
_py.PyBytes_FromStringAndSize.argtypes = [C.c_void_p, C.c_ssize_t]
_py.PyBytes_AsString.restype = C.c_void_p
_py.PyBytes_AsString.argtypes = [C.py_object]


_HUGE_MIN = 8 << 20
try:
    _madvise = C.CDLL(None, use_errno=True).madvise
    _madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    _madvise.restype = C.c_int
except Exception:                                    # pragma: no cover
    _madvise = None


def _advise_huge(addr, n):
    """Fresh memory of a large buffer is first touched by the engine's copy threads or a file read, one page fault per 4 KiB
    (21 000 for an 88 MB result, more time than the copy itself) -- with transparent huge pages one per 2 MiB, where the
    system allows them on request.  A refusal changes nothing."""
    if n >= _HUGE_MIN and _madvise is not None:
        lo = (addr + 0x1FFFFF) & ~0x1FFFFF
        hi = (addr + n) & ~0x1FFFFF
        if hi > lo:
            _madvise(C.c_void_p(lo), C.c_size_t(hi - lo), 14)       # MADV_HUGEPAGE


def _new_bytes(n):
    """A fresh, uninitialised bytes object of n >= 1 bytes for the engine to fill, and its address: no zero fill,
    and no copy when the engine fills it completely."""
    obj = _py.PyBytes_FromStringAndSize(None, max(int(n), 1))
    addr = _py.PyBytes_AsString(obj)
    _advise_huge(addr, int(n))
    return obj, C.c_void_p(addr)


new_buffer = _new_bytes          # for callers that keep an output buffer across calls (see Context.gunzip_stream)


def new_fillable(n):
    """(bytes object, writable byte view of it) for `readinto`: an input window that is not zero-filled first -- a
    bytearray(n) costs a memset of the whole window however little the file then delivers."""
    obj, addr = _new_bytes(n)
    return obj, memoryview((C.c_ubyte * max(int(n), 1)).from_address(addr.value)).cast("B")


# raw-pointer views of four C-API functions: the object below is owned through a bare pointer until it is handed over
_raw_new = C.PYFUNCTYPE(C.c_void_p, C.c_void_p, C.c_ssize_t)(("PyBytes_FromStringAndSize", C.pythonapi))
_raw_buf = C.PYFUNCTYPE(C.c_void_p, C.c_void_p)(("PyBytes_AsString", C.pythonapi))
_raw_resize = C.PYFUNCTYPE(C.c_int, C.POINTER(C.c_void_p), C.c_ssize_t)(("_PyBytes_Resize", C.pythonapi))
_raw_decref = C.PYFUNCTYPE(None, C.c_void_p)(("Py_DecRef", C.pythonapi))


class _Out:
    """An engine output buffer on its way to becoming the result: a fresh bytes object that Python has not seen yet (it is
    held through a bare pointer with its single reference), so it can be cut to the produced length in place
    (_PyBytes_Resize, what CPython's own zlib module does) instead of being copied into a second object."""
    __slots__ = ("ptr", "cap")

    def __init__(self, n):
        self.cap = max(int(n), 1)
        self.ptr = C.c_void_p(_raw_new(None, self.cap))
        if not self.ptr.value:
            raise MemoryError("cannot allocate the result")
        _advise_huge(_raw_buf(self.ptr), self.cap)

    def addr(self):
        return C.c_void_p(_raw_buf(self.ptr))

    def resize(self, n):
        """Grow (or cut) the buffer in place; its address may change."""
        n = max(int(n), 1)
        if _raw_resize(C.byref(self.ptr), n) != 0:
            self.ptr = C.c_void_p(None)          # _PyBytes_Resize released the object on failure
            raise MemoryError("cannot resize the result")
        self.cap = n

    def take(self, n):
        n = min(int(n), self.cap)
        if n == 0:
            self._drop()
            return b""
        if n != self.cap and _raw_resize(C.byref(self.ptr), n) != 0:
            self.ptr = C.c_void_p(None)          # _PyBytes_Resize released the object on failure
            raise MemoryError("cannot resize the result")
        res = C.cast(self.ptr, C.py_object).value     # a new reference for Python ...
        self._drop()                                   # ... and ours is given up
        return res

    def _drop(self):
        if self.ptr is not None and self.ptr.value:
            _raw_decref(self.ptr)
        self.ptr = C.c_void_p(None)

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass


def _take(obj, n):
    return obj if n == len(obj) else obj[:n]


# Large host buffers of the streaming writers and readers (collected input, packed output, file windows).  Fresh memory costs a
# page fault per 4 KiB on first touch -- 3.3 us each on the GPU boxes, 54 ms for a 64 MiB buffer, more than compressing it -- so
# such buffers ask for transparent huge pages and go back to a small process-wide pool when their owner closes (like the
# contexts' device workspaces, which also live as long as the process): the next file opened finds them warm.
_POOL_MAX_BYTES = max(0, int(os.environ.get("ZNGAMD_HOST_POOL_MIB", "1024"))) << 20      # (0: nothing is kept)
_buffer_pool, _buffer_pool_lock = [], threading.Lock()


def take_buffer(n):
    """A writable buffer of at least n bytes (contents arbitrary), warm if the pool has one."""
    n = max(int(n), 1)
    with _buffer_pool_lock:
        best = None
        for i, b in enumerate(_buffer_pool):
            if len(b) >= n and (best is None or len(b) < len(_buffer_pool[best])):
                best = i
        if best is not None and len(_buffer_pool[best]) <= 2 * n + (1 << 20):
            return _buffer_pool.pop(best)
    # an anonymous private mapping, not a bytearray: bytearray(n) writes n zeros at once -- a page fault per 4 KiB before the request
    # for huge pages can be made, 100 ms for the 320 MiB of a reader's window -- where the mapping's pages come into being when
    # they are first written, two MiB at a time
    buf = _mmap.mmap(-1, n, flags=_mmap.MAP_PRIVATE | _mmap.MAP_ANONYMOUS)
    if n >= _HUGE_MIN:
        try:
            buf.madvise(_mmap.MADV_HUGEPAGE)
        except (AttributeError, OSError, ValueError):
            pass
    return buf


def give_buffer(buf):
    """Hand a buffer from take_buffer() back (nothing may still read or write it)."""
    if not isinstance(buf, (bytearray, _mmap.mmap)) or len(buf) < (1 << 20):
        return
    with _buffer_pool_lock:
        if sum(len(b) for b in _buffer_pool) + len(buf) <= _POOL_MAX_BYTES:
            _buffer_pool.append(buf)


def take_window(n):
    """(buffer, address) for Context.gunzip_stream(into=...): a pooled buffer the engine decodes a window into."""
    buf = take_buffer(n)
    anchor = C.c_char.from_buffer(buf)
    addr = C.c_void_p(C.addressof(anchor))
    del anchor
    return buf, addr


def crc32_combine_many(crc, crcs, lens):
    """crc32_combine over a run of pieces in one call (a writer thread that folds 512 blocks one foreign call at a time hands
    the interpreter lock back and forth 1 000 times while its caller wants it)."""
    n = len(crcs)
    a = (C.c_uint32 * max(n, 1))(*crcs)
    b = (C.c_uint64 * max(n, 1))(*lens)
    return load().zngamd_crc32_combine_many(crc & 0xFFFFFFFF, a, b, n)


def _index_rec_dtype():
    global INDEX_REC
    if INDEX_REC is None:
        INDEX_REC = np.dtype([("in_len", "<u4"), ("out_len", "<u4"), ("e", "<u2", (65,))])
        assert INDEX_REC.itemsize == 138
    return INDEX_REC


def _za_member(payload):
    """an empty gzip member whose FEXTRA field is one 'Z','A' subfield"""
    import struct
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 4 + len(payload)) + b"ZA" + struct.pack("<H", len(payload)) +
            payload + b"\x03\x00" + bytes(8))


def index_members(recs, data_member_bytes):
    """The index members + locator for the records of one data member (a list of INDEX_REC arrays in unit order)."""
    import struct
    allrec = np.concatenate(recs) if len(recs) != 1 else recs[0]
    out, nm = [], 0
    for lo in range(0, len(allrec), INDEX_PER_MEMBER):
        part = allrec[lo:lo + INDEX_PER_MEMBER]
        out.append(_za_member(struct.pack("<BBHI", 3, 1, len(part), lo) + part.tobytes()))
        nm += 1
    body = b"".join(out)
    loc = _za_member(struct.pack("<BBHIIQQ", 3, 2, 0, nm, len(allrec), len(body), data_member_bytes))
    assert len(loc) == INDEX_LOCATOR_BYTES
    return body + loc


def parse_index_tail(fp, start, end):
    """The index of a file [start, end) that is ONE data member followed by index members (and, possibly, the plain empty member
    the reference's close() leaves): -> (in_len, out_len, rows) as numpy arrays (rows: units x INDEX_STRIDE u32), or None.  The
    file position is left where it was."""
    import struct
    here = fp.tell()
    try:
        tail_skip = 0
        for attempt in (0, 1):
            if end - start < INDEX_LOCATOR_BYTES + tail_skip + 20:
                return None
            fp.seek(end - tail_skip - INDEX_LOCATOR_BYTES)
            loc = fp.read(INDEX_LOCATOR_BYTES)
            if len(loc) == INDEX_LOCATOR_BYTES and loc[:4] == b"\x1f\x8b\x08\x04" and loc[12:14] == b"ZA" and loc[16] == 3 and loc[17] == 2:
                break
            if attempt == 0:                          # behind a trailing plain empty member?
                fp.seek(end - 20)
                plain = fp.read(20)
                if len(plain) == 20 and plain[:4] == b"\x1f\x8b\x08\x00" and plain[10:] == b"\x03\x00" + bytes(8):
                    tail_skip = 20
                    continue
            return None
        _v, _k, _p, nm, nu, ibytes, dbytes = struct.unpack("<BBHIIQQ", loc[16:44])
        if loc[44:] != b"\x03\x00" + bytes(8):
            return None
        ix0 = end - tail_skip - INDEX_LOCATOR_BYTES - ibytes
        if ix0 - dbytes != start or nu == 0 or nu > (1 << 26) or ibytes > (1 << 34):
            return None
        fp.seek(ix0)
        blob = fp.read(ibytes)
        if len(blob) != ibytes:
            return None
        rt = _index_rec_dtype()
        parts, at, seen = [], 0, 0
        for _ in range(nm):
            if blob[at:at + 4] != b"\x1f\x8b\x08\x04" or blob[at + 12:at + 14] != b"ZA":
                return None
            slen = struct.unpack_from("<H", blob, at + 14)[0]
            v, k, n, first = struct.unpack_from("<BBHI", blob, at + 16)
            if v != 3 or k != 1 or first != seen or slen != 8 + n * rt.itemsize:
                return None
            parts.append(np.frombuffer(blob, rt, n, at + 24))
            seen += n
            at += 16 + slen + 10
        if seen != nu or at != ibytes:
            return None
        rec = np.concatenate(parts) if len(parts) != 1 else parts[0]
        if int(rec["in_len"].astype(np.int64).sum()) + 10 + 10 != dbytes:        # header, units, 03 00, trailer
            return None
        rows = np.zeros((nu, INDEX_STRIDE), np.uint32)
        nseg = (rec["out_len"].astype(np.int64) + 2047) >> 11
        cum = np.cumsum(rec["e"].astype(np.uint32), axis=1, dtype=np.uint32)
        cum[np.arange(65)[None, :] > nseg[:, None]] = 0
        cum[rec["e"].max(axis=1) == 0] = 0                          # stored units
        rows[:, :65] = cum
        return np.ascontiguousarray(rec["in_len"]), np.ascontiguousarray(rec["out_len"]), rows
    except (OSError, ValueError, struct.error):
        return None
    finally:
        try:
            fp.seek(here)
        except (OSError, ValueError):
            pass


def block_table(blocks):
    """(ctypes table, count) of a list of (off, len, dict_len, flags) for Context.deflate_blocks."""
    n = len(blocks)
    arr = (Block * max(n, 1))()
    for i, (off, ln, dl, fl) in enumerate(blocks):
        if ln > 0xFFFFFFFF or dl > 0xFFFFFFFF:            # ctypes would cut the u32 fields silently
            raise OverflowError("a block is limited to 4 GiB - 1 bytes: split it")
        arr[i] = Block(off, ln, dl, fl, 0)
    return arr, n


class Context:
    """One engine context = one GPU + one HIP stream + grow-only device workspaces."""

    def __init__(self, device=None):
        L = load()
        if device is None:
            device = int(os.environ.get("ZNGAMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = L.zngamd_device_count()
            if n > 0:
                device %= n
        h = C.c_void_p()
        r = L.zngamd_ctx_create(device, C.byref(h))
        if r != OK:
            raise RuntimeError(
                f"zng_amd: no usable GPU (zngamd_ctx_create({device}) -> {r}); this engine has no CPU path")
        self.L, self.h, self.device = L, h, device
        self._scratch, self._scratch_lock = None, threading.Lock()
        self._tls = threading.local()

    @property
    def last_needed(self):
        """Room the calling thread's last inflate call asked for (0 = it fitted); per thread, like the engine's message."""
        return getattr(self._tls, "needed", 0)

    @last_needed.setter
    def last_needed(self, v):
        self._tls.needed = v

    def close(self):
        if self.h:
            self.L.zngamd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def err(self):
        return self.L.zngamd_last_error(self.h).decode("utf-8", "replace")

    def _chk(self, r, ok=(OK,)):
        if r not in ok:
            raise EngineError(r, self.err())
        return r

    def sync(self):
        """wait for everything queued on the context's stream"""
        self._chk(self.L.zngamd_sync(self.h))

    # ---- checksums
    def crc32(self, data, value=0):
        p, keep = _addr(data)
        out = C.c_uint32(0)
        self._chk(self.L.zngamd_crc32(self.h, value & 0xFFFFFFFF, p, memoryview(data).nbytes, C.byref(out)))
        return out.value

    def adler32(self, data, value=1):
        p, keep = _addr(data)
        out = C.c_uint32(0)
        self._chk(self.L.zngamd_adler32(self.h, value & 0xFFFFFFFF, p, memoryview(data).nbytes, C.byref(out)))
        return out.value

    def crc32_combine(self, crc1, crc2, len2):
        return self.L.zngamd_crc32_combine(crc1 & 0xFFFFFFFF, crc2 & 0xFFFFFFFF, len2)

    # ---- deflate
    def deflate_blocks(self, buf, blocks, level, out_cap, joined=False, into=None, index=False):
        """blocks: list of (off, len, dict_len, flags), or a table made by block_table() (a writer whose batches have the
        same shape makes it once).  -> (list of bytes|None, list of crc, overflowed); with joined=True the first element is
        ONE object, the blocks' outputs back to back (a writer that only concatenates them saves the allocation and release
        of one object per block) and a list of lengths is appended to the result: the engine copies the packed stream
        straight into it (zngamd_deflate_blocks_packed).  `into` (joined only): a bytearray of at least n * out_cap bytes
        that takes the output -- the result is then a memoryview of its filled part (a writer that keeps two of them
        writes into warm memory; a fresh object costs a page fault per 4 KiB).  `index` (joined only): the segment-index records
        of the call's units come back as a fifth element (None after an overflow) -- from the SAME engine call
        (zngamd_deflate_blocks_packed_indexed): a context that several writer threads share answers zngamd_deflate_index for its
        last deflate call, whoever made it."""
        arr, n = blocks if isinstance(blocks, tuple) and len(blocks) == 2 and isinstance(blocks[0], C.Array) else block_table(blocks)
        p, keep = _addr(buf)
        lens = (C.c_uint32 * max(n, 1))()
        crcs = (C.c_uint32 * max(n, 1))()
        need = max(n, 1) * out_cap
        if joined:
            total = C.c_uint64(0)
            if index:
                nu = int(self.L.zngamd_count_units(arr, n))
                uin = np.empty(max(nu, 1), np.uint32); uout = np.empty(max(nu, 1), np.uint32); rows = np.empty((max(nu, 1), INDEX_STRIDE), np.uint32)
                tail = (nu, uin.ctypes.data_as(C.c_void_p), uout.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p))
                call = self.L.zngamd_deflate_blocks_packed_indexed
            else:
                tail = ()
                call = self.L.zngamd_deflate_blocks_packed
            if into is not None and len(into) >= need:
                anchor = C.c_char.from_buffer(into)
                r = call(self.h, p, memoryview(buf).nbytes, arr, n, level,
                         C.cast(C.addressof(anchor), C.c_void_p), len(into), out_cap,
                         C.cast(lens, C.c_void_p), C.cast(crcs, C.c_void_p), C.byref(total), *tail)
                del anchor
                self._chk(r, (OK, E_OVERFLOW))
                res = None if r == E_OVERFLOW else memoryview(into)[:total.value]
            else:
                out = _Out(need)
                r = call(self.h, p, memoryview(buf).nbytes, arr, n, level, out.addr(), need, out_cap,
                         C.cast(lens, C.c_void_p), C.cast(crcs, C.c_void_p), C.byref(total), *tail)
                self._chk(r, (OK, E_OVERFLOW))
                res = None if r == E_OVERFLOW else out.take(total.value)
            if index:
                return res, list(crcs[:n]), r == E_OVERFLOW, list(lens[:n]), (None if r == E_OVERFLOW else self._index_records(nu, uin[:nu], uout[:nu], rows[:nu]))
            return res, list(crcs[:n]), r == E_OVERFLOW, list(lens[:n])
        # The per-block outputs land in a buffer that lives with the context: device-to-host copies into memory that has
        # been touched (and registered by the runtime) before run several times faster than into fresh pages.  The lock
        # covers the call and the slicing (engine calls of one context are serial anyway).
        with self._scratch_lock:
            if self._scratch is None or len(self._scratch) < need:
                self._scratch = bytearray(need + need // 4)
            out = self._scratch
            arr_out = (C.c_char * len(out)).from_buffer(out)
            r = self.L.zngamd_deflate_blocks(self.h, p, memoryview(buf).nbytes, arr, n, level,
                                             C.cast(arr_out, C.c_void_p), out_cap, C.cast(lens, C.c_void_p), C.cast(crcs, C.c_void_p))
            del arr_out
            self._chk(r, (OK, E_OVERFLOW))
            res = []
            mv = memoryview(out)
            for i in range(n):
                if lens[i] == 0xFFFFFFFF:
                    res.append(None)
                else:
                    res.append(bytes(mv[i * out_cap:i * out_cap + lens[i]]))
            del mv
        return res, list(crcs[:n]), r == E_OVERFLOW

    def deflate_stream(self, data, level, window_bits=15, prefix=b"", trailer=None):
        """-> (prefix + raw deflate bytes + trailer(crc32, adler32), crc32, adler32).  The container's header and trailer are
        written into the result object itself, around the bytes the engine puts there: no second copy of the payload."""
        p, keep = _addr(data)
        n = memoryview(data).nbytes
        cap = n + (n // 16384 + 1) * 16 + 192          # (a call of up to 128 KiB is cut into units of 16 KiB: 10 bytes each at worst, stored)
        room = len(prefix) + (8 if trailer is not None else 0)
        out = _Out(cap + room)
        base = out.addr().value
        ol = C.c_uint64(0)
        crc, ad = C.c_uint32(0), C.c_uint32(1)
        self._chk(self.L.zngamd_deflate_stream(self.h, p, n, level, window_bits, C.c_void_p(base + len(prefix)), cap,
                                               C.byref(ol), C.byref(crc), C.byref(ad)))
        total = len(prefix) + ol.value
        if prefix:
            C.memmove(base, prefix, len(prefix))
        if trailer is not None:
            t = trailer(crc.value, ad.value)
            C.memmove(base + total, t, len(t))
            total += len(t)
        return out.take(total), crc.value, ad.value

    def debug_keep(self, on=True):
        self._chk(self.L.zngamd_debug_keep(self.h, 1 if on else 0))

    def debug_fetch(self, what, unit, nbytes):
        b = C.create_string_buffer(nbytes)
        self._chk(self.L.zngamd_debug_fetch(self.h, what, unit, C.cast(b, C.c_void_p), nbytes))
        return b.raw

    # ---- inflate
    def inflate_raw(self, data, out_cap, zdict=b""):
        """-> (code, out bytes, in_used, crc32, adler32)"""
        p, keep = _addr(data)
        dp, dkeep = _addr(zdict) if len(zdict) else (None, None)
        out = _Out(out_cap)
        ol, used = C.c_uint64(0), C.c_uint64(0)
        crc, ad = C.c_uint32(0), C.c_uint32(1)
        r = self.L.zngamd_inflate_raw(self.h, p, memoryview(data).nbytes, dp, len(zdict),
                                      out.addr(), out_cap, C.byref(ol), C.byref(used),
                                      C.byref(crc), C.byref(ad))
        if r in (E_HIP, E_ARG):
            raise EngineError(r, self.err())
        # BUF_ERROR with a size above the capacity = "this is how much room the stream needs" (nothing was copied)
        self.last_needed = ol.value if (r == BUF_ERROR and ol.value > out_cap) else 0
        if self.last_needed:
            return r, b"", 0, 0, 1
        return r, out.take(min(ol.value, out_cap)), used.value, crc.value, ad.value

    def inflate_resume(self, data, start_bit, zdict, out_cap):
        """-> (code, out bytes, in_bits, block_bits, block_out); code E_OVERFLOW = out_cap reached"""
        p, keep = _addr(data)
        dp, dkeep = _addr(zdict) if len(zdict) else (None, None)
        out = _Out(out_cap)
        ol, ib, bb, bo = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        r = self.L.zngamd_inflate_resume(self.h, p, memoryview(data).nbytes, start_bit, dp, len(zdict),
                                         out.addr(), out_cap, C.byref(ol), C.byref(ib), C.byref(bb), C.byref(bo))
        if r in (E_HIP, E_ARG):
            raise EngineError(r, self.err())
        return r, out.take(min(ol.value, out_cap)), ib.value, bb.value, bo.value

    def gunzip(self, data, out_cap):
        """-> (code, out bytes, n_members)"""
        p, keep = _addr(data)
        out = _Out(out_cap)
        ol, nm = C.c_uint64(0), C.c_uint32(0)
        r = self.L.zngamd_gunzip(self.h, p, memoryview(data).nbytes, out.addr(), out_cap,
                                 C.byref(ol), C.byref(nm))
        if r in (E_HIP, E_ARG):
            raise EngineError(r, self.err())
        # BUF_ERROR with a size above the capacity = "this is how much room the stream needs"
        self.last_needed = ol.value if (r == BUF_ERROR and ol.value > out_cap) else 0
        return r, out.take(min(ol.value, out_cap)), nm.value

    def gunzip_partial(self, data, out_cap):
        """Window of a longer stream -> (code, out bytes, n_members, in_consumed); see zngamd_gunzip_partial."""
        p, keep = _addr(data)
        out = _Out(out_cap)
        ol, nm, used = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        r = self.L.zngamd_gunzip_partial(self.h, p, memoryview(data).nbytes, out.addr(), out_cap,
                                         C.byref(ol), C.byref(nm), C.byref(used))
        if r in (E_HIP, E_ARG):
            raise EngineError(r, self.err())
        self.last_needed = ol.value if (r == BUF_ERROR and ol.value > out_cap) else 0
        return r, out.take(min(ol.value, out_cap)), nm.value, used.value

    def gunzip_stream(self, state, data, out_cap, last, view=False, into=None):
        """Stateful window of a longer stream -> (code, out bytes, n_members, in_consumed); see zngamd_gunzip_stream."""
        p, keep = _addr(data)
        if into is not None:                    # (object, address) from new_buffer(): reused from window to window, so
            out, op = into                      # its pages are faulted in once
            view = True
        else:
            out, op = _new_bytes(out_cap)
        ol, nm, used = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        r = self.L.zngamd_gunzip_stream(self.h, C.byref(state), p, memoryview(data).nbytes, 1 if last else 0, op, out_cap,
                                        C.byref(ol), C.byref(nm), C.byref(used))
        if r in (E_HIP, E_ARG):
            raise EngineError(r, self.err())
        self.last_needed = ol.value if (r == BUF_ERROR and ol.value > out_cap) else 0
        n = min(ol.value, out_cap)
        if view:                                # the caller only reads from it: no copy of the filled part
            return r, memoryview(out)[:n], nm.value, used.value
        return r, _take(out, n), nm.value, used.value

    @staticmethod
    def gzip_members_room(n, block_size):
        """Bytes that always hold the members of n bytes of input."""
        nb = max(1, (n + block_size - 1) // max(block_size, 1))
        return n + n // 16 + nb * 2800 + 64         # index: 4 bytes per 256 bytes of input; flat headers; stored worst case

    def gzip_members(self, data, block_size, level, into=None):
        """One indexed gzip member per block_size bytes of `data`; into = a buffer of the caller's with gzip_members_room()
        bytes (warm memory: a fresh result object costs a page fault per 4 KiB): the result is then a view of it."""
        p, keep = _addr(data)
        n = memoryview(data).nbytes
        cap = self.gzip_members_room(n, block_size)
        ol = C.c_uint64(0)
        if into is not None and len(into) >= cap:
            anchor = C.c_char.from_buffer(into)
            r = self.L.zngamd_gzip_members(self.h, p, n, block_size, level, C.cast(C.addressof(anchor), C.c_void_p), len(into), C.byref(ol))
            del anchor
            self._chk(r)
            return memoryview(into)[:ol.value]
        out = _Out(cap)
        self._chk(self.L.zngamd_gzip_members(self.h, p, n, block_size, level, out.addr(), cap, C.byref(ol)))
        return out.take(ol.value)

    # ---- the writer's segment index (dict-chained streams written with FLAG_FLATHDR)
    def deflate_index(self, n_units):
        """The segment index of this context's LAST deflate call: n_units rows of INDEX_STRIDE u32, as a device buffer
        (zngamd_deflate_index_dev)."""
        from . import devmem
        d = devmem.empty(self, 4 * INDEX_STRIDE * max(1, n_units))
        self._chk(self.L.zngamd_deflate_index_dev(self.h, d.vp(), n_units))
        return d

    def inflate_units_indexed_dev(self, d_def, def_len, unit_in_len, unit_out_len, d_index, d_out, out_cap, d_dict=None, dict_len=0):
        """zngamd_inflate_units_indexed_dev: unit-parallel decode of ONE dict-chained stream of this engine with its index.
        d_def / d_index / d_out / d_dict: device pointers (ints or c_void_p); unit_*_len: sequences of ints.
        -> (code, out_len); code STREAM_END, or E_INDEX / DATA_ERROR / BUF_ERROR as the C call returns them."""
        n = len(unit_in_len)
        # (numpy arrays go to the engine as they are: a list of 32 768 sizes costs a millisecond to turn into a C array)
        ka = np.ascontiguousarray(unit_in_len, dtype=np.uint32)
        kb = np.ascontiguousarray(unit_out_len, dtype=np.uint32)
        a = ka.ctypes.data_as(C.POINTER(C.c_uint32))
        b = kb.ctypes.data_as(C.POINTER(C.c_uint32))
        ol = C.c_uint64(0)
        r = self.L.zngamd_inflate_units_indexed_dev(self.h, C.c_void_p(int(d_def)), def_len, a, b, n, C.c_void_p(int(d_index)),
                                                    C.c_void_p(int(d_dict)) if d_dict else None, dict_len,
                                                    C.c_void_p(int(d_out)), out_cap, C.byref(ol))
        return r, ol.value

    def deflate_index_records(self, blocks):
        """The units of this context's LAST deflate call (made with `blocks`: a list or a block_table) as index records
        (INDEX_REC: compressed bytes, output bytes, 65 two-byte entries -- the first segment's bit offset, then every segment's
        length in bits; all zeros: a unit of stored blocks)."""
        arr, n = blocks if isinstance(blocks, tuple) and len(blocks) == 2 and isinstance(blocks[0], C.Array) else block_table(blocks)
        nu = int(self.L.zngamd_count_units(arr, n))
        uin = np.empty(nu, np.uint32); uout = np.empty(nu, np.uint32); rows = np.empty((nu, INDEX_STRIDE), np.uint32)
        self._chk(self.L.zngamd_deflate_index(self.h, nu, uin.ctypes.data_as(C.c_void_p), uout.ctypes.data_as(C.c_void_p),
                                              rows.ctypes.data_as(C.c_void_p)))
        return self._index_records(nu, uin, uout, rows)

    @staticmethod
    def _index_records(nu, uin, uout, rows):
        rec = np.zeros(nu, _index_rec_dtype())
        rec["in_len"], rec["out_len"] = uin, uout
        nseg = (uout.astype(np.int64) + 2047) >> 11
        e = rows[:, :65].astype(np.int64)
        d = np.diff(e, axis=1, prepend=0)
        d[np.arange(65)[None, :] > nseg[:, None]] = 0            # (entries behind the end-of-block one mean nothing)
        if d.min(initial=0) < 0 or d.max(initial=0) > 0xFFFF:
            raise EngineError(E_ARG, "segment index out of range")
        rec["e"] = d.astype(np.uint16)
        return rec

    # ---- measurement
    def profiling(self, on):
        self._chk(self.L.zngamd_profiling(self.h, 1 if on else 0))

    def kernel_times(self, reset=True):
        # the library says how many it writes (a library older than the query -- a variant build of an earlier round under
        # ZNGAMD_LIB -- wrote as many as this table is long)
        nk = max(len(K_NAMES), int(self.L.zngamd_kernel_class_count())) if hasattr(self.L, "zngamd_kernel_class_count") else len(K_NAMES)
        ms = (C.c_double * nk)()
        ln = (C.c_uint64 * nk)()
        self._chk(self.L.zngamd_kernel_times(self.h, ms, ln, 1 if reset else 0))
        return {k: (ms[i], ln[i]) for i, k in enumerate(K_NAMES)}

    def decode_paths(self, reset=True):
        """Members gunzip() decoded per path since the last reset: indexed, bgzf, chunked, sequential."""
        m = (C.c_uint64 * 4)()
        self._chk(self.L.zngamd_decode_paths(self.h, m, 1 if reset else 0))
        return dict(zip(("indexed", "bgzf", "chunked", "sequential"), (int(x) for x in m)))


_default = None
_default_lock = threading.Lock()
_pool = {}


def writer_devices(limit=None):
    """Devices a block-parallel writer of THIS process spreads its batches over: ZNGAMD_DEVICES ("0,1,2,3"; a device may be
    named twice, which gives two contexts on it) or every visible GPU, cut to `limit` entries.  A process that was given
    one GPU by its launcher (LOCAL_RANK / ZNGAMD_DEVICE set) stays on it."""
    spec = os.environ.get("ZNGAMD_DEVICES")
    if spec:
        devs = [int(x) for x in spec.replace(";", ",").split(",") if x.strip() != ""]
    elif "ZNGAMD_DEVICE" in os.environ or "LOCAL_RANK" in os.environ:
        devs = [default_context().device]
    else:
        devs = list(range(max(1, load().zngamd_device_count())))
    if limit is not None:
        devs = devs[:max(1, limit)]
    return devs


def contexts(limit=None):
    """One context per entry of writer_devices(limit); entry 0 is the process-wide default context when it names its device.
    Contexts live as long as the process (their device workspaces are grow-only)."""
    out = []
    with _default_lock:
        pass
    seen = {}
    for d in writer_devices(limit):
        k = (d, seen.get(d, 0))
        seen[d] = seen.get(d, 0) + 1
        with _default_lock:
            c = _pool.get(k)
        if c is None:
            if k[1] == 0 and default_context().device == d:
                c = default_context()
            else:
                c = Context(device=d)
            with _default_lock:
                c = _pool.setdefault(k, c)
        out.append(c)
    return out


def deflate_blocks_multi(ctxs, buf, blocks, level, out_cap, into=None, table=None, index=None):
    """deflate_blocks(joined=True) over several contexts: the blocks are cut into contiguous ranges of about equal input,
    one per context; every range goes to its GPU as the slice of `buf` it needs (its blocks and the dictionary in front of
    its first block -- the previous range's input tail), the ranges run side by side (the engine calls release the GIL) and
    the results come back in block order.  This is the reference's worker fan-out (gzip_ng_threaded.py:233-246, :316-321)
    with contiguous ranges in place of round-robin, and its in-order drain (:382-398).
    -> (packed bytes, crcs, overflowed, lens) like Context.deflate_blocks(joined=True); `into` and `table` (block_table(blocks),
    made once by a caller whose batches repeat) serve the one-GPU case."""
    n = len(blocks)
    g = min(len(ctxs), n)
    # `index` (a list, or None): the batch's segment-index records are appended to it, in unit order (Context.deflate_index_records
    # of every range, taken right behind the range's engine call on its own context)
    def one():
        r = ctxs[0].deflate_blocks(buf, table if table is not None else blocks, level, out_cap, joined=True, into=into, index=index is not None)
        if index is not None and r[4] is not None:
            index.append(r[4])
        return r[:4]
    if g <= 1 or sys.is_finalizing():      # (no new threads while the interpreter shuts down)
        return one()
    total = sum(b[1] for b in blocks)
    if total < (8 << 20):
        return one()
    mv = memoryview(buf)
    if mv.format != "B" or mv.ndim != 1:
        mv = mv.cast("B")
    # contiguous ranges by cumulative input bytes
    cuts, acc, k = [0], 0, 1
    for i, b in enumerate(blocks):
        acc += b[1]
        if k < g and acc * g >= total * k and i + 1 < n:
            cuts.append(i + 1)
            k += 1
    cuts.append(n)
    parts = [None] * (len(cuts) - 1)
    recs = [None] * (len(cuts) - 1)
    errs = []

    def run(j):
        try:
            sub = blocks[cuts[j]:cuts[j + 1]]
            lo = min(o - d for o, _, d, _ in sub)
            hi = max(o + ln for o, ln, _, _ in sub)
            rel = [(o - lo, ln, d, f) for o, ln, d, f in sub]
            r = ctxs[j].deflate_blocks(mv[lo:hi], rel, level, out_cap, joined=True, index=index is not None)
            parts[j] = r[:4]
            if index is not None:
                recs[j] = r[4]
        except BaseException as exc:                      # raised in the caller's thread below
            errs.append(exc)
    ths = [threading.Thread(target=run, args=(j,), name=f"zng-amd-gpu{j}") for j in range(1, len(parts))]
    for t in ths:
        t.start()
    run(0)
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    if index is not None and all(r is not None for r in recs):
        index.extend(recs)
    packed = b"".join(p[0] for p in parts)
    crcs = [c for p in parts for c in p[1]]
    lens = [x for p in parts for x in p[3]]
    return packed, crcs, any(p[2] for p in parts), lens


def default_context():
    """Process-wide context on the GPU chosen by ZNGAMD_DEVICE / LOCAL_RANK (default 0)."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Context()
        return _default
