"""oracle/ -- TEST INFRASTRUCTURE ONLY.

ctypes face of libza_oracle.so (the plain-C CPU restatement, see oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ZA_ORACLE_SO") or os.path.join(_HERE, "libza_oracle.so")      # ZA_ORACLE_SO: the sanitizer build (make asan)

SEG = 2048
MAX_UNIT = 131072
MAX_SEGS = 64
WIN = 32768
FLAG_FINAL = 1
FLAG_FLATHDR = 2
CHUNK_SHIFT = 11
MAX_CHUNKS = MAX_UNIT >> CHUNK_SHIFT

OK, STREAM_END, NEED_DICT = 0, 1, 2
STREAM_ERROR, DATA_ERROR, MEM_ERROR, BUF_ERROR = -2, -3, -4, -5
GZ_BAD_MAGIC, GZ_BAD_METHOD, GZ_BAD_HCRC, GZ_BAD_CRC, GZ_BAD_LENGTH, GZ_TRUNCATED = \
    -101, -102, -103, -104, -105, -106


def build(force=False):
    if os.environ.get("ZA_ORACLE_SO"):
        return _SO
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libza_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class _Debug(C.Structure):
    _fields_ = [("prevdist", C.c_void_p), ("best", C.c_void_p), ("tokens", C.c_void_p),
                ("seg_ntok", C.c_void_p), ("hist", C.c_void_p), ("lens", C.c_void_p),
                ("seg_bits", C.c_void_p), ("btype", C.c_void_p), ("chunk_idx", C.c_void_p),
                ("linkB", C.c_void_p), ("linkC", C.c_void_p), ("best_dp", C.c_void_p), ("dp_cost", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.za_o_crc32.restype = C.c_uint32
        L.za_o_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.za_o_adler32.restype = C.c_uint32
        L.za_o_adler32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.za_o_crc32_combine.restype = C.c_uint32
        L.za_o_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.za_o_deflate_unit.restype = C.c_long
        L.za_o_deflate_unit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32),
                                        C.POINTER(_Debug)]
        L.za_o_deflate_stream.restype = C.c_long
        L.za_o_deflate_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                          C.c_void_p, C.c_size_t]
        L.za_o_inflate_raw.restype = C.c_int
        L.za_o_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.za_o_gunzip.restype = C.c_int
        L.za_o_gunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.za_o_zlib_decompress.restype = C.c_int
        L.za_o_zlib_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        L.za_o_bench_blocks.restype = C.c_int
        L.za_o_bench_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def crc32(data, value=0):
    return lib().za_o_crc32(value & 0xFFFFFFFF, bytes(data), len(data))


def adler32(data, value=1):
    return lib().za_o_adler32(value & 0xFFFFFFFF, bytes(data), len(data))


def crc32_combine(crc1, crc2, len2):
    return lib().za_o_crc32_combine(crc1, crc2, len2)


def seg_shift(n, flags=0):
    """log2 of the segment size of a unit of n bytes (oracle.h za_o_seg_shift)"""
    return lib().za_o_seg_shift(int(n), int(flags))


def deflate_unit(data, zdict=b"", level=6, flags=0, debug=False, cap=None):
    """One codec unit (<= 128 KiB) primed with `zdict` (<= 32 KiB): -> (bytes, crc[, debug dict])."""
    data = bytes(data)
    zdict = bytes(zdict)[-WIN:]
    n, dl = len(data), len(zdict)
    buf = np.frombuffer(zdict + data + b"\0" * 8, dtype=np.uint8).copy()
    if cap is None:
        cap = n + n // 8 + 600
    out = np.zeros(cap, dtype=np.uint8)
    crc = C.c_uint32(0)
    dbg = None
    arrs = {}
    if debug:
        seg = 1 << seg_shift(n, flags)
        nseg = (n + seg - 1) // seg
        arrs = dict(prevdist=np.zeros(dl + n, np.uint16), best=np.zeros(max(n, 1), np.uint32),
                    tokens=np.zeros(max(nseg * seg, 1), np.uint32),
                    seg_ntok=np.zeros(MAX_SEGS, np.uint32), hist=np.zeros(320, np.uint32),
                    lens=np.zeros(320, np.uint8), seg_bits=np.zeros(MAX_SEGS + 1, np.uint32),
                    btype=np.zeros(1, np.int32), chunk_idx=np.zeros(MAX_CHUNKS + 1, np.uint32),
                    linkB=np.zeros(dl + n, np.uint16), linkC=np.zeros(dl + n, np.uint16),
                    best_dp=np.zeros(max(n, 1), np.uint32), dp_cost=np.zeros(258, np.uint32))
        dbg = _Debug(*[arrs[k].ctypes.data for k in
                       ("prevdist", "best", "tokens", "seg_ntok", "hist", "lens", "seg_bits", "btype", "chunk_idx",
                        "linkB", "linkC", "best_dp", "dp_cost")])
    r = lib().za_o_deflate_unit(buf.ctypes.data + dl, dl, n, level, flags, out.ctypes.data, cap,
                                C.byref(crc), C.byref(dbg) if dbg is not None else None)
    if r < 0:
        raise RuntimeError(f"oracle deflate_unit failed: {r}")
    res = out[:r].tobytes()
    if debug:
        arrs["best"] = arrs["best"][:n]
        arrs["best_dp"] = arrs["best_dp"][:n]
        arrs["btype"] = int(arrs["btype"][0])
        return res, crc.value, arrs
    return res, crc.value


def deflate_stream(data, level=6, flags=FLAG_FINAL, window_bits=15):
    data = bytes(data)
    lib().za_o_set_max_dist(1 << window_bits)
    cap = len(data) + len(data) // 8 + 1024
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().za_o_deflate_stream(data, len(data), level, flags, out.ctypes.data, cap)
    lib().za_o_set_max_dist(WIN)
    if r < 0:
        raise RuntimeError(f"oracle deflate_stream failed: {r}")
    return out[:r].tobytes()


def inflate_raw(data, out_cap, zdict=b""):
    """-> (code, bytes_out, in_used)"""
    data = bytes(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    used, got = C.c_size_t(0), C.c_size_t(0)
    zdict = bytes(zdict)
    r = lib().za_o_inflate_raw(data, len(data), out.ctypes.data, out_cap,
                               zdict if zdict else None, len(zdict), C.byref(used), C.byref(got))
    return r, out[:got.value].tobytes(), used.value


def gunzip(data, out_cap):
    """-> (code, bytes_out, n_members)"""
    data = bytes(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    got, nm = C.c_size_t(0), C.c_int(0)
    r = lib().za_o_gunzip(data, len(data), out.ctypes.data, out_cap, C.byref(got), C.byref(nm))
    return r, out[:got.value].tobytes(), nm.value


def zlib_decompress(data, out_cap):
    data = bytes(data)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    got = C.c_size_t(0)
    r = lib().za_o_zlib_decompress(data, len(data), out.ctypes.data, out_cap, C.byref(got))
    return r, out[:got.value].tobytes()


def bench_blocks(arr, block, level, threads):
    """arr: contiguous np.uint8.  -> (t_deflate_s, t_inflate_s, compressed_bytes)"""
    td, ti, cb = C.c_double(0), C.c_double(0), C.c_size_t(0)
    r = lib().za_o_bench_blocks(arr.ctypes.data, arr.size, block, level, threads,
                                C.byref(td), C.byref(ti), C.byref(cb))
    if r != 0:
        raise RuntimeError(f"oracle bench failed: {r}")
    return td.value, ti.value, cb.value
