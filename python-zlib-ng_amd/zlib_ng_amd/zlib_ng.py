"""zlib_ng -- drop-in face of the reference's C module `zlib_ng.zlib_ng`
(reference src/zlib_ng/zlib_ngmodule.c, stub src/zlib_ng/zlib_ng.pyi:41-99), served by the MI355X engine.

Every byte of DEFLATE, inflate, CRC-32 and Adler-32 work is done by HIP kernels behind the C ABI
(include/zng_amd.h).  The Python code here only builds/validates container framing (RFC 1950 / 1952
headers and trailers), maps status codes to the reference's exception types and messages
(zlib_ngmodule.c:68-95, :2458-2629) and moves buffers.  There is no CPU fallback: without the library
and a GPU these functions raise RuntimeError.

Implemented: compress / decompress (a10), crc32 / adler32 /
crc32_combine (a4, a5, a12), _ParallelCompress (a1, a2), _GzipReader (a8), compressobj (buffering
writer used by gzip_ng.GzipNGFile), decompressobj / _ZlibDecompressor (SURVEY.md section 8f rank 2:
block-resumable decoding on the sequential wavefront decoder).
"""
import gzip as _gzip
import io as _io
import operator as _operator
import os as _os
import struct as _struct
import sys as _sys
import threading as _threading

from . import _lib

# ---- constants (zlib_ngmodule.c:3030-3069) ----------------------------------------------------------
MAX_WBITS = 15
DEFLATED = 8
DEF_MEM_LEVEL = 8
DEF_BUF_SIZE = 16 * 1024
Z_NO_COMPRESSION, Z_BEST_SPEED, Z_BEST_COMPRESSION, Z_DEFAULT_COMPRESSION = 0, 1, 9, -1
Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED, Z_DEFAULT_STRATEGY = 1, 2, 3, 4, 0
Z_NO_FLUSH, Z_PARTIAL_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_FINISH, Z_BLOCK, Z_TREES = 0, 1, 2, 3, 4, 5, 6
ZLIB_VERSION = "1.2.12"
ZLIB_RUNTIME_VERSION = "1.2.12"
ZLIBNG_VERSION = "zng_amd-0.1"
ZLIBNG_RUNTIME_VERSION = ZLIBNG_VERSION

BadGzipFile = _gzip.BadGzipFile


class error(Exception):
    """Raised for codec errors (mirrors zlib_ng.error, zlib_ngmodule.c:3012-3017)."""


error.__module__ = "zlib_ng"

_MSG = {_lib.BUF_ERROR: "incomplete or truncated stream", _lib.STREAM_ERROR: "inconsistent stream state",
        _lib.DATA_ERROR: "invalid input data"}


def _zerr(code, while_, detail=None):
    # message shape of zlib_error(), zlib_ngmodule.c:68-95
    msg = detail or _MSG.get(code)
    return error(f"Error {code} {while_}: {msg}" if msg else f"Error {code} {while_}")


def _ctx():
    return _lib.default_context()


def _view(data):
    try:
        mv = memoryview(data)
    except TypeError:
        raise TypeError(f"a bytes-like object is required, not '{type(data).__name__}'") from None
    if not mv.contiguous:
        raise BufferError("memoryview: underlying buffer is not C-contiguous")
    return mv.cast("B") if mv.format != "B" or mv.ndim != 1 else mv


def _ssize(value):
    """An argument Argument Clinic converts with its ssize_t converter (zlib_ngmodule.c: bufsize, max_length, length):
    anything with __index__, OverflowError beyond sys.maxsize."""
    value = _operator.index(value)
    if not (-_sys.maxsize - 1 <= value <= _sys.maxsize):
        raise OverflowError("Python int too large to convert to C ssize_t")
    return value


# ---- checksums (zlib_ngmodule.c:1455-1596) -----------------------------------------------------------
def crc32(data, value=0):
    return _ctx().crc32(_view(data), int(value) & 0xFFFFFFFF)


def adler32(data, value=1):
    return _ctx().adler32(_view(data), int(value) & 0xFFFFFFFF)


def crc32_combine(crc1, crc2, crc2_length):
    return _lib.load().zngamd_crc32_combine(int(crc1) & 0xFFFFFFFF, int(crc2) & 0xFFFFFFFF, int(crc2_length))


# ---- one-shot compress / decompress (zlib_ngmodule.c:199-373, :1816-1871) ------------------------------
def _check_level(level):
    if not isinstance(level, int):
        raise TypeError(f"an integer is required (got type {type(level).__name__})")
    if not (-1 <= level <= 9):
        raise error("Bad compression level")


def _container(wbits):
    """-> (kind, window_bits) for the deflate side; zng_deflateInit2 rules."""
    if 9 <= wbits <= 15:
        return "zlib", wbits
    if -15 <= wbits <= -9:
        return "raw", -wbits
    if 25 <= wbits <= 31:
        return "gzip", wbits - 16
    if wbits in (8, -8, 24):            # zlib promotes an 8-bit window to 9 (raw -8 is rejected upstream too)
        if wbits == 8:
            return "zlib", 9
        if wbits == 24:
            return "gzip", 9
    raise error("Bad compression level")    # Z_STREAM_ERROR from init is reported with this text (:224-226)


def _zlib_header(level, window_bits, dictid=None):
    """RFC 1950 header; with a preset dictionary FDICT is set and DICTID (Adler-32 of the dictionary) follows."""
    lv = 6 if level == -1 else level
    flevel = 0 if lv < 2 else 1 if lv < 6 else 2 if lv == 6 else 3
    head = (((window_bits - 8) << 4) | 8) << 8 | (flevel << 6) | (0x20 if dictid is not None else 0)
    head += 31 - head % 31
    out = _struct.pack(">H", head)
    return out if dictid is None else out + _struct.pack(">I", dictid & 0xFFFFFFFF)


def _gzip_header(level):
    lv = 6 if level == -1 else level
    xfl = 2 if lv == 9 else 4 if lv == 1 else 0
    return bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, xfl, 3])


def compress(data, /, level=Z_DEFAULT_COMPRESSION, wbits=MAX_WBITS):
    """Returns a bytes object containing compressed data (zlib_compress_impl, zlib_ngmodule.c:199-273)."""
    mv = _view(data)
    _check_level(level)
    kind, wb = _container(wbits)
    # header and trailer are written into the result object around the engine's bytes (no concatenation: that would copy the
    # whole payload once more)
    if kind == "raw":
        return _ctx().deflate_stream(mv, level, wb)[0]
    if kind == "zlib":
        return _ctx().deflate_stream(mv, level, wb, _zlib_header(level, wb), lambda crc, adler: _struct.pack(">I", adler))[0]
    size = mv.nbytes & 0xFFFFFFFF
    return _ctx().deflate_stream(mv, level, wb, _gzip_header(level), lambda crc, adler: _struct.pack("<II", crc, size))[0]


def _inflate_all(body, zdict=b"", hint=0):
    """Raw inflate with geometric growth of the output buffer -> (code, out, used, crc, adler)."""
    ctx = _ctx()
    cap = max(min(hint, 1032 * len(body) + 64), 4 * len(body), 1 << 16)     # deflate expands at most ~1032:1
    while True:
        code, out, used, crc, ad = ctx.inflate_raw(body, cap, zdict)
        if code == _lib.BUF_ERROR and ctx.last_needed > cap:      # the engine already knows the size
            cap = ctx.last_needed
            continue
        if code == _lib.BUF_ERROR and len(out) >= cap:
            cap *= 4
            continue
        return code, out, used, crc, ad


def _parse_gzip_header(buf, pos=0):
    """-> offset of the first deflate byte; raises BadGzipFile / EOFError like the reference reader."""
    n = len(buf)
    if n - pos < 10:
        raise EOFError("Compressed file ended before the end-of-stream marker was reached")
    if buf[pos:pos + 2] != b"\x1f\x8b":
        raise BadGzipFile(f"Not a gzipped file ({bytes(buf[pos:pos + 2])!r})")
    if buf[pos + 2] != 8:
        raise BadGzipFile("Unknown compression method")
    flags = buf[pos + 3]
    cur = pos + 10
    trunc = EOFError("Compressed file ended before the end-of-stream marker was reached")
    if flags & 4:
        if cur + 2 >= n:
            raise trunc
        cur += 2 + (buf[cur] | buf[cur + 1] << 8)
        if cur >= n:
            raise trunc
    for bit in (8, 16):
        if flags & bit:
            z, span, seen = -1, 4096, cur           # NUL-terminated field: look in growing pieces, not in a copy of the rest
            while z < 0 and seen < n:
                seen = min(n, cur + span)
                z = bytes(buf[cur:seen]).find(b"\0")
                span *= 16
            if z < 0:
                raise trunc
            cur += z + 1
    if flags & 2:
        if cur + 2 >= n:
            raise trunc
        want = buf[cur] | buf[cur + 1] << 8
        got = crc32(bytes(buf[pos:cur])) & 0xFFFF
        if want != got:
            raise BadGzipFile(f"Corrupted gzip header. Checksums do not match: {got:04x} != {want:04x}")
        cur += 2
    return cur


def decompress(data, /, wbits=MAX_WBITS, bufsize=DEF_BUF_SIZE):
    """Returns a bytes object containing the uncompressed data (zlib_decompress_impl, :275-373)."""
    buf = _view(data)                        # a view: the payload is never copied on the host
    bufsize = _ssize(bufsize)
    if bufsize < 0:
        raise ValueError("bufsize must be non-negative")
    W = "while decompressing data"
    if wbits == 0 or 8 <= wbits <= 15:
        kind = "zlib"
    elif -15 <= wbits <= -8:
        kind = "raw"
    elif 24 <= wbits <= 31 or wbits == 16:
        kind = "gzip"
    elif 40 <= wbits <= 47 or wbits == 32:
        kind = "auto"
    else:
        raise _zerr(_lib.STREAM_ERROR, "while preparing to decompress data")
    if kind == "auto":
        kind = "gzip" if buf[:2] == b"\x1f\x8b" else "zlib"
        wbits = wbits - 32                   # 0: take the window from the zlib header
    if kind == "raw":
        code, out, used, _, _ = _inflate_all(buf, hint=bufsize)
        if code == _lib.STREAM_END:
            return out
        raise _zerr(code if code in _MSG else _lib.DATA_ERROR, W)
    if kind == "zlib":
        if len(buf) < 2:
            raise _zerr(_lib.BUF_ERROR, W)
        cmf, flg = buf[0], buf[1]
        if (cmf & 15) != 8 or ((cmf << 8) | flg) % 31:
            raise _zerr(_lib.DATA_ERROR, W, "incorrect header check")
        win = (cmf >> 4) + 8
        if win > 15 or (wbits != 0 and win > (wbits & 15 if wbits > 15 else wbits)):
            raise _zerr(_lib.DATA_ERROR, W, "invalid window size")
        if flg & 0x20:
            raise _zerr(_lib.NEED_DICT, W)
        code, out, used, _, ad = _inflate_all(buf[2:], hint=bufsize)
        if code != _lib.STREAM_END:
            raise _zerr(code if code in _MSG else _lib.DATA_ERROR, W)
        tail = bytes(buf[2 + used:2 + used + 4])
        if len(tail) < 4:
            raise _zerr(_lib.BUF_ERROR, W)
        if _struct.unpack(">I", tail)[0] != ad:
            raise _zerr(_lib.DATA_ERROR, W, "incorrect data check")
        return out
    # gzip, one member (zng_inflate with wbits 16+ stops at the first trailer)
    if len(buf) < 10:
        raise _zerr(_lib.BUF_ERROR, W)
    if buf[:2] != b"\x1f\x8b":
        raise _zerr(_lib.DATA_ERROR, W, "incorrect header check")
    if buf[2] != 8:
        raise _zerr(_lib.DATA_ERROR, W, "unknown compression method")
    if buf[3] & 0xE0:                        # inflate() refuses reserved FLG bits; the gzip reader (like CPython's) does not look
        raise _zerr(_lib.DATA_ERROR, W, "unknown header flags set")
    try:
        start = _parse_gzip_header(buf)
    except EOFError:
        raise _zerr(_lib.BUF_ERROR, W) from None
    except BadGzipFile:
        raise _zerr(_lib.DATA_ERROR, W, "header crc mismatch") from None
    code, out, used, crc, _ = _inflate_all(buf[start:], hint=bufsize)
    if code != _lib.STREAM_END:
        raise _zerr(code if code in _MSG else _lib.DATA_ERROR, W)
    tail = bytes(buf[start + used:start + used + 8])
    if len(tail) < 8:
        raise _zerr(_lib.BUF_ERROR, W)
    tcrc, tlen = _struct.unpack("<II", tail)
    if tcrc != crc:
        raise _zerr(_lib.DATA_ERROR, W, "incorrect data check")
    if tlen != len(out) & 0xFFFFFFFF:
        raise _zerr(_lib.DATA_ERROR, W, "incorrect length check")
    return out


# ---- _ParallelCompress (zlib_ngmodule.c:1598-1797) -------------------------------------------------------
class _ParallelCompress:
    """A reusable block compressor: compress_and_crc(data, zdict) = reset, set the 32 KiB dictionary,
    CRC-32, one deflate ending in a sync flush.  `compress_and_crc_batch` submits many blocks in one
    engine call (what the threaded writer uses instead of one thread per block)."""

    def __init__(self, buffersize, level=Z_DEFAULT_COMPRESSION):
        if not isinstance(buffersize, int) or not isinstance(level, int):
            raise TypeError("an integer is required")
        if buffersize > 0xFFFFFFFF:
            raise ValueError(f"buffersize must be at most {0xFFFFFFFF}, got {buffersize}")
        if not (-1 <= level <= 9):
            raise error("Bad compression level")
        self._buffersize = buffersize
        self._level = level

    def compress_and_crc(self, *args):
        if len(args) != 2:
            raise TypeError(f"compress_and_crc takes exactly 2 arguments, got {len(args)}")
        return self.compress_and_crc_batch([args])[0]

    def compress_and_crc_batch(self, items):
        """items: iterable of (data, zdict) -> list of (compressed bytes, crc32).  When a block's dictionary is exactly the
        tail of the block before it (what the threaded writer produces) it is not copied again: the block is laid directly
        behind its predecessor and primed from there."""
        parts, blocks, pos = [], [], 0
        prev = None                                  # data of the previous item (memoryview)
        for data, zdict in items:
            d, z = _view(data), _view(zdict)
            if d.nbytes + z.nbytes > 0xFFFFFFFF:
                raise OverflowError(f"Can only compress {0xFFFFFFFF} bytes of data")
            z = z[-32768:] if z.nbytes > 32768 else z
            chained = (prev is not None and z.nbytes and z.nbytes == min(prev.nbytes, 32768)
                       and z == prev[prev.nbytes - z.nbytes:])
            if chained:                              # the dictionary is what already lies in front of this block
                parts.append(d)
                blocks.append((pos, d.nbytes, z.nbytes, 0))
                pos += d.nbytes
            else:
                parts.append(z)
                parts.append(d)
                blocks.append((pos + z.nbytes, d.nbytes, z.nbytes, 0))
                pos += z.nbytes + d.nbytes
            prev = d
        if not blocks:
            return []
        outs, crcs, overflowed = _ctx().deflate_blocks(b"".join(parts), blocks, self._level, max(self._buffersize, 1))
        if overflowed:
            raise OverflowError(f"Compressed output exceeds buffer size of {self._buffersize}")
        return list(zip(outs, crcs))


_ParallelCompress.__module__ = "zlib_ng"


# ---- streaming objects: thin bindings of zngamd_stream_* (include/zng_amd.h), laid out like the reference's C methods ----------
import ctypes as _C


class _ZStream(_C.Structure):                # zngamd_stream
    _fields_ = [("next_in", _C.c_void_p), ("avail_in", _C.c_uint32), ("total_in", _C.c_uint64),
                ("next_out", _C.c_void_p), ("avail_out", _C.c_uint32), ("total_out", _C.c_uint64),
                ("msg", _C.c_char_p), ("state", _C.c_void_p), ("adler", _C.c_uint32), ("reserved", _C.c_uint32),
                ("zalloc", _C.c_void_p), ("zfree", _C.c_void_p), ("opaque", _C.c_void_p)]


def _slib():
    L = _lib.load()
    if not getattr(L, "_zs_ready", False):
        P, vp = _C.POINTER(_ZStream), _C.c_void_p
        L.zngamd_stream_deflate_init.argtypes = [vp, P, _C.c_int, _C.c_int, _C.c_int, _C.c_int, _C.c_int]
        L.zngamd_stream_deflate.argtypes = [P, _C.c_int]
        L.zngamd_stream_deflate_set_dictionary.argtypes = [P, vp, _C.c_uint32]
        L.zngamd_stream_deflate_copy.argtypes = [P, P]
        L.zngamd_stream_deflate_end.argtypes = [P]
        L.zngamd_stream_deflate_reset.argtypes = [P]
        L.zngamd_stream_inflate_reset.argtypes = [P]
        L.zngamd_stream_inflate_init.argtypes = [vp, P, _C.c_int]
        L.zngamd_stream_inflate.argtypes = [P, _C.c_int]
        L.zngamd_stream_inflate_set_dictionary.argtypes = [P, vp, _C.c_uint32]
        L.zngamd_stream_inflate_copy.argtypes = [P, P]
        L.zngamd_stream_inflate_end.argtypes = [P]
        L.zngamd_stream_pending.argtypes = [P, _C.POINTER(_C.c_uint64)]
        L.zngamd_stream_inflate_ahead.argtypes = [P, _C.c_uint64]
        L._zs_ready = True
    return L


def _stream_error(zst, err, while_):
    # zlib_error(), zlib_ngmodule.c:68-95: the stream's own message first, then the fallbacks per code
    if err == _lib.MEM_ERROR:
        return MemoryError("Can't allocate memory for (de)compression object" if not zst.msg else zst.msg.decode("utf-8", "replace"))
    return _zerr(err, while_, zst.msg.decode("utf-8", "replace") if zst.msg else None)


class _OutBuf:
    """Growing output buffer of a streaming call (arrange_output_buffer, zlib_ngmodule.c:142-197): starts at `length`,
    doubles, never beyond `limit`.  It is the result object itself (an uninitialised bytes object held through a bare
    pointer, grown and finally cut to size in place like CPython's own _PyBytes_Resize use): no zero fill, no final copy."""

    def __init__(self, zst, length, limit=None):
        self.zst, self.limit = zst, limit
        self.out = _lib._Out(max(1, length if limit is None else min(length, max(limit, 1))))
        self.used = 0
        self._point()

    def _point(self):
        self.base = self.out.addr().value
        self.zst.next_out = self.base + self.used
        self.zst.avail_out = min(self.out.cap - self.used, 0xFFFFFFFF)

    def sync(self):
        self.used = (self.zst.next_out or self.base) - self.base

    def grow(self):
        """Room for more output; False when the limit is reached."""
        self.sync()
        if self.used < self.out.cap:
            self._point()
            return True
        if self.limit is not None and self.out.cap >= self.limit:
            return False
        # doubled as the reference does -- or straight to what the stream holds ready (zngamd_stream_pending), so that a
        # batch of output is copied once instead of through a ladder of doubled buffers
        queued = _C.c_uint64(0)
        _slib().zngamd_stream_pending(_C.byref(self.zst), _C.byref(queued))
        new = max(self.out.cap * 2, self.used + queued.value)
        if self.limit is not None:
            new = min(new, self.limit)
        self.out.resize(new)
        self._point()
        return True

    def result(self):
        self.sync()
        self.zst.next_out = None
        self.zst.avail_out = 0
        return self.out.take(self.used)


_CHUNK = 1 << 30          # input pieces per engine call (arrange_input_buffer cuts at UINT32_MAX, :142-147)


class _Compress:
    """zlib.compressobj look-alike (zlib_ngmodule.c:376-430, :530-575, :718-850) on zngamd_stream_deflate*."""

    def __init__(self, level, method, wbits, memLevel, strategy, zdict):
        L = _slib()
        self._zst = _ZStream()
        self._lock = _threading.Lock()
        self._init = False
        if not isinstance(level, int):
            raise TypeError(f"an integer is required (got type {type(level).__name__})")
        err = L.zngamd_stream_deflate_init(_ctx().h, _C.byref(self._zst), level, method, wbits, memLevel, strategy)
        if err == _lib.MEM_ERROR:
            raise MemoryError("Can't allocate memory for compression object")
        if err != _lib.OK:
            raise ValueError("Invalid initialization option")
        self._init = True
        self._finished = False
        if zdict is not None:
            z = _view(zdict)
            p, keep = _lib._addr(z)
            err = L.zngamd_stream_deflate_set_dictionary(_C.byref(self._zst), p, z.nbytes)
            if err == _lib.STREAM_ERROR:
                raise ValueError("Invalid dictionary")
            if err != _lib.OK:
                raise ValueError("deflateSetDictionary()")

    def __del__(self):
        try:
            if self._init:
                _slib().zngamd_stream_deflate_end(_C.byref(self._zst))
                self._init = False
        except Exception:
            pass

    @property
    def _crc(self):
        """CRC-32 of everything compressed so far (complete once the object has been flushed): gzip_ng.GzipNGFile takes its
        trailer value from here instead of sending every write() to the GPU a second time."""
        return self._zst.adler

    def _run(self, view, flush, while_):
        """Feed `view` (may be None) and collect what deflate hands out; with Z_FINISH until the stream has ended."""
        L, zst = _slib(), self._zst
        out = _OutBuf(zst, DEF_BUF_SIZE)
        n = view.nbytes if view is not None else 0
        base, keep = _lib._addr(view) if n else (None, None)
        pos = 0
        while True:
            ln = min(n - pos, _CHUNK)
            last = pos + ln >= n
            zst.next_in = (base.value + pos) if ln else None
            zst.avail_in = ln
            mode = flush if last else Z_NO_FLUSH
            while True:
                err = L.zngamd_stream_deflate(_C.byref(zst), mode)
                if err not in (_lib.OK, _lib.STREAM_END, _lib.BUF_ERROR):
                    out.result()
                    raise _stream_error(zst, err, while_)
                if zst.avail_out != 0 and zst.avail_in == 0:
                    break
                out.grow()
            pos += ln
            if last:
                break
        return out.result()

    def compress(self, data, /):
        with self._lock:
            view = _view(data)
            if self._finished:
                raise _zerr(_lib.STREAM_ERROR, "while compressing data")
            return self._run(view, Z_NO_FLUSH, "while compressing data")

    def flush(self, mode=Z_FINISH, /):
        mode = _operator.index(mode)
        with self._lock:
            if mode == Z_NO_FLUSH:
                return b""
            if self._finished or not (0 <= mode <= Z_BLOCK):       # deflate() answers Z_STREAM_ERROR to both
                raise _zerr(_lib.STREAM_ERROR, "while flushing")
            res = self._run(None, mode, "while flushing")
            if mode == Z_FINISH:
                self._finished = True
            return res

    def copy(self):
        """zlib_Compress_copy_impl (zlib_ngmodule.c:795-850): an independent compressor in the same state."""
        with self._lock:
            if self._finished:
                raise ValueError("Inconsistent stream state")
            o = _Compress.__new__(_Compress)
            o._zst = _ZStream()
            o._lock = _threading.Lock()
            o._init = False
            err = _slib().zngamd_stream_deflate_copy(_C.byref(o._zst), _C.byref(self._zst))
            if err == _lib.MEM_ERROR:
                raise MemoryError("Can't allocate memory for compression object")
            if err != _lib.OK:
                raise ValueError("Inconsistent stream state")
            o._init = True
            o._finished = False
            return o

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()


_Compress.__module__ = "zlib_ng"


def compressobj(level=Z_DEFAULT_COMPRESSION, method=DEFLATED, wbits=MAX_WBITS, memLevel=DEF_MEM_LEVEL,
                strategy=Z_DEFAULT_STRATEGY, zdict=None):
    return _Compress(level, method, wbits, memLevel, strategy, zdict)


class _InflateStream:
    """Shared part of the two decompressor objects: the zngamd_stream, its dictionary, the inflate loop."""

    def _open(self, wbits, zdict):
        L = _slib()
        if not isinstance(wbits, int):
            raise TypeError(f"'{type(wbits).__name__}' object cannot be interpreted as an integer")
        self._zst = _ZStream()
        self._init = False
        self._zdict = bytes(_view(zdict)) if zdict is not None else b""
        err = L.zngamd_stream_inflate_init(_ctx().h, _C.byref(self._zst), wbits)
        if err == _lib.MEM_ERROR:
            raise MemoryError("Can't allocate memory for decompression object")
        if err != _lib.OK:
            raise ValueError("Invalid initialization option")
        self._init = True
        if self._zdict and wbits < 0:              # raw stream: the dictionary is its history from the start (:490-500)
            self._set_zdict()

    def _set_zdict(self):
        err = _slib().zngamd_stream_inflate_set_dictionary(_C.byref(self._zst), _C.cast(_C.c_char_p(self._zdict), _C.c_void_p), len(self._zdict))
        if err != _lib.OK:
            raise _stream_error(self._zst, err, "while setting zdict")

    def _close(self):
        if getattr(self, "_init", False):
            _slib().zngamd_stream_inflate_end(_C.byref(self._zst))
            self._init = False

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def _inflate(self, view, start_len, limit):
        """The double loop of zlib_Decompress_decompress_impl / decompress_buf (:622-716, :1102-1195): feed `view` in pieces,
        grow the output up to `limit` (None: no limit).  -> (output bytes, last return code, bytes of `view` not consumed)."""
        L, zst = _slib(), self._zst
        out = _OutBuf(zst, start_len, limit)
        L.zngamd_stream_inflate_ahead(_C.byref(zst), limit if limit is not None else 0xFFFFFFFFFFFFFFFF)     # what this call will take in all
        n = view.nbytes
        base, keep = _lib._addr(view) if n else (None, None)
        pos, err = 0, _lib.OK
        full = False
        while True:
            ln = min(n - pos, _CHUNK)
            zst.next_in = (base.value + pos) if ln else None
            zst.avail_in = ln
            while True:
                if not out.grow():
                    full = True
                    break
                err = L.zngamd_stream_inflate(_C.byref(zst), Z_SYNC_FLUSH)
                if err == _lib.NEED_DICT:
                    if self._zdict:
                        self._set_zdict()
                        continue                   # "repeat the call to inflate"
                    out.result()
                    raise _zerr(_lib.NEED_DICT, "while decompressing data")
                if err not in (_lib.OK, _lib.BUF_ERROR, _lib.STREAM_END):
                    out.result()
                    raise _stream_error(zst, err, "while decompressing data")
                if zst.avail_out != 0 or err == _lib.STREAM_END:
                    break
            pos += ln - zst.avail_in
            if full or err == _lib.STREAM_END or pos >= n or zst.avail_in:
                break
        return out.result(), err, n - pos


class _Decompress(_InflateStream):
    """zlib.decompressobj look-alike (zlib_ngmodule.c:622-1037): decompress(data, max_length), flush, copy,
    unused_data, unconsumed_tail, eof."""

    def __init__(self, wbits=MAX_WBITS, zdict=b""):
        self._lock = _threading.Lock()
        self.unused_data = b""
        self.unconsumed_tail = b""
        self.eof = False
        self._ended = False      # flush() after the end of the stream releases it (inflateEnd): no copy() afterwards
        self._open(wbits, zdict)

    def _save_unconsumed(self, view, left, err):
        # save_unconsumed_input, zlib_ngmodule.c:579-620
        tail = bytes(view[view.nbytes - left:]) if left else b""
        if err == _lib.STREAM_END:
            if tail:
                self.unused_data += tail
            self.unconsumed_tail = b""
        else:
            self.unconsumed_tail = tail

    def decompress(self, data, /, max_length=0):
        view = _view(data)
        max_length = _ssize(max_length)
        if max_length < 0:
            raise ValueError("max_length must be non-negative")
        with self._lock:
            if self.eof:                        # inflate() after the end: nothing is consumed, nothing comes out
                if view.nbytes:
                    self.unused_data += bytes(view)
                self.unconsumed_tail = b""
                return b""
            limit = max_length or None
            out, err, left = self._inflate(view, min(DEF_BUF_SIZE, max_length) if max_length else DEF_BUF_SIZE, limit)
            self._save_unconsumed(view, left, err)
            if err == _lib.STREAM_END:
                self.eof = True
            return out

    def flush(self, length=DEF_BUF_SIZE, /):
        length = _ssize(length)
        if length <= 0:
            raise ValueError("length must be greater than zero")
        with self._lock:
            view = _view(self.unconsumed_tail)
            if self.eof:                        # inflateEnd on a finished stream: no copy() afterwards
                self._ended = True
                self._close()
                return b""
            out, err, left = self._inflate(view, length, None)
            self._save_unconsumed(view, left, err)
            if err == _lib.STREAM_END:
                self.eof = True
                self._ended = True
                self._close()
            return out

    def copy(self):
        with self._lock:
            if self._ended or not self._init:
                raise ValueError("Inconsistent stream state")
            o = _Decompress.__new__(_Decompress)
            o._lock = _threading.Lock()
            o._zst = _ZStream()
            o._init = False
            o._zdict = self._zdict
            err = _slib().zngamd_stream_inflate_copy(_C.byref(o._zst), _C.byref(self._zst))
            if err == _lib.MEM_ERROR:
                raise MemoryError("Can't allocate memory for decompression object")
            if err != _lib.OK:
                raise ValueError("Inconsistent stream state")
            o._init = True
            o.unused_data, o.unconsumed_tail, o.eof = self.unused_data, self.unconsumed_tail, self.eof
            o._ended = False
            return o

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()


_Decompress.__module__ = "zlib_ng"


def decompressobj(wbits=MAX_WBITS, zdict=b""):
    return _Decompress(wbits, zdict)


class _ZlibDecompressor(_InflateStream):
    """bz2/lzma-style decompressor (zlib_ngmodule.c:1040-1438): decompress(data, max_length=-1), eof,
    unused_data, needs_input.  Input that is not needed yet stays here as compressed bytes (:1198-1308)."""

    def __init__(self, wbits=MAX_WBITS, zdict=b""):
        self._lock = _threading.Lock()
        self._pending = b""          # unconsumed input of earlier calls
        self.eof = False
        self.unused_data = b""
        self.needs_input = True
        self._open(wbits, zdict)

    def decompress(self, data, max_length=-1):
        max_length = _ssize(max_length)
        with self._lock:
            if self.eof:
                raise EOFError("End of stream already reached")
            data = bytes(_view(data))
            view = _view(self._pending + data if self._pending else data)
            if max_length < 0:
                limit, start = None, DEF_BUF_SIZE
            else:
                limit, start = max_length, min(max_length, 4 << 20)
            if limit == 0:
                self._pending = bytes(view)
                self.needs_input = False if view.nbytes else True
                return b""
            out, err, left = self._inflate(view, max(start, 1), limit)
            tail = bytes(view[view.nbytes - left:]) if left else b""
            if err == _lib.STREAM_END:
                self.eof = True
                self.needs_input = False
                self._pending = b""
                if tail:
                    self.unused_data = tail
                self._close()
            elif not tail:
                self._pending = b""
                self.needs_input = True
            else:
                self._pending = tail
                self.needs_input = False
            return out


_ZlibDecompressor.__module__ = "zlib_ng"


# ---- _GzipReader (zlib_ngmodule.c:2215-2930) ------------------------------------------------------------------
_GZ_ERR = {
    _lib.E_GZ_METHOD: lambda m: BadGzipFile("Unknown compression method"),
    _lib.E_GZ_CRC: lambda m: BadGzipFile(m if m.startswith("CRC check failed") else "CRC check failed"),
    _lib.E_GZ_LENGTH: lambda m: BadGzipFile("Incorrect length of data produced"),
    _lib.E_GZ_HCRC: lambda m: BadGzipFile("Corrupted gzip header. Checksums do not match"),
    _lib.E_GZ_TRUNC: lambda m: EOFError("Compressed file ended before the end-of-stream marker was reached"),
}


class _GzipReader:
    """Multi-member gzip reader with the surface of the reference's C type (zlib_ngmodule.c:2215-2930).
    The compressed input is read in windows (ZNGAMD_READ_WINDOW bytes, default 64 MiB; a bytes-like `fp` is one
    window).  Every window goes to the GPU engine, which decodes all members that are complete inside it -- two-pass
    for indexed members, one launch for BGZF members, chunk-parallel for ordinary ones -- and says where the first
    incomplete member starts; that tail is kept and the next window appended.  A member larger than the window is
    decoded block-wise across windows (the engine keeps the bit offset of the next block header, the last 32 KiB of
    output, CRC and length in a small state).  An error found after N good bytes is raised when the reader reaches byte N, as
    the streaming reference does."""

    def __init__(self, fp, /, buffersize=32 * 1024):
        if buffersize < 1:
            raise ValueError(f"buffersize must be at least 1, got {buffersize}")
        self._fp = fp
        self._is_file = hasattr(fp, "read")
        self._start = None
        if self._is_file:
            try:
                self._start = fp.tell()
            except (AttributeError, OSError, ValueError):
                self._start = None
        self._closed = False
        self._mtime = 0
        self._lock = _threading.Lock()
        self._size = -1
        self._decode_ahead = _os.environ.get("ZNGAMD_NO_DECODE_AHEAD") is None     # (gzip_ng_threaded's reader, which has a pump thread of its own, turns it off; ZNGAMD_NO_DECODE_AHEAD=1 does it for every reader)
        # Reading and decoding ahead touch `fp` from a thread of their own and ask for whole windows: only for sources that can be
        # sought (regular files, BytesIO) -- on a pipe or a socket a consumer that stops early would wait in close() for a window
        # that may never come, and a caller that goes on using its file object would meet a second reader.
        self._bulk_ok = False
        if self._is_file and self._start is not None:
            try:
                self._bulk_ok = bool(fp.seekable())
            except (AttributeError, OSError, ValueError):
                self._bulk_ok = False
        # (r06) a file written by this package's threaded writer carries, behind its data member, the segment index of that
        # member's blocks (empty members with a 'ZA' FEXTRA subfield, a locator last: _lib.parse_index_tail): a source that can
        # be sought is asked for it once, here, before anything reads ahead; the engine then decodes the units of every window
        # side by side (zngamd_gz_state.index).  Any doubt -- no locator, sizes that do not add up -- and the file is read as
        # any gzip file is.
        self._index = None
        if self._bulk_ok and _os.environ.get("ZNGAMD_NO_INDEX") is None:
            try:
                end = fp.seek(0, 2)
                fp.seek(self._start)
                got = _lib.parse_index_tail(fp, self._start, end) if end - self._start >= (1 << 16) else None
                if got is not None:
                    uin, uout, rows = got
                    h = _C.c_void_p()
                    ctx = _ctx()
                    if ctx.L.zngamd_index_create(ctx.h, len(uin), uin.ctypes.data_as(_C.c_void_p), uout.ctypes.data_as(_C.c_void_p),
                                                 rows.ctypes.data_as(_C.c_void_p), _C.byref(h)) == 0:
                        self._index = h
            except Exception:
                self._index = None
        self._reset()

    @property
    def _last_mtime(self):
        """MTIME of the stream's header once it has been read, None before that or when it is zero
        (GzipReader_get_last_mtime, zlib_ngmodule.c:2887-2893)."""
        return self._mtime or None

    def _reset(self):
        self._release_windows()  # FIRST: a thread that is decoding or reading ahead works on the state below; it is waited for here
        self._window = max(1 << 16, int(_os.environ.get("ZNGAMD_READ_WINDOW", 64 << 20)))
        self._buf = b""          # decoded, not yet handed out
        self._boff = 0
        self._pos = 0            # position in the decompressed stream
        self._carry = b""        # compressed bytes of a member that was not complete yet
        self._in_eof = False
        self._done = False       # nothing more to decode
        self._error = None       # raised once the buffered good bytes are gone
        self._first = True
        self._state = _lib.GzState()      # where a member larger than the window is being continued
        if getattr(self, "_index", None) is not None:
            self._state.index = self._index   # (the first member's; the engine forgets it when that member ends)
        self._out = None                  # decoded window (buffer, address); reused, so its pages are faulted in once
        self._spare = []                  # buffers given back by a consumer that took whole windows (_take_window)
        self._in_buf = None               # the compressed window: one buffer, refilled (the engine call is through by then)
        self._ahead = None                # the next window's file bytes being read on a thread of their own (_start_ahead)
        self._dec_ahead = None            # the next window being decoded on a thread of its own (_start_decode_ahead)
        self._windows_taken = 0           # decoded windows the consumer has taken: decoding ahead starts with the second
        self._calls = 0

    def _release_windows(self):
        """Window buffers go back to the process-wide pool (_lib.take_buffer): the next reader finds them warm."""
        self._buf = b""
        if getattr(self, "_dec_ahead", None) is not None:    # a decode-ahead still running: wait, its buffer goes back below
            t, box = self._dec_ahead
            self._dec_ahead = None
            t.join()
            res = box.get("res")
            if res is not None and res[1] is not None:
                _lib.give_buffer(res[1][0])
        if getattr(self, "_ahead", None) is not None:        # a read-ahead still running: wait, drop what it read
            try:
                ahead = self._join_ahead()
                if ahead is not None:
                    _lib.give_buffer(ahead[0])
            except BaseException:
                pass
        for pair in [getattr(self, "_out", None)] + list(getattr(self, "_spare", [])):
            if pair is not None:
                _lib.give_buffer(pair[0])
        _lib.give_buffer(getattr(self, "_in_buf", None))
        self._out, self._spare, self._in_buf = None, [], None

    # -- compressed input, one window at a time
    _AHEAD_FRONT = 4 << 20        # room kept in front of a window read ahead, for the unconsumed tail of the window before it

    def _file_read_into(self, buf):
        """Fill `buf` from the file -> (bytes delivered, end of file seen).  Touches nothing but the file."""
        got, eof = 0, False
        into = getattr(self._fp, "readinto", None)
        while got < len(buf):
            if into is not None:
                n = into(buf[got:])
                if not n:
                    eof = True
                    break
            else:
                chunk = self._fp.read(len(buf) - got)
                if not chunk:
                    eof = True
                    break
                n = len(chunk)
                buf[got:got + n] = chunk
            got += n
        return got, eof

    def _start_ahead(self):
        """Read the next window's bytes from the file on a thread of its own while the engine decodes this one (both release
        the interpreter lock): into a second buffer, behind room for the tail this window will leave over."""
        if not self._bulk_ok or self._in_eof or self._ahead is not None or self._closed or _sys.is_finalizing():
            return
        front, want = self._AHEAD_FRONT, self._window
        buf = _lib.take_buffer(front + want)
        box = {}

        def work():
            try:
                box["got"], box["eof"] = self._file_read_into(memoryview(buf)[front:front + want])
            except BaseException as exc:                 # raised by the reader's own thread when it takes the window over
                box["error"] = exc
        t = _threading.Thread(target=work, name="zng-amd-read-ahead")
        t.start()
        self._ahead = (t, buf, box, front)

    def _join_ahead(self):
        """-> (buffer, front room, bytes read, end of file) of the window read ahead, None when there is none."""
        if self._ahead is None:
            return None
        t, buf, box, front = self._ahead
        self._ahead = None
        t.join()
        if "error" in box:
            _lib.give_buffer(buf)
            raise box["error"]
        return buf, front, box["got"], box["eof"]

    def _read_window(self, carry):
        """The unconsumed tail of the previous window followed by up to one window of new input, in one buffer."""
        if not self._is_file:
            self._in_eof = True
            return bytes(carry) + bytes(_view(self._fp)) if self._first else carry
        keep = len(carry)
        ahead = self._join_ahead()
        if ahead is not None:
            buf, front, got, eof = ahead
            old = self._in_buf
            if keep <= front:                                    # the tail goes in front of what was read ahead: no copy of the window
                view = memoryview(buf)
                if keep:
                    view[front - keep:front] = carry
                self._in_buf = buf
                if old is not None:
                    _lib.give_buffer(old)                        # (the carry was copied out of it above)
                self._in_eof = eof
                return view[front - keep:front + got]
            joined = _lib.take_buffer(keep + got)                # a tail longer than the room (a member without block boundaries for megabytes)
            view = memoryview(joined)
            view[:keep] = carry
            view[keep:keep + got] = memoryview(buf)[front:front + got]
            _lib.give_buffer(buf)
            self._in_buf = joined
            if old is not None:
                _lib.give_buffer(old)
            self._in_eof = eof
            return view[:keep + got]
        need = keep + self._window
        old = self._in_buf
        if old is None or len(old) < need:
            self._in_buf = _lib.take_buffer(need)            # (pooled, not zero-filled by us: only what the file delivers is used)
        buf = memoryview(self._in_buf)[:need]
        if keep:
            buf[:keep] = carry                               # (the carry may lie in the old buffer, or at the front of this one)
        if old is not None and old is not self._in_buf:
            _lib.give_buffer(old)
        got, eof = self._file_read_into(buf[keep:])
        if eof:
            self._in_eof = True
        return buf[:keep + got]

    def _error_for(self, code, data, ctx):
        msg = ctx.err()
        if code == _lib.E_GZ_MAGIC:
            # locate the offending bytes the way the reference reports them
            return _magic_error(data, ctx) or BadGzipFile("Not a gzipped file (b'??')")
        if code in _GZ_ERR:
            return _GZ_ERR[code](msg)
        return _zerr(code if code in _MSG else _lib.DATA_ERROR, "while decompressing data")

    def _decode_window(self):
        """The next decoded window -> (bytes view, its buffer, end of stream, pending error).  Works on the decoding side of the
        state only (input windows, carry, the engine's stream state) and on buffers nobody reads, never on _buf / _boff / _out of
        the window the consumer is reading: it may run on a thread beside the consumer (_start_decode_ahead)."""
        ctx = _ctx()
        pair = None
        while True:
            data = self._carry if self._in_eof else self._read_window(self._carry)
            self._carry = b""
            if self._first:
                self._first = False
                if len(data) >= 8:
                    self._mtime = _struct.unpack_from("<I", data, 4)[0]
                if len(data) >= 2 and data[:2] != b"\x1f\x8b":
                    return b"", pair, True, BadGzipFile(f"Not a gzipped file ({bytes(data[:2])!r})")
            if not data:
                return b"", pair, True, None
            final = self._in_eof
            isize = _struct.unpack_from("<I", data, len(data) - 4)[0] if (final and len(data) >= 18) else 0
            cap = max(1 << 16, 4 * len(data), min(isize, 1032 * len(data)) + 64)      # (ISIZE is untrusted: bounded by what deflate can expand)
            while True:
                if pair is None and self._spare:
                    try:
                        pair = self._spare.pop()         # a window buffer its consumer has finished with
                    except IndexError:
                        pair = None
                if pair is None or len(pair[0]) < cap:
                    if pair is not None:
                        _lib.give_buffer(pair[0])
                    pair = _lib.take_window(cap + cap // 4)       # head-room: windows differ a little in size
                cap = len(pair[0])
                self._start_ahead()                      # the file read of the next window runs beside the engine call
                code, out, nm, used = ctx.gunzip_stream(self._state, data, cap, final, into=pair)
                if code == _lib.BUF_ERROR and (len(out) >= cap or ctx.last_needed > cap):
                    cap = max(cap * 4, ctx.last_needed + 64)
                    continue
                break
            if code != _lib.OK:
                return out, pair, True, self._error_for(code, data, ctx)
            if final:
                done = True
                if 0 < used < len(data):
                    self._carry = bytes(memoryview(data)[used:])   # a continued member ended inside the last window: the rest follows
                    done = False
                if len(out) or done:
                    return out, pair, done, None
                continue
            if used == 0:
                # neither a complete member nor a complete deflate block in the window: take a larger one
                self._carry = data
                self._window *= 2
                continue
            self._carry = bytes(memoryview(data)[used:])
            if len(out):
                return out, pair, False, None

    def _start_decode_ahead(self):
        """The window after this one is decoded on a thread of its own while the consumer copies this one out (the engine call,
        the file read and the copy all release the interpreter lock): gzip_ng.open's reader, which has no pump thread of its
        own, alternated 13 ms of decoding with 15 ms of copying per window."""
        # (only once the consumer has come back for a second window: a reader that wants the first bytes of a file does not
        # pay for the decode of a window it never asks for)
        if (not self._decode_ahead or not self._bulk_ok or self._windows_taken < 2 or self._dec_ahead is not None or self._closed
                or self._done or self._error is not None or _sys.is_finalizing()):
            return
        box = {}

        def work():
            try:
                box["res"] = self._decode_window()
            except BaseException as exc:                 # raised where the window would have been taken over
                box["exc"] = exc
        t = _threading.Thread(target=work, name="zng-amd-decode-ahead")
        t.start()
        self._dec_ahead = (t, box)

    def _next_decoded(self):
        if self._dec_ahead is not None:
            t, box = self._dec_ahead
            self._dec_ahead = None
            t.join()
            if "exc" in box:
                raise box["exc"]
            return box["res"]
        return self._decode_window()

    def _fill(self):
        """Decode until some output is buffered, the stream ends, or an error is pending."""
        while not self._done and self._boff >= len(self._buf):
            buf, pair, done, err = self._next_decoded()
            self._buf = b""                              # (drop the view of the window that is finished before its buffer goes back)
            if self._out is not None and self._out is not pair:
                self._give_back(self._out)
            self._out = pair
            self._buf, self._boff, self._done = buf, 0, done
            if err is not None:
                self._error = err
                self._done = True
            self._windows_taken += 1
            self._start_decode_ahead()

    def _check(self):
        if self._closed:
            raise ValueError("I/O operation on closed file.")

    def _avail(self):
        if self._boff >= len(self._buf) and not self._done:
            self._fill()
        return len(self._buf) - self._boff

    def readinto(self, b, /):
        self._check()
        with self._lock:
            mv = _view(b)
            n = min(mv.nbytes, self._avail())
            if n <= 0:
                if self._error is not None:
                    raise self._error
                self._size = self._pos
                return 0
            self._calls += 1
            if (n >= 32768 and (self._calls & 7) == 0 and self._dec_ahead is not None and self._out is not None and not mv.readonly
                    and len(self._buf) <= len(self._out[0])):
                # while the next window is being decoded on its thread, every eighth large copy lets go of the interpreter lock
                # (a foreign call): that thread needs it between its file read and its engine call
                anchor = _C.c_char.from_buffer(mv)
                _C.memmove(_C.addressof(anchor), self._out[1].value + self._boff, n)
                del anchor
            else:
                mv[:n] = memoryview(self._buf)[self._boff:self._boff + n]      # one copy, no intermediate bytes object
            self._boff += n
            self._pos += n
            return n

    def read(self, size=-1, /):
        self._check()
        if size is None or size < 0:
            return self.readall()
        with self._lock:
            n = min(int(size), self._avail())
            if n <= 0:
                if self._error is not None and size > 0:
                    raise self._error
                if size > 0:
                    self._size = self._pos
                return b""
            piece = bytes(memoryview(self._buf)[self._boff:self._boff + n])      # the one copy
            self._boff += n
            self._pos += n
            return piece

    def _take_window(self):
        """(view, token): the unread rest of the decoded window as a memoryview whose buffer now belongs to the caller (the next
        window is decoded into another one; hand the token to _give_back when done), None at the end of the stream; a pending
        error is raised once the good bytes are gone.
        For a consumer that overlaps its own work with the decode of the next window (gzip_ng_threaded's reader)."""
        self._check()
        with self._lock:
            if self._avail() <= 0:
                if self._error is not None:
                    raise self._error
                self._size = self._pos
                return None
            view = memoryview(self._buf)[self._boff:]
            self._pos += len(view)
            self._boff = len(self._buf)
            token, self._out = self._out, None
            return view, token

    def _give_back(self, token):
        """A window buffer handed out by _take_window is free again (its pages are faulted in and known to the driver:
        decoding into it again is much cheaper than into a fresh one)."""
        if token is not None:
            if len(self._spare) < 3 and not self._closed:
                self._spare.append(token)
            else:
                _lib.give_buffer(token[0])

    def readall(self):
        self._check()
        with self._lock:
            parts = []
            while self._avail() > 0:
                parts.append(bytes(self._buf[self._boff:]))
                self._pos += len(self._buf) - self._boff
                self._boff = len(self._buf)
            if self._error is not None:
                raise self._error
            self._size = self._pos
            return parts[0] if len(parts) == 1 else b"".join(parts)

    def _skip_to(self, target):
        while self._pos < target and self._avail() > 0:
            n = min(target - self._pos, len(self._buf) - self._boff)
            self._boff += n
            self._pos += n

    def seek(self, offset, whence=0, /):
        self._check()
        with self._lock:
            if whence == 0:
                target = offset
            elif whence == 1:
                target = self._pos + offset
            elif whence == 2:
                # the size is only known at the end of the stream
                self._skip_to(1 << 62)
                if self._error is not None:
                    raise self._error
                self._size = self._pos
                target = self._size + offset
            else:
                raise ValueError(f"Invalid format for whence: {whence}")
            target = max(0, target)
            if target < self._pos:
                # backwards: decode again from the start (what the reference's reader does as well)
                back = self._pos - target
                if back <= self._boff:
                    self._boff -= back
                    self._pos = target
                    return self._pos
                if self._is_file and self._start is None:
                    raise _io.UnsupportedOperation("underlying stream is not seekable")
                size = self._size
                self._reset()                                # (waits for a read-ahead that is still using the file)
                self._size = size
                if self._is_file:
                    self._fp.seek(self._start)
            self._skip_to(target)
            return self._pos

    def tell(self):
        self._check()
        return self._pos

    def close(self):
        self._closed = True
        with self._lock:
            self._release_windows()
            ix, self._index = getattr(self, "_index", None), None
            if ix is not None:                   # (no window is being decoded any more: nothing names the handle)
                try:
                    self._state.index = None
                    _ctx().L.zngamd_index_destroy(ix)
                except Exception:
                    pass

    def readable(self):
        return True

    def writable(self):
        return False

    def seekable(self):
        return True

    @property
    def closed(self):
        return self._closed

    def fileno(self):
        raise _io.UnsupportedOperation("fileno")

    def isatty(self):
        return False

    def flush(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        pass


def _magic_error(raw, ctx):
    """Walk the good members with the engine to find the two bytes that are not a gzip magic."""
    raw = bytes(raw)
    pos, n = 0, len(raw)
    try:
        while pos < n:
            if raw[pos:pos + 2] != b"\x1f\x8b":
                return BadGzipFile(f"Not a gzipped file ({raw[pos:pos + 2]!r})")
            start = _parse_gzip_header(raw, pos)
            code, out, used, _, _ = _inflate_all(raw[start:])
            if code != _lib.STREAM_END:
                return None
            pos = start + used + 8
            while pos < n and raw[pos] == 0:
                pos += 1
    except Exception:
        return None
    return None


_GzipReader.__module__ = "zlib_ng"
