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
    raw, crc, adler = _ctx().deflate_stream(mv, level, wb)
    if kind == "raw":
        return raw
    if kind == "zlib":
        return _zlib_header(level, wb) + raw + _struct.pack(">I", adler)
    return _gzip_header(level) + raw + _struct.pack("<II", crc, mv.nbytes & 0xFFFFFFFF)


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


# ---- compressobj: buffering writer ------------------------------------------------------------------------
class _Compress:
    """Incremental compressor with zlib.compressobj's surface (compress / flush).  Input is collected
    and handed to the engine in large dictionary-chained batches; every batch ends on a sync-flush
    boundary, so the concatenation is one valid deflate stream."""

    _BATCH = 8 << 20

    def __init__(self, level, method, wbits, memLevel, strategy, zdict):
        try:
            _check_level(level)
        except error:
            raise ValueError("Invalid initialization option") from None
        if method != DEFLATED or not (1 <= memLevel <= 9) or strategy not in (0, 1, 2, 3, 4):
            raise ValueError("Invalid initialization option")
        try:
            self._kind, self._wb = _container(wbits)
        except error:
            raise ValueError("Invalid initialization option") from None
        self._level = level
        self._pending = bytearray()
        self._tail = bytes(_view(zdict))[-32768:] if zdict is not None else b""
        self._started = False
        self._finished = False
        self._crc, self._adler, self._size = 0, 1, 0
        self._lock = _threading.Lock()
        # zlib container with a preset dictionary: FDICT + DICTID in the header (deflateSetDictionary, zlib_ngmodule.c:401)
        self._dictid = adler32(_view(zdict)) if (zdict is not None and self._kind == "zlib") else None

    def _emit(self, final):
        out = []
        if not self._started:
            self._started = True
            if self._kind == "zlib":
                out.append(_zlib_header(self._level, self._wb, self._dictid))
            elif self._kind == "gzip":
                out.append(_gzip_header(self._level))
        data = bytes(self._pending)
        self._pending.clear()
        if data or final:
            ctx = _ctx()
            buf = self._tail + data
            flags = (_lib.FLAG_FINAL if final else 0) | ((self._wb & 15) << 8)      # ZNGAMD_FLAG_WBITS: stay inside the declared window
            outs, crcs, _ = ctx.deflate_blocks(buf, [(len(self._tail), len(data), len(self._tail), flags)],
                                               self._level, len(data) + len(data) // 8 + (len(data) // 131072 + 2) * 64)
            out.append(outs[0])
            self._crc = crc32_combine(self._crc, crcs[0], len(data))
            if self._kind == "zlib" and data:
                self._adler = ctx.adler32(data, self._adler)
            self._size += len(data)
            self._tail = buf[-32768:]
        if final:
            self._finished = True
            if self._kind == "zlib":
                out.append(_struct.pack(">I", self._adler))
            elif self._kind == "gzip":
                out.append(_struct.pack("<II", self._crc, self._size & 0xFFFFFFFF))
        return b"".join(out)

    def _emit_direct(self, view):
        """A large piece with nothing pending: its first 128 KiB go through a small buffer behind the dictionary tail, the
        rest is compressed where it lies in the caller's buffer (each unit primed by the bytes in front of it)."""
        out = []
        if not self._started:
            out.append(self._emit(False))                   # header only (nothing pending)
        ctx = _ctx()
        n = view.nbytes
        first = min(131072, n)
        wflag = (self._wb & 15) << 8
        head = self._tail + bytes(view[:first])
        outs, crcs, _ = ctx.deflate_blocks(head, [(len(self._tail), first, len(self._tail), wflag)], self._level, first + first // 8 + 256)
        out.append(outs[0])
        self._crc = crc32_combine(self._crc, crcs[0], first)
        # the engine's block length is a u32: the rest goes in pieces of at most 1 GiB (whole units), each primed by the 32 KiB
        # in front of it in the caller's buffer (the reference loops in UINT32_MAX pieces, zlib_ngmodule.c:142-147)
        pos = first
        while pos < n:
            ln = min(n - pos, 1 << 30)
            packed, crcs, _, _ = ctx.deflate_blocks(view, [(pos, ln, 32768, wflag)], self._level,
                                                    ln + ln // 8 + (ln // 131072 + 2) * 64, joined=True)
            out.append(packed)
            self._crc = crc32_combine(self._crc, crcs[0], ln)
            pos += ln
        if self._kind == "zlib":
            self._adler = ctx.adler32(view, self._adler)
        self._size += n
        self._tail = bytes(view[-32768:]) if n >= 32768 else (self._tail + bytes(view))[-32768:]
        return b"".join(out)

    def compress(self, data, /):
        with self._lock:
            if self._finished:
                raise _zerr(_lib.STREAM_ERROR, "while compressing data")
            view = _view(data)
            if not self._pending and view.nbytes >= self._BATCH and view.contiguous:
                return self._emit_direct(view)
            self._pending += view
            if len(self._pending) >= self._BATCH:
                return self._emit(False)
            return b""

    def flush(self, mode=Z_FINISH, /):
        mode = _operator.index(mode)
        with self._lock:
            if mode == Z_NO_FLUSH:
                return b""
            if self._finished or not (0 <= mode <= Z_BLOCK):       # deflate() answers Z_STREAM_ERROR to both
                raise _zerr(_lib.STREAM_ERROR, "while flushing")
            return self._emit(mode == Z_FINISH)

    def copy(self):
        """zlib_Compress_copy_impl (zlib_ngmodule.c:790-850): an independent compressor in the same state.  All stream
        state lives here (pending input, 32 KiB dictionary tail, running checksums), so the copy is exact."""
        with self._lock:
            if self._finished:
                raise ValueError("Inconsistent stream state")
            o = _Compress.__new__(_Compress)
            o._kind, o._wb, o._level, o._dictid = self._kind, self._wb, self._level, self._dictid
            o._pending = bytearray(self._pending)
            o._tail = self._tail
            o._started, o._finished = self._started, self._finished
            o._crc, o._adler, o._size = self._crc, self._adler, self._size
            o._lock = _threading.Lock()
            return o

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()


_Compress.__module__ = "zlib_ng"


def compressobj(level=Z_DEFAULT_COMPRESSION, method=DEFLATED, wbits=MAX_WBITS, memLevel=DEF_MEM_LEVEL,
                strategy=Z_DEFAULT_STRATEGY, zdict=None):
    return _Compress(level, method, wbits, memLevel, strategy, zdict)


# ---- incremental inflate: decompressobj / _ZlibDecompressor (zlib_ngmodule.c:456-502, :622-716, :1040-1438) ----------
class _InflateCore:
    """Block-resumable inflate on the GPU engine.  Input is kept from the last deflate-block header onward;
    every call decodes from there (start bit + up to 32 KiB of history) with the sequential wavefront decoder
    (`zngamd_inflate_resume`) and hands out only what is new.  Container header / trailer bytes are handled
    here; payload checksums are computed by the engine."""

    def __init__(self, wbits, zdict):
        if wbits == 0 or 8 <= wbits <= 15:
            self.kind = "zlib"
        elif -15 <= wbits <= -8:
            self.kind = "raw"
        elif 24 <= wbits <= 31 or wbits == 16:
            self.kind = "gzip"
        elif 40 <= wbits <= 47 or wbits == 32:
            self.kind = "auto"
        else:
            raise ValueError("Invalid initialization option")
        self.wbits = wbits
        self.zdict = bytes(_view(zdict)) if zdict is not None else b""
        self.window = self.zdict[-32768:] if self.kind == "raw" else b""
        self.buf = bytearray()
        self.start_bit = 0
        self.skip = 0
        self.header_done = self.kind == "raw"
        self.deflate_done = False
        self.eof = False
        self.unused = b""
        self.check = 1                    # running Adler-32 (zlib) or CRC-32 (gzip)
        self.total = 0

    def clone(self):
        import copy as _copy
        o = _copy.copy(self)
        o.buf = bytearray(self.buf)
        return o

    def _header(self):
        W = "while decompressing data"
        b = self.buf
        if self.kind == "auto" and len(b) >= 2:
            self.kind = "gzip" if b[:2] == b"\x1f\x8b" else "zlib"
            self.wbits -= 32                 # 0: take the window from the zlib header
        if self.kind == "zlib":
            if len(b) < 2:
                return False
            cmf, flg = b[0], b[1]
            if (cmf & 15) != 8 or ((cmf << 8) | flg) % 31:
                raise _zerr(_lib.DATA_ERROR, W, "incorrect header check")
            if (cmf >> 4) + 8 > (self.wbits if self.wbits else 15):
                raise _zerr(_lib.DATA_ERROR, W, "invalid window size")
            need = 2
            if flg & 0x20:
                if len(b) < 6:
                    return False
                if not self.zdict:
                    raise _zerr(_lib.NEED_DICT, W)
                if _struct.unpack(">I", bytes(b[2:6]))[0] != adler32(self.zdict):
                    raise _zerr(_lib.DATA_ERROR, "while setting zdict")
                self.window = self.zdict[-32768:]
                need = 6
            del b[:need]
            self.check = 1
        elif self.kind == "gzip":
            if len(b) >= 4 and b[:2] == b"\x1f\x8b" and b[2] == 8 and b[3] & 0xE0:
                raise _zerr(_lib.DATA_ERROR, W, "unknown header flags set")
            try:
                start = _parse_gzip_header(bytes(b))
            except EOFError:
                return False
            except BadGzipFile as e:
                msg = str(e)
                raise _zerr(_lib.DATA_ERROR, W, "unknown compression method" if msg.startswith("Unknown compression") else
                            "header crc mismatch" if msg.startswith("Corrupted gzip header") else "incorrect header check") from None
            del b[:start]
            self.check = 0
        else:
            return False
        self.header_done = True
        return True

    def _trailer(self):
        need = {"zlib": 4, "gzip": 8, "raw": 0}[self.kind]
        W = "while decompressing data"
        if self.kind == "gzip" and 4 <= len(self.buf) < 8:
            # inflate() compares the CRC as soon as its four bytes are there, before ISIZE has arrived
            if _struct.unpack("<I", bytes(self.buf[:4]))[0] != self.check:
                raise _zerr(_lib.DATA_ERROR, W, "incorrect data check")
        if len(self.buf) < need:
            return
        t = bytes(self.buf[:need])
        if self.kind == "zlib" and _struct.unpack(">I", t)[0] != self.check:
            raise _zerr(_lib.DATA_ERROR, W, "incorrect data check")
        if self.kind == "gzip":
            crc, isize = _struct.unpack("<II", t)
            if crc != self.check:
                raise _zerr(_lib.DATA_ERROR, W, "incorrect data check")
            if isize != self.total & 0xFFFFFFFF:
                raise _zerr(_lib.DATA_ERROR, W, "incorrect length check")
        self.unused = bytes(self.buf[need:])
        self.buf.clear()
        self.eof = True

    def feed(self, data, limit=None):
        """Append `data`, decode, return the new output (at most `limit` bytes when given) and the number of
        input bytes of this call that zlib would report as still unconsumed."""
        if self.eof:
            self.unused += bytes(data)
            return b"", 0
        self.buf += data
        if not self.header_done and not self._header():
            return b"", 0
        if self.deflate_done:
            self._trailer()
            return b"", 0
        ctx = _ctx()
        want = None if limit is None else self.skip + limit
        cap = max(1 << 16, 8 * len(self.buf) + self.skip)
        if want is not None and want < cap:
            cap = want
        while True:
            code, out, in_bits, bb, bo = ctx.inflate_resume(bytes(self.buf), self.start_bit, self.window, max(cap, 1))
            if code == _lib.E_OVERFLOW and (want is None or cap < want):      # our guess was short, not the caller's limit
                cap = cap * 4 if want is None else min(cap * 4, want)
                continue
            break
        new = out[self.skip:]
        if code == _lib.DATA_ERROR:
            raise _zerr(_lib.DATA_ERROR, "while decompressing data")
        if new:
            self.check = crc32(new, self.check) if self.kind == "gzip" else adler32(new, self.check) if self.kind == "zlib" else 0
            self.total += len(new)
        left = 0
        if code == _lib.STREAM_END:
            del self.buf[:(in_bits + 7) // 8]
            self.deflate_done = True
            self.skip = 0
            self._trailer()
        elif code == _lib.E_OVERFLOW:
            # output limit reached in the middle of a block: report the input beyond the stop position as unconsumed (it
            # stays buffered here as well) and move the resume point to the header of the block the decoder stopped in,
            # so that the next call re-decodes at most that one block's delivered part instead of the whole buffer
            left = max(0, len(self.buf) - (in_bits + 7) // 8)
            self.window = (self.window + out[:bo])[-32768:]
            del self.buf[:bb // 8]
            self.start_bit = bb & 7
            self.skip = len(out) - bo
        else:                                   # input ran out: move the resume point to the last block header
            self.window = (self.window + out[:bo])[-32768:]
            del self.buf[:bb // 8]
            self.start_bit = bb & 7
            self.skip = len(out) - bo
        return new, left


class _Decompress:
    """zlib.decompressobj look-alike (zlib_ngmodule.c:622-1037): decompress(data, max_length), flush, copy,
    unused_data, unconsumed_tail, eof."""

    def __init__(self, wbits=MAX_WBITS, zdict=b""):
        self._core = _InflateCore(wbits, zdict)
        self._lock = _threading.Lock()
        self.unused_data = b""
        self.unconsumed_tail = b""
        self.eof = False
        self._ahead = 0          # bytes at the end of the core buffer that were handed back as unconsumed_tail
        self._ended = False      # flush() after the end of the stream releases it (inflateEnd): no copy() afterwards

    def _sync(self):
        self.eof = self._core.eof
        self.unused_data = self._core.unused

    def decompress(self, data, /, max_length=0):
        data = bytes(_view(data))
        max_length = _ssize(max_length)
        if max_length < 0:
            raise ValueError("max_length must be non-negative")
        with self._lock:
            if self._ahead:                     # the caller feeds unconsumed_tail back: already buffered
                data = data[min(self._ahead, len(data)):]
                self._ahead = 0
            out, left = self._core.feed(data, max_length or None)
            self.unconsumed_tail = bytes(self._core.buf[len(self._core.buf) - left:]) if left else b""
            self._ahead = left
            self._sync()
            return out

    def flush(self, length=DEF_BUF_SIZE, /):
        if _ssize(length) <= 0:
            raise ValueError("length must be greater than zero")
        with self._lock:
            self._ahead = 0
            out, _ = self._core.feed(b"", None)
            self.unconsumed_tail = b""
            self._sync()
            self._ended = self._ended or self.eof
            return out

    def copy(self):
        with self._lock:
            if self._ended:
                raise ValueError("Inconsistent stream state")
            o = _Decompress.__new__(_Decompress)
            o._core = self._core.clone()
            o._lock = _threading.Lock()
            o.unused_data, o.unconsumed_tail, o.eof, o._ahead = self.unused_data, self.unconsumed_tail, self.eof, self._ahead
            o._ended = False
            return o

    __copy__ = copy

    def __deepcopy__(self, memo):
        return self.copy()


_Decompress.__module__ = "zlib_ng"


def decompressobj(wbits=MAX_WBITS, zdict=b""):
    return _Decompress(wbits, zdict)


class _ZlibDecompressor:
    """bz2/lzma-style decompressor (zlib_ngmodule.c:1040-1438): decompress(data, max_length=-1), eof,
    unused_data, needs_input."""

    def __init__(self, wbits=MAX_WBITS, zdict=b""):
        self._core = _InflateCore(wbits, zdict)
        self._lock = _threading.Lock()
        self._more = False       # the last call stopped at max_length: output may be waiting behind buffered input
        self.needs_input = True

    @property
    def eof(self):
        return self._core.eof

    @property
    def unused_data(self):
        return self._core.unused

    def decompress(self, data, max_length=-1):
        """max_length bounds what is DECODED, not only what is returned: input that is not needed yet stays buffered as
        compressed bytes (zlib_ngmodule.c:1198-1308), so a small input cannot make the object hold a large output."""
        max_length = _ssize(max_length)
        with self._lock:
            if self._core.eof:
                raise EOFError("End of stream already reached")
            data = bytes(_view(data))
            if max_length == 0:
                self._core.buf += data
                self._more = True
                self.needs_input = False
                return b""
            out, _ = self._core.feed(data, max_length if max_length > 0 else None)
            self._more = max_length > 0 and len(out) >= max_length and not self._core.eof
            self.needs_input = not self._more and not self._core.eof
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
        self._reset()

    @property
    def _last_mtime(self):
        """MTIME of the stream's header once it has been read, None before that or when it is zero
        (GzipReader_get_last_mtime, zlib_ngmodule.c:2887-2893)."""
        return self._mtime or None

    def _reset(self):
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
        self._out = None                  # decoded window (object, address); reused, so its pages are faulted in once
        self._spare = []                  # buffers given back by a consumer that took whole windows (_take_window)

    # -- compressed input, one window at a time
    def _read_window(self, carry):
        """The unconsumed tail of the previous window followed by up to one window of new input, in one buffer."""
        if not self._is_file:
            self._in_eof = True
            return carry + bytes(_view(self._fp)) if self._first else carry
        keep = len(carry)
        obj, buf = _lib.new_fillable(keep + self._window)      # not zero-filled: only what the file delivers is ever touched
        buf[:keep] = carry
        got = keep
        into = getattr(self._fp, "readinto", None)
        while got < len(buf):
            if into is not None:
                n = into(buf[got:])
                if not n:
                    self._in_eof = True
                    break
            else:
                chunk = self._fp.read(len(buf) - got)
                if not chunk:
                    self._in_eof = True
                    break
                n = len(chunk)
                buf[got:got + n] = chunk
            got += n
        buf.release()
        return memoryview(obj)[:got]

    def _set_error(self, code, data, ctx):
        msg = ctx.err()
        if code == _lib.E_GZ_MAGIC:
            # locate the offending bytes the way the reference reports them
            self._error = _magic_error(data, ctx) or BadGzipFile("Not a gzipped file (b'??')")
        elif code in _GZ_ERR:
            self._error = _GZ_ERR[code](msg)
        else:
            self._error = _zerr(code if code in _MSG else _lib.DATA_ERROR, "while decompressing data")

    def _fill(self):
        """Decode until some output is buffered, the stream ends, or an error is pending."""
        ctx = _ctx()
        while not self._done and self._boff >= len(self._buf):
            data = self._carry if self._in_eof else self._read_window(self._carry)
            self._carry = b""
            if self._first:
                self._first = False
                if len(data) >= 8:
                    self._mtime = _struct.unpack_from("<I", data, 4)[0]
                if len(data) >= 2 and data[:2] != b"\x1f\x8b":
                    self._buf, self._boff, self._done = b"", 0, True
                    self._error = BadGzipFile(f"Not a gzipped file ({bytes(data[:2])!r})")
                    return
            if not data:
                self._done = True
                return
            final = self._in_eof
            isize = _struct.unpack_from("<I", data, len(data) - 4)[0] if (final and len(data) >= 18) else 0
            cap = max(1 << 16, 4 * len(data), isize + 64)
            while True:
                if self._out is None and self._spare:
                    self._out = self._spare.pop()        # a window buffer the threaded reader's consumer has finished with
                if self._out is None or len(self._out[0]) < cap:
                    self._buf = b""                      # drop the view of the old buffer before replacing it
                    self._out = _lib.new_buffer(cap + cap // 4)   # head-room: windows differ a little in size
                cap = len(self._out[0])
                code, out, nm, used = ctx.gunzip_stream(self._state, data, cap, final, into=self._out)
                if code == _lib.BUF_ERROR and (len(out) >= cap or ctx.last_needed > cap):
                    cap = max(cap * 4, ctx.last_needed + 64)
                    continue
                break
            if code != _lib.OK:
                self._buf, self._boff, self._done = out, 0, True
                self._set_error(code, data, ctx)
                return
            if final:
                self._buf, self._boff = out, 0
                if 0 < used < len(data):
                    self._carry = bytes(memoryview(data)[used:])   # a continued member ended inside the last window: the rest follows
                else:
                    self._done = True
                if self._boff < len(self._buf) or self._done:
                    return
                continue
            if used == 0:
                # neither a complete member nor a complete deflate block in the window: take a larger one
                self._carry = data
                self._window *= 2
                continue
            self._buf, self._boff = out, 0
            self._carry = bytes(memoryview(data)[used:])

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
        if token is not None and len(self._spare) < 3:
            self._spare.append(token)

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
                if self._is_file:
                    if self._start is None:
                        raise _io.UnsupportedOperation("underlying stream is not seekable")
                    self._fp.seek(self._start)
                size = self._size
                self._reset()
                self._size = size
            self._skip_to(target)
            return self._pos

    def tell(self):
        self._check()
        return self._pos

    def close(self):
        self._closed = True

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
