"""gzip_ng -- drop-in face of the reference's `zlib_ng.gzip_ng` (reference src/zlib_ng/gzip_ng.py:33-205)
on the MI355X engine: `open`, `GzipNGFile` / `GzipFile`, one-shot `compress` / `decompress`,
`BadGzipFile`, `READ_BUFFER_SIZE`.

The single-stream file object keeps CPython's gzip.GzipFile for header / trailer / buffering logic and
swaps in this package's compressor object and C-ABI backed reader, as the reference does
(gzip_ng.py:140-149).  Compression inside one GzipNGFile is one LZ77 window over the whole file (sequential
by construction); the data-parallel writer is gzip_ng_threaded.
"""
import ctypes
import gzip
import io
import os
import struct
import sys
import threading
import time

from . import _lib, zlib_ng
from .zlib_ng import _GzipReader

__all__ = ["GzipFile", "open", "compress", "decompress", "BadGzipFile", "READ_BUFFER_SIZE"]

_COMPRESS_LEVEL_FAST = zlib_ng.Z_BEST_SPEED
_COMPRESS_LEVEL_TRADEOFF = zlib_ng.Z_DEFAULT_COMPRESSION
_COMPRESS_LEVEL_BEST = zlib_ng.Z_BEST_COMPRESSION

#: bytes requested from the underlying file per refill when decompressing (gzip_ng.py:42)
READ_BUFFER_SIZE = 512 * 1024

FTEXT, FHCRC, FEXTRA, FNAME, FCOMMENT = 1, 2, 4, 8, 16
_SMALL_WRITE = 32 * 1024           # writes shorter than this are collected ...
_SMALL_BATCH = 1 << 20             # ... and go to the compressor in batches of this size
READ, WRITE = gzip.READ, gzip.WRITE
BadGzipFile = gzip.BadGzipFile


def _is_pathlike(obj):
    return isinstance(obj, (str, bytes)) or hasattr(obj, "__fspath__")


def open(filename, mode="rb", compresslevel=_COMPRESS_LEVEL_TRADEOFF, encoding=None, errors=None, newline=None):
    """gzip.open look-alike (gzip_ng.py:51-95): binary modes give a GzipNGFile, text modes wrap it in a
    TextIOWrapper."""
    text = "t" in mode
    if text and "b" in mode:
        raise ValueError("Invalid mode: %r" % (mode,))
    if not text:
        for name, val in (("encoding", encoding), ("errors", errors), ("newline", newline)):
            if val is not None:
                raise ValueError(f"Argument '{name}' not supported in binary mode")
    raw_mode = mode.replace("t", "")
    if ("w" in raw_mode or "a" in raw_mode or "x" in raw_mode) and os.environ.get("ZNGAMD_WRITER_MEMBERS", "0") not in ("", "0"):
        # an addition (off unless the environment asks for it): the file as independent indexed members -- written by the block
        # writer of gzip_ng_threaded, which has that format; still a gzip file for every reader, and the members this engine's
        # own reader decodes with one wavefront each
        from . import gzip_ng_threaded
        wmode = raw_mode if "b" in raw_mode else raw_mode + "b"
        fobj = gzip_ng_threaded.FlushableBufferedWriter(
            gzip_ng_threaded._ThreadedGzipWriter(filename, wmode, level=compresslevel, threads=1, block_size=128 * 1024, indexed_members=True),
            buffer_size=(1 << 16) - 1)
        return io.TextIOWrapper(fobj, encoding, errors, newline) if text else fobj
    if _is_pathlike(filename):
        fobj = GzipNGFile(filename, raw_mode, compresslevel)
    elif hasattr(filename, "read") or hasattr(filename, "write"):
        fobj = GzipNGFile(None, raw_mode, compresslevel, filename)
    else:
        raise TypeError("filename must be a str or bytes object, or a file")
    return io.TextIOWrapper(fobj, encoding, errors, newline) if text else fobj


class _PipelinedDeflate:
    """What GzipNGFile compresses with: the face of a compressobj (compress, flush; `_crc`), the machinery of the threaded writer.
    Input is collected in a pooled buffer; a full buffer (8 MiB at first, twice as much each time, 64 MiB at most) is compressed by
    the engine on a thread of its own while the caller fills the second buffer; every batch is primed by the 32 KiB in front of it
    and ends on a sync-flush boundary, so the pieces are ONE raw deflate stream over one window -- the stream
    `zlib_ng.compressobj(level, DEFLATED, -15)` gives from the engine's own batching, with the engine call and the caller side by
    side instead of one after the other (the reference's benchmark_scripts/gzipwrite128kblocks.py writes through this object:
    4.4 -> 9 GB/s).  compress() hands out what the batch before has produced; flush() compresses what is collected, waits, and with
    Z_FINISH ends the stream (`03 00`: an empty final block)."""
    _R = 32768
    _UNLOCKED_COPY_FROM = 32 * 1024

    def __init__(self, level):
        zlib_ng._check_level(level)
        self._level = level
        self._limit, self._limit_max = 8 << 20, 64 << 20
        self._buf, self._view, self._addr, self._other = None, None, 0, None
        self._n = 0                 # bytes collected in _buf behind the room for the tail
        self._t = 0                 # bytes of history in front of them (the tail of the batch before)
        self._thread, self._box = None, None
        self._crc = 0
        self._calls = 0
        self._finished = False

    def _adopt(self, buf):
        self._buf, self._view = buf, memoryview(buf)
        anchor = ctypes.c_char.from_buffer(buf)
        self._addr = ctypes.addressof(anchor)
        del anchor

    def _work(self, view, t, n, box):
        try:
            cap = n + (n // _lib.UNIT_MAX + 1) * 64 + 64
            packed, crcs, over, _ = zlib_ng._ctx().deflate_blocks(view, [(t, n, t, 0)], self._level, cap, joined=True)
            if over:
                raise zlib_ng.error("Error -5 while compressing data: incomplete or truncated stream")
            box["out"], box["crc"] = packed, crcs[0]
        except BaseException as exc:                  # raised by the caller's next call
            box["error"] = exc

    def _collect(self):
        """Output of the batch that is under way (waits for it), b"" when there is none."""
        th, box = self._thread, self._box
        if th is None and box is None:
            return b""
        self._thread = self._box = None
        if th is not None:
            th.join()
        if "error" in box:
            raise box["error"]
        self._crc = zlib_ng.crc32_combine(self._crc, box["crc"], box["n"])
        return box["out"]

    def _submit(self, wait):
        """The collected bytes as one engine batch; the buffers change places and the batch's last 32 KiB go in front of the next."""
        n, t, R = self._n, self._t, self._R
        view = self._view[R - t:R + n]
        box = {"n": n}
        other = self._other if self._other is not None else _lib.take_buffer(R + self._limit_max)
        keep = min(R, t + n)
        memoryview(other)[R - keep:R] = self._view[R + n - keep:R + n]
        self._other = self._buf
        self._adopt(other)
        self._n, self._t = 0, keep
        self._limit = min(self._limit_max, 2 * self._limit)
        if wait or sys.is_finalizing():
            self._work(view, t, n, box)
            self._box = box
        else:
            self._thread = threading.Thread(target=self._work, args=(view, t, n, box), name="zng-amd-gzip-batch")
            self._box = box
            self._thread.start()

    def compress(self, data):
        if self._finished:
            raise ValueError("Inconsistent stream state")
        mv = data if isinstance(data, memoryview) else memoryview(data)
        if mv.format != "B" or mv.ndim != 1:
            mv = mv.cast("B")
        out, pos, total, R = [], 0, mv.nbytes, self._R
        if self._buf is None:
            self._adopt(_lib.take_buffer(R + self._limit_max))
        while pos < total:
            room = self._limit - self._n
            if room <= 0:
                out.append(self._collect())        # (the batch in front has to be through before the next starts: one window)
                self._submit(wait=False)
                continue
            k = min(room, total - pos)
            self._calls += 1
            if k >= self._UNLOCKED_COPY_FROM and (self._calls & 7) == 0 and self._thread is not None:
                src, keep = _lib._addr(mv[pos:pos + k])         # (with the interpreter lock released now and then: see gzip_ng_threaded)
                ctypes.memmove(self._addr + R + self._n, src, k)
                del keep
            else:
                self._view[R + self._n:R + self._n + k] = mv[pos:pos + k]
            self._n += k
            pos += k
        res = [x for x in out if x]
        return res[0] if len(res) == 1 else b"".join(res)

    def flush(self, mode=zlib_ng.Z_FINISH):
        if mode == zlib_ng.Z_NO_FLUSH or self._finished:
            return b""
        out = [self._collect()]
        if self._n:
            self._submit(wait=True)
            out.append(self._collect())
        if mode == zlib_ng.Z_FULL_FLUSH:
            self._t = 0                              # the history is forgotten: what follows can be decoded on its own
        if mode == zlib_ng.Z_FINISH:
            out.append(b"\x03\x00")
            self._finished = True
            for b in (self._buf, self._other):
                if b is not None:
                    _lib.give_buffer(b)
            self._view = None
            self._buf = self._other = None
        return b"".join(x for x in out if x)


class GzipNGFile(gzip.GzipFile):
    """gzip.GzipFile whose deflate / inflate / CRC work runs on the GPU engine (gzip_ng.py:98-176)."""

    def __init__(self, filename=None, mode=None, compresslevel=_COMPRESS_LEVEL_BEST, fileobj=None, mtime=None):
        super().__init__(filename, mode, compresslevel, fileobj, mtime)
        if self.mode == WRITE:
            self.compress = _PipelinedDeflate(compresslevel)
            self._small, self._small_n = [], 0       # writes below _SMALL_WRITE wait here for a batch
        elif self.mode == READ:
            self._buffer = io.BufferedReader(_GzipReader(self.fileobj, READ_BUFFER_SIZE))

    def __repr__(self):
        return "<gzip_ng " + repr(self.fileobj)[1:-1] + " " + hex(id(self)) + ">"

    # gzip.GzipFile.close() writes `self.crc` after flushing the compressor.  The deflate kernels already produce the
    # CRC-32 of every block they compress, so the running value is taken from the compressor (complete once it has been
    # flushed) instead of sending every write() to the GPU a second time.
    @property
    def crc(self):
        c = getattr(self, "compress", None)
        return c._crc if (getattr(self, "mode", None) == WRITE and isinstance(c, (_PipelinedDeflate, zlib_ng._Compress))) else self._crc_read

    @crc.setter
    def crc(self, value):
        self._crc_read = value

    def write(self, data):
        self._check_not_closed()
        if self.mode != WRITE:
            import errno
            raise OSError(errno.EBADF, "write() on read-only GzipNGFile object")
        if self.fileobj is None:
            raise ValueError("write() on closed GzipNGFile object")
        view = data if isinstance(data, bytes) else memoryview(data)
        nbytes = len(data) if isinstance(data, bytes) else view.nbytes
        if nbytes == 0:
            return 0
        if nbytes < _SMALL_WRITE:
            # line-sized writes (the reference's benchmark_scripts/gzipwritelines.py): collected here and handed to the
            # compressor a MiB at a time -- a call into the engine per line would cost more than the line
            self._small.append(data if isinstance(data, bytes) else bytes(view))
            self._small_n += nbytes
            if self._small_n >= _SMALL_BATCH:
                self._drain_small()
        else:
            self._drain_small()
            out = self.compress.compress(view)
            if out:                                  # (most calls only add to the engine's batch)
                self.fileobj.write(out)
        self.size += nbytes
        self.offset += nbytes
        return nbytes

    def _drain_small(self):
        if self._small_n:
            parts, self._small, self._small_n = self._small, [], 0
            out = self.compress.compress(parts[0] if len(parts) == 1 else b"".join(parts))
            if out:
                self.fileobj.write(out)

    def flush(self, zlib_mode=zlib_ng.Z_SYNC_FLUSH):
        if self.mode == WRITE and self.fileobj is not None:
            self._drain_small()
        return super().flush(zlib_mode)

    def close(self):
        if self.mode == WRITE and self.fileobj is not None:
            try:
                self._drain_small()
            except Exception:
                self._small, self._small_n = [], 0
                super().close()
                raise
        super().close()


GzipFile = GzipNGFile
_GzipNGReader = _GzipReader


def compress(data, compresslevel=_COMPRESS_LEVEL_BEST, *, mtime=None):
    """One-shot gzip member (gzip_ng.py:184-197): engine output with wbits=31, then mtime and OS=255 patched in."""
    member = zlib_ng.compress(data, level=compresslevel, wbits=31)
    stamp = int(time.time() if mtime is None else mtime)
    return struct.pack("<4sLBB", member[:4], stamp, member[8], 255) + member[10:]


def decompress(data):
    """One-shot gunzip of any number of members (gzip_ng.py:200-205).  The whole input goes to the engine in one call
    and the result is the engine's output object; anything but a clean decode is replayed through the reader, which
    raises the reference's exceptions."""
    mv = memoryview(data)
    if mv.contiguous and mv.nbytes >= 18:
        mv = mv.cast("B") if (mv.format != "B" or mv.ndim != 1) else mv
        ctx = zlib_ng._ctx()
        isize = int.from_bytes(mv[mv.nbytes - 4:], "little")
        # (ISIZE is untrusted input: never more than what deflate can expand this many bytes to)
        cap = max(1 << 16, min(isize, 1032 * mv.nbytes) + 64, 4 * mv.nbytes)
        for _ in range(4):
            code, out, _n = ctx.gunzip(mv, cap)
            if code == 0:
                return out
            if code == zlib_ng._lib.BUF_ERROR and ctx.last_needed > cap:
                # what is reported is what the members seen so far need: with more members behind them, at least double
                cap = max(ctx.last_needed + 64, 2 * cap)
                continue
            break
    return _GzipReader(data).readall()


# ---- command line (reference: python -m zlib_ng.gzip_ng, gzip_ng.py:208-317) -------------------------------------
def _argument_parser():
    import argparse
    ap = argparse.ArgumentParser(
        prog="python -m zlib_ng_amd.gzip_ng",
        description="gzip-compatible (de)compression on the MI355X engine. Reads stdin when no file is given.")
    ap.add_argument("file", nargs="?")
    lv = ap.add_mutually_exclusive_group()
    for n in range(1, 10):
        names = [f"-{n}"] + (["--fast"] if n == 1 else ["--best"] if n == 9 else [])
        lv.add_argument(*names, action="store_const", dest="compresslevel", const=n,
                        help=f"compression level {n}" + (" (default 6)" if n == 6 else ""))
    lv.add_argument("-d", "--decompress", action="store_true", help="decompress")
    out = ap.add_mutually_exclusive_group()
    out.add_argument("-c", "--stdout", action="store_true", help="write to standard output")
    out.add_argument("-o", "--output", help="write to this file")
    ap.add_argument("-n", "--no-name", action="store_true", help="do not store the file name and time stamp")
    ap.add_argument("-f", "--force", action="store_true", help="overwrite existing output")
    ap.add_argument("-b", "--buffer-size", type=int, default=READ_BUFFER_SIZE, help="bytes read per request")
    ap.set_defaults(compresslevel=_COMPRESS_LEVEL_TRADEOFF)
    return ap


def main(argv=None):
    import os
    import shutil
    import sys
    args = _argument_parser().parse_args(argv)
    global READ_BUFFER_SIZE
    READ_BUFFER_SIZE = args.buffer_size
    decompress_mode = args.decompress
    src_name = args.file
    if args.output:
        dst_name = args.output
    elif args.stdout or src_name is None:
        dst_name = None
    elif decompress_mode:
        stem, ext = os.path.splitext(src_name)
        if ext not in (".gz", ".tgz"):
            sys.exit(f"filename doesn't end in .gz: {src_name!r}. Cannot determine output filename.")
        dst_name = stem + (".tar" if ext == ".tgz" else "")
    else:
        dst_name = src_name + ".gz"
    if dst_name is not None and os.path.exists(dst_name) and not args.force:
        answer = input(f"{dst_name} already exists; do you wish to overwrite (y/n)? ")
        if answer.strip().lower() not in ("y", "yes"):
            sys.exit("not overwritten")
    raw_in = sys.stdin.buffer if src_name is None else builtins_open(src_name, "rb")
    raw_out = sys.stdout.buffer if dst_name is None else builtins_open(dst_name, "wb")
    try:
        if decompress_mode:
            with GzipNGFile(fileobj=raw_in, mode="rb") as gz:
                shutil.copyfileobj(gz, raw_out, args.buffer_size)
        else:
            kw = {"filename": "", "mtime": 0} if (args.no_name or src_name is None) else {"filename": os.path.basename(src_name)}
            with GzipNGFile(fileobj=raw_out, mode="wb", compresslevel=args.compresslevel, **kw) as gz:
                shutil.copyfileobj(raw_in, gz, args.buffer_size)
    finally:
        if raw_in is not sys.stdin.buffer:
            raw_in.close()
        if raw_out is not sys.stdout.buffer:
            raw_out.close()


import builtins as _builtins  # noqa: E402

builtins_open = _builtins.open

if __name__ == "__main__":  # pragma: no cover
    main()
