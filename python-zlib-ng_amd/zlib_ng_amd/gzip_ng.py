"""gzip_ng -- drop-in face of the reference's `zlib_ng.gzip_ng` (reference src/zlib_ng/gzip_ng.py:33-205)
on the MI355X engine: `open`, `GzipNGFile` / `GzipFile`, one-shot `compress` / `decompress`,
`BadGzipFile`, `READ_BUFFER_SIZE`.

The single-stream file object keeps CPython's gzip.GzipFile for header / trailer / buffering logic and
swaps in this package's compressor object and C-ABI backed reader, as the reference does
(gzip_ng.py:140-149).  Compression inside one GzipNGFile is one LZ77 window over the whole file (sequential
by construction); the data-parallel writer is gzip_ng_threaded.
"""
import gzip
import io
import struct
import time

from . import zlib_ng
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
    if _is_pathlike(filename):
        fobj = GzipNGFile(filename, raw_mode, compresslevel)
    elif hasattr(filename, "read") or hasattr(filename, "write"):
        fobj = GzipNGFile(None, raw_mode, compresslevel, filename)
    else:
        raise TypeError("filename must be a str or bytes object, or a file")
    return io.TextIOWrapper(fobj, encoding, errors, newline) if text else fobj


class GzipNGFile(gzip.GzipFile):
    """gzip.GzipFile whose deflate / inflate / CRC work runs on the GPU engine (gzip_ng.py:98-176)."""

    def __init__(self, filename=None, mode=None, compresslevel=_COMPRESS_LEVEL_BEST, fileobj=None, mtime=None):
        super().__init__(filename, mode, compresslevel, fileobj, mtime)
        if self.mode == WRITE:
            self.compress = zlib_ng.compressobj(compresslevel, zlib_ng.DEFLATED, -zlib_ng.MAX_WBITS,
                                                zlib_ng.DEF_MEM_LEVEL, 0)
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
        return c._crc if (getattr(self, "mode", None) == WRITE and isinstance(c, zlib_ng._Compress)) else self._crc_read

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
