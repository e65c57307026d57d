"""gzip_ng_threaded -- drop-in face of the reference's block-parallel gzip reader / writer
(reference src/zlib_ng/gzip_ng_threaded.py:22-437), with the blocks compressed on the GPU.

Reference design: N worker threads each own a `_ParallelCompress`; `write()` cuts the stream into blocks
of at most `block_size`, primes block i with the last 32 KiB of block i-1 and deals blocks round-robin to
the workers; a writer thread drains results in order and folds CRCs with crc32_combine.

Here the data-parallel axis is the GPU, not threads: `threads` is the number of blocks kept in flight and the upper bound
of GPUs one writer uses -- with threads > 1 every batch is cut into contiguous block ranges, one per visible GPU (or per
entry of ZNGAMD_DEVICES), each range primed by the input in front of it, compressed side by side and written in order
(`_lib.deflate_blocks_multi`; a process that its launcher bound to one GPU stays on it).
`write()` cuts and primes blocks exactly as the reference does, but does not hand them over one by one: the bytes are collected in
a pooled buffer (one copy per byte; every eighth large copy with the interpreter lock released), and a full buffer -- 8 MiB at
first, twice as much each time, 64 MiB at most -- goes to the engine as ONE batch on a thread of its own
(`zngamd_deflate_blocks_packed`: the blocks' outputs come back packed, in one piece, into one of two output buffers) while the
caller fills the second buffer; the file write of a batch runs beside the compression of the next.  The reference's queue
interface is still there (`input_queues`: what is put on them is drained by one worker thread in round-robin order and
compressed as a batch, `_ParallelCompress.compress_and_crc_batch`).  The byte stream has the reference's framing: 10-byte
header (with its OS/XFL byte order), sync-flushed raw-deflate blocks, `03 00`, CRC32, ISIZE; `flush()` ends the member and
starts a new one; `close()` after the buffered writer's implicit flush leaves a trailing empty member.
"""
import builtins
import collections
import ctypes
import io
import multiprocessing
import os
import queue
import struct
import sys
import threading

from . import _lib, gzip_ng, zlib_ng

DEFLATE_WINDOW_SIZE = 2 ** 15
_UNLOCKED_COPY_FROM = 32 * 1024          # pieces from this size on are copied with the interpreter lock released


def open(filename, mode="rb", compresslevel=gzip_ng._COMPRESS_LEVEL_TRADEOFF, encoding=None, errors=None,
         newline=None, *, threads=1, block_size=1024 * 1024, indexed_members=None, exact_framing=None):
    """Like gzip.open for streamed reading / writing (no seeking).  threads == 0 defers to gzip_ng.open,
    threads < 0 uses the CPU count (gzip_ng_threaded.py:22-75).

    indexed_members (an addition; writing only; None = the environment's ZNGAMD_WRITER_MEMBERS, off when unset): the file is
    written as independent gzip members of at most 128 KiB with this engine's chunk index in their FEXTRA field -- still a
    gzip file for every gzip reader (RFC 1952 members, as bgzip writes them), and the format this engine's reader decodes
    with one wavefront per member instead of through the chunk pipeline.  Off, the byte stream is the reference's.

    exact_framing (an addition; writing only; None = the environment's ZNGAMD_WRITER_EXACT, off when unset): the single-member
    layout of the reference without this engine's additions -- no flat block headers, no segment index in the members behind the
    data (see _ThreadedGzipWriter)."""
    if threads == 0:
        return gzip_ng.open(filename, mode, compresslevel, encoding, errors, newline)
    if threads < 0:
        try:
            threads = len(os.sched_getaffinity(0))
        except Exception:
            try:
                threads = multiprocessing.cpu_count()
            except Exception:
                threads = 1
    if "r" in mode:
        stream = io.BufferedReader(_ThreadedGzipReader(filename, block_size=block_size))
    else:
        # The reference buffers block_size bytes in front of its writer (gzip_ng_threaded.py:70-75).  The writer below collects
        # small writes itself, so this buffer only has to spare it a Python call per line: at most 64 KiB - 1, which lets the
        # reference's benchmark pattern (128 KiB per call, benchmark_scripts/gzipwrite128kblocks.py) and anything larger reach
        # the writer without being copied here first -- io.BufferedWriter hands on what is longer than its buffer as it is.
        stream = FlushableBufferedWriter(
            _ThreadedGzipWriter(filename, mode.replace("t", "b"), block_size=block_size, level=compresslevel,
                                threads=threads, indexed_members=indexed_members, exact_framing=exact_framing),
            buffer_size=max(io.DEFAULT_BUFFER_SIZE, min(block_size, 1 << 16) - 1))
    return io.TextIOWrapper(stream, encoding, errors, newline) if "t" in mode else stream


def open_as_binary_stream(filename, open_mode):
    if isinstance(filename, (str, bytes)) or hasattr(filename, "__fspath__"):
        return builtins.open(filename, open_mode), True
    if hasattr(filename, "read") or hasattr(filename, "write"):
        return filename, False
    raise TypeError("filename must be a str or bytes object, or a file")


class _ThreadedGzipReader(io.RawIOBase):
    """Streamed reading with one pump thread.  The windowed GPU reader (`zlib_ng._GzipReader`) decodes a whole window of the
    compressed file per engine call; the pump takes each decoded window over as it is (no per-block objects, no copy) and
    parks one while the next is being read, sent and decoded, so that the consumer's copying out of window n overlaps the
    file I/O, PCIe and kernels of window n + 1 (the engine call and the file read release the GIL).  readinto() copies
    straight from the parked window into the caller's buffer: one copy per byte in all.  Same role as the reference's
    prefetching reader (gzip_ng_threaded.py:90-167), whose queue of `queue_size` blocks of `block_size` bytes this
    replaces; both arguments are accepted and only size the inner reader's first request."""

    def __init__(self, filename, queue_size=2, block_size=1024 * 1024):
        source, owns = open_as_binary_stream(filename, "rb")
        self.raw, self.closefd = source, owns
        self.fileobj = zlib_ng._GzipReader(source, buffersize=8 * block_size)
        self.fileobj._decode_ahead = False           # (the pump below is what decodes ahead here)
        self.block_size = block_size
        self.pos = 0
        # (the reference's tests pass a mode string in this position; anything that is not a positive count means 2)
        self._depth = 1                              # decoded windows parked ahead of the one being consumed
        self._parked = collections.deque()           # decoded pieces waiting for the consumer
        self._cv = threading.Condition()
        self._finished = False                       # pump has ended (end of stream or failure)
        self._failure = None
        self._current = memoryview(b"")
        self._current_addr, self._current_keep = 0, None   # where the unread rest of the window lies (copies without the interpreter lock)
        self._calls = 0
        self._token = None                           # buffer of the window being consumed (goes back to the inner reader)
        self._closed = False
        self._stop = False
        self._owner = threading.current_thread()
        self._pump_thread = threading.Thread(target=self._pump, name="zng-amd-reader")
        self._pump_thread.start()

    def _wanted(self):
        return not self._stop and self._owner.is_alive()

    def _pump(self):
        try:
            while self._wanted():
                piece = self.fileobj._take_window()
                if piece is None:
                    break
                with self._cv:
                    while len(self._parked) >= self._depth and self._wanted():
                        self._cv.wait(0.05)
                    self._parked.append(piece)
                    self._cv.notify_all()
        except Exception as exc:                     # handed to the consumer once the good pieces are gone
            self._failure = exc
        finally:
            with self._cv:
                self._finished = True
                self._cv.notify_all()

    def _next_piece(self):
        with self._cv:
            while not self._parked and not self._finished:
                self._cv.wait(0.05)
            if self._parked:
                piece = self._parked.popleft()
                self._cv.notify_all()
                return piece
        if self._failure is not None:
            raise self._failure
        return None

    def readinto(self, b):
        if self._closed:
            raise ValueError("I/O operation on closed file")
        out = memoryview(b).cast("B")
        if not len(self._current):
            self._current = memoryview(b"")
            self.fileobj._give_back(self._token)         # the window just finished can be decoded into again
            self._token = None
            piece = self._next_piece()
            if piece is None:
                return 0
            self._current, self._token = piece
            self._current_addr, self._current_keep = _lib._addr(self._current)
            self._current_addr = self._current_addr.value or 0
        n = min(len(out), len(self._current))
        self._calls += 1
        if n >= _UNLOCKED_COPY_FROM and (self._calls & 7) == 0 and self._current_addr and not out.readonly:
            # (the pump thread needs the interpreter lock between its file read and its engine call: a consumer that always
            # copies with the lock held makes it wait for the switch interval each time; every eighth copy lets go of it)
            anchor = ctypes.c_char.from_buffer(out)
            ctypes.memmove(ctypes.addressof(anchor), self._current_addr, n)
            del anchor
        else:
            out[:n] = self._current[:n]
        self._current = self._current[n:]
        self._current_addr += n
        self.pos += n
        return n

    def readable(self):
        return True

    def tell(self):
        if self._closed:
            raise ValueError("I/O operation on closed file")
        return self.pos

    def close(self):
        if self._closed:
            return
        self._closed = True
        self._stop = True
        with self._cv:
            self._cv.notify_all()
        self._pump_thread.join()
        self._current = memoryview(b"")
        self._current_addr, self._current_keep = 0, None
        self.fileobj._give_back(self._token)         # window buffers go back to the inner reader, and with it to the pool
        self._token = None
        while self._parked:
            self.fileobj._give_back(self._parked.popleft()[1])
        self.fileobj.close()
        if self.closefd:
            self.raw.close()

    @property
    def closed(self):
        return self._closed


class FlushableBufferedWriter(io.BufferedWriter):
    def flush(self):
        super().flush()
        self.raw.flush()


class _ThreadedGzipWriter(io.RawIOBase):
    """Block-parallel gzip writer; see the module docstring for how the reference's thread fan-out maps
    onto one engine batch per drain."""

    def __init__(self, filename, mode="wb", level=zlib_ng.Z_DEFAULT_COMPRESSION, threads=1, queue_size=1,
                 block_size=1024 * 1024, indexed_members=None, exact_framing=None):
        self._closed = True           # so that __del__/__exit__ are harmless if __init__ fails
        if indexed_members is None:
            indexed_members = os.environ.get("ZNGAMD_WRITER_MEMBERS", "0") not in ("", "0")
        self._members = bool(indexed_members)
        # (r06) The single-member framing of the reference (header | sync-flushed blocks | 03 00 | CRC ISIZE | empty member) is kept
        # byte for byte in its LAYOUT; two things differ unless exact_framing (ZNGAMD_WRITER_EXACT=1) asks for the r05 bytes: a block
        # of at most 64 KiB is cut into 2 KiB segments like every larger one (FLAG_SEG2K; the blocks of full size are the same bytes
        # either way), and the member is followed by EMPTY members whose FEXTRA field holds the blocks' segment index (_lib.INDEX_*)
        # -- any gzip reader skips them, this engine's reader decodes the units of a window side by side with them (19.7 against
        # 39.5 ms per 4 GiB on the device).
        if exact_framing is None:
            exact_framing = os.environ.get("ZNGAMD_WRITER_EXACT", "0") not in ("", "0")
        self._exact = bool(exact_framing) or self._members
        self._bflag = 0 if self._exact else _lib.FLAG_SEG2K
        self._index_ok = not self._exact
        self._index_recs = []                        # packed records of the member being written (one array per batch)
        self._index_units = 0
        self._member_bytes = 0                       # bytes of the member being written (header to trailer)
        self._member_size = max(1, min(block_size, 128 * 1024))
        self._members_written = 0                    # bytes of members written so far
        if "t" in mode or "r" in mode:
            raise ValueError("Only binary writing is supported")
        if "b" not in mode:
            mode += "b"
        if threads < 1:
            raise ValueError(f"threads should be at least 1, got {threads}")
        self.lock = threading.Lock()
        self._calling_thread = threading.current_thread()
        self.exception = None
        self.level = level
        self.previous_block = b""
        self.block_size = block_size
        # incompressible data grows a little; 10 % head-room as in gzip_ng_threaded.py:229-231
        self.compressors = [zlib_ng._ParallelCompress(buffersize=block_size + max(block_size // 10, 500), level=level)]
        self.threads = threads
        self._ctxs = None                            # contexts of the GPUs this writer spreads its batches over (made on first use)
        # The reference keeps queue_size blocks per worker thread in flight.  One engine batch replaces the N worker
        # threads, and a batch of 8 blocks is all launch overhead: the queues are made deep enough for about 32 MiB
        # of pending input per batch (a memory bound of the same kind as the reference's threads * queue_size blocks).
        depth = max(queue_size, -(-(32 << 20) // (max(block_size, 1) * threads)))
        self._batch_blocks = threads * depth
        self.input_queues = [queue.Queue(depth) for _ in range(threads)]
        self.output_queues = []
        self.compression_workers = []
        self.output_worker = threading.Thread(target=self._compress_and_write)
        self.index = 0
        self._drain_index = 0
        self._crc = 0
        self._size = 0
        self._write_thread = None                    # file write of the last bulk batch, still running
        # writes below this are collected up to this many bytes: a batch of 64 MiB is 512 units of work for 1 024 SIMDs -- the
        # kernels of a 32 MiB batch took 1.7 ms, four times the time per byte of a full device
        self._coalesce_limit = max(8 * block_size, 64 << 20)
        # ... reached in steps: the first batch is 8 MiB, every later one twice the one before -- the engine starts on the first
        # bytes while the caller is still writing (a file of 256 MiB in four batches of 64 spent two of its five time slots
        # with only one side working)
        self._batch_limit = min(self._coalesce_limit, max(8 * block_size, 8 << 20))
        self._calls = 0
        self._small, self._small_n = None, 0
        self._small_view = None                      # writable view of _small: a slice assignment through it is one memcpy
        self._small_addr = 0                         # its address: large pieces are copied with the interpreter lock released
        self._small_other = None                     # the second collecting buffer: one fills while the other is compressed
        self._packed = [None, None]                  # output buffers of the batches, used in turn (one is written to the file
        self._packed_turn = 0                        # while the next batch fills the other): warm memory, no object per batch
        self._table_key, self._table = None, None    # block table of the last batch: batches of one shape share it
        self._batch_thread, self._batch_error = None, None
        self._write_error = None
        self.running = False
        self.raw, self.closefd = open_as_binary_stream(filename, mode)
        self._closed = False
        self._write_gzip_header()
        self.start()

    def _check_closed(self, msg=None):
        if self._closed:
            raise ValueError("I/O operation on closed file")

    def _write_gzip_header(self):
        if self._members:                            # every member brings its own header
            return
        # gzip_ng_threaded.py:269-284: note the order of the last two bytes (OS, then XFL)
        xfl = 2 if self.level == zlib_ng.Z_BEST_COMPRESSION else 4 if self.level == zlib_ng.Z_BEST_SPEED else 0
        self.raw.write(struct.pack("BBBBIBB", 0x1f, 0x8b, 8, 0, 0, 0xff, xfl))
        self._member_bytes = 10

    def start(self):
        self.running = True
        self.output_worker.start()

    def stop(self):
        """Stop without caring for queued work."""
        self.running = False
        for q in self.input_queues:                  # wake the worker now: it looks at `running` every 50 ms otherwise, and
            try:                                     # close() of a file of 256 MiB spent most of its time in this join
                q.put_nowait(None)
            except queue.Full:
                pass
        self.output_worker.join()

    def write(self, b):
        if self._closed:
            raise ValueError("I/O operation on closed file")
        if self.exception is not None:
            with self.lock:
                raise self.exception
        nbytes = b.nbytes if isinstance(b, memoryview) else len(b)
        if nbytes >= self._coalesce_limit:
            self._flush_small()
            return self._write_bulk(memoryview(b).cast("B"), nbytes)
        if nbytes == 0:
            return 0
        # Small writes (the reference's benchmark pattern: 128 KiB per call) are collected in one buffer and go to the engine
        # like one large write: the stream is cut into block_size blocks, each primed by the 32 KiB in front of it, exactly as
        # if the caller had written the collected bytes at once.  (One queue round trip and one engine batch per call cost
        # more than the compression itself.)
        R = DEFLATE_WINDOW_SIZE
        if self._small is None:
            self._adopt_small(_lib.take_buffer(R + self._coalesce_limit))  # [room for the 32 KiB in front][collected bytes]
        n = self._small_n
        if n + nbytes > self._batch_limit:
            self._flush_small(wait=False)            # the full buffer is compressed and written while the other one fills
            n = 0
            if nbytes > self._batch_limit:
                self._batch_limit = min(self._coalesce_limit, max(2 * self._batch_limit, nbytes))
        self._calls += 1
        if nbytes >= _UNLOCKED_COPY_FROM and (self._calls & 7) == 0 and self._batch_thread is not None:
            # While a batch is under way, every eighth large piece is copied with the interpreter lock released (a foreign call):
            # the batch thread needs the lock between its engine call, the file write and its bookkeeping, and a caller that never
            # lets go of it -- a loop of write() calls does not -- makes every one of those wait for the interpreter's switch
            # interval (5 ms; a batch takes 4).  Not every piece: taking the lock back costs the caller several microseconds
            # whenever another thread holds it (16 us per 128 KiB call against 9).
            src, keep = _lib._addr(b)
            ctypes.memmove(self._small_addr + R + n, src, nbytes)
            del keep
        else:
            try:
                self._small_view[R + n:R + n + nbytes] = b          # (a bytearray slice assignment copies a memoryview twice)
            except (TypeError, ValueError):                         # not a flat byte buffer: through the bytearray
                self._small[R + n:R + n + nbytes] = b
        self._small_n = n + nbytes
        return nbytes

    def _adopt_small(self, buf):
        self._small = buf
        self._small_view = memoryview(buf)
        anchor = ctypes.c_char.from_buffer(buf)
        self._small_addr = ctypes.addressof(anchor)                 # (the bytearray is never resized: the address stays)
        del anchor

    def _flush_small(self, wait=True):
        """The collected bytes as ONE engine batch: the tail of what was written before is put in front of them in the
        same buffer, so the first block is primed like every other.  The batch runs on a thread of its own (the engine call
        and the file write release the GIL) while the caller goes on filling the second buffer -- the reference's worker
        threads compress while write() returns (gzip_ng_threaded.py:299-322); wait=True returns when the batch is through."""
        n, self._small_n = self._small_n, 0
        if n:
            self._batch_limit = min(self._coalesce_limit, 2 * self._batch_limit)
            self._join_batch()                       # batches are strictly in order: CRC folding and the file are sequential
            for q in self.input_queues:
                q.join()
            with self.lock:
                if self.exception:
                    raise self.exception
            R, bs, buf = DEFLATE_WINDOW_SIZE, self.block_size, self._small
            if self._members:                        # independent members: no history in front, no block table
                view, blocks, table = memoryview(buf)[R:R + n], None, None
            else:
                tail = memoryview(self.previous_block)[-R:]
                t = tail.nbytes
                buf[R - t:R] = tail
                view = memoryview(buf)[R - t:R + n]
                key = (t, n, bs)
                if key != self._table_key:
                    blocks = [(t + o, min(bs, n - o), min(R, t + o), self._bflag) for o in range(0, n, bs)]
                    self._table_key, self._table = key, (blocks, _lib.block_table(blocks))
                blocks, table = self._table
                last = n - ((n - 1) // bs) * bs
                self.previous_block = bytes(buf[R + n - last:R + n])
            self._size += n
            if wait or sys.is_finalizing():
                self._emit(view, blocks, table)
            else:
                self._batch_thread = threading.Thread(target=self._emit_guarded, args=(view, blocks, table), name="zng-amd-writer-batch")
                self._batch_thread.start()
                other = self._small_other if self._small_other is not None else _lib.take_buffer(len(buf))
                self._small_other = buf
                self._adopt_small(other)
        if wait:
            self._join_batch()

    def _emit_guarded(self, view, blocks, table):
        try:
            self._emit(view, blocks, table)
        except BaseException as exc:                 # raised by the next call of the owner
            self._batch_error = exc

    def _join_batch(self):
        t, self._batch_thread = self._batch_thread, None
        if t is not None:
            t.join()
        if self._batch_error is not None:
            exc, self._batch_error = self._batch_error, None
            with self.lock:
                self.exception = exc
            raise exc

    def _write_bulk(self, view, nbytes):
        """A write of many blocks at once: same cutting and priming as block by block, but the blocks go to the engine as
        slices of one buffer (no per-block objects, no queue round trips).  Queued blocks are finished first."""
        self._join_batch()
        for q in self.input_queues:
            q.join()
        with self.lock:
            if self.exception:
                raise self.exception
        if self._members:
            step = 256 << 20
            for lo in range(0, nbytes, step):
                self._emit(view[lo:min(nbytes, lo + step)], None)
            self._size += nbytes
            return nbytes
        tail = bytes(memoryview(self.previous_block)[-DEFLATE_WINDOW_SIZE:])
        bs = self.block_size
        emit = self._emit

        # the first block needs the tail of what was written before: a small buffer of its own; every later block is primed
        # by the bytes in front of it in the caller's buffer, which goes to the engine as it is (no copy of the payload)
        first = min(bs, nbytes)
        emit(tail + bytes(view[:first]), [(len(tail), first, len(tail), self._bflag)])
        step = max(bs, (256 << 20) // bs * bs)                       # engine batches of at most 256 MiB
        src = view.obj if isinstance(view.obj, bytes) and len(view.obj) == nbytes else view
        lo = first
        while lo < nbytes:
            hi = min(nbytes, lo + step)
            if hi - first == nbytes - first:                          # everything in one batch: the caller's buffer itself
                emit(src, [(o, min(bs, hi - o), min(DEFLATE_WINDOW_SIZE, o), self._bflag) for o in range(lo, hi, bs)])
            else:                                                      # a slice with 32 KiB of history in front
                d = min(DEFLATE_WINDOW_SIZE, lo)
                emit(view[lo - d:hi], [(d + o - lo, min(bs, hi - o), min(DEFLATE_WINDOW_SIZE, d + o - lo), self._bflag) for o in range(lo, hi, bs)])
            lo = hi
        self._size += nbytes
        last = nbytes - ((nbytes - 1) // bs) * bs                    # the next block is primed with the last block, as always
        self.previous_block = bytes(view[nbytes - last:nbytes])
        return nbytes

    def _emit(self, buf, blocks, table=None):
        """One engine batch (spread over the writer's GPUs) and the write of its output."""
        if self._members:
            # indexed members (zngamd_gzip_members): header, chunk index, FINAL deflate block, CRC32 and ISIZE per member, in one piece
            ctx = self._contexts()[0]
            turn = self._packed_turn
            self._packed_turn = turn ^ 1
            need = ctx.gzip_members_room(memoryview(buf).nbytes, self._member_size)
            into = self._packed[turn]
            if into is None or len(into) < need:                     # (the file write of the batch before last, which used it, is through)
                _lib.give_buffer(into)
                into = self._packed[turn] = _lib.take_buffer(need)
            packed = ctx.gzip_members(buf, self._member_size, self.level, into=into)
            self._members_written += len(packed)
            self._settle_write()
            if not isinstance(self.raw, (io.FileIO, io.BufferedWriter, io.BufferedRandom, io.BytesIO)):
                packed = bytes(packed)               # an object of the caller's: it may keep what it is given
            if sys.is_finalizing():
                self.raw.write(packed)
                return
            self._write_thread = threading.Thread(target=self._write_later, args=(packed,), name="zng-amd-writer-io")
            self._write_thread.start()
            return
        cap = self.block_size + max(self.block_size // 10, 500)
        turn = self._packed_turn
        self._packed_turn = turn ^ 1
        need = len(blocks) * cap
        into = self._packed[turn]
        if into is None or len(into) < need:                     # (the file write of the batch before last, which used it, is through)
            _lib.give_buffer(into)
            into = self._packed[turn] = _lib.take_buffer(need)
        recs = [] if self._index_ok else None
        packed, crcs, overflowed, _ = _lib.deflate_blocks_multi(self._contexts(), buf, blocks, self.level, cap, into=into, table=table, index=recs)
        if overflowed:
            raise OverflowError(f"Compressed output exceeds buffer size of {cap}")
        self._crc = _lib.crc32_combine_many(self._crc, crcs, [b[1] for b in blocks])
        self._member_bytes += len(packed)
        if self._index_ok:
            # the batch's segment index, range by range of the contexts that compressed it; a batch that comes back without one
            # leaves the member without an index (it is all or nothing)
            if recs and sum(int(r["in_len"].astype("int64").sum()) for r in recs) == len(packed):
                self._index_recs.extend(recs)
                self._index_units += sum(len(r) for r in recs)
            else:
                self._index_ok = False
        # the file write of this batch runs beside the compression of the next one (the engine call and the write
        # both release the GIL); the previous batch's write has to be through first: the order is the stream
        self._settle_write()
        if not isinstance(self.raw, (io.FileIO, io.BufferedWriter, io.BufferedRandom, io.BytesIO)):
            packed = bytes(packed)                   # an object of the caller's: it may keep what it is given
        if sys.is_finalizing():                      # closed by the garbage collector at exit: a thread started now never runs
            self.raw.write(packed)
            return
        self._write_thread = threading.Thread(target=self._write_later, args=(packed,), name="zng-amd-writer-io")
        self._write_thread.start()

    def _contexts(self):
        if self._ctxs is None:
            self._ctxs = _lib.contexts(self.threads)
        return self._ctxs

    def _write_later(self, packed):
        try:
            self.raw.write(packed)
        except Exception as exc:                     # raised by the next call that touches the file
            self._write_error = exc

    def _settle_write(self):
        """Wait for a batch write that is still running (and raise what it ran into)."""
        t, self._write_thread = self._write_thread, None
        if t is not None:
            t.join()
        if self._write_error is not None:
            exc, self._write_error = self._write_error, None
            with self.lock:
                self.exception = exc
            raise exc

    def _end_gzip_stream(self, closing=False):
        self._check_closed()
        self._flush_small()
        self._settle_write()
        for q in self.input_queues:
            q.join()
        if self._members:
            if closing and not self._members_written:   # nothing was ever written: an empty member makes it a gzip file
                self.raw.write(struct.pack("BBBBIBB", 0x1f, 0x8b, 8, 0, 0, 0, 0xff) + b"\x03\x00" + bytes(8))
                self._members_written = 20
            self._crc = self._size = 0
            self.raw.flush()
            return
        # empty final block, then CRC32 and ISIZE (gzip_ng_threaded.py:332-338)
        self.raw.write(b"\x03\x00" + struct.pack("<II", self._crc, self._size & 0xFFFFFFFF))
        self._member_bytes += 10
        if self._index_ok and self._index_units:
            self.raw.write(_lib.index_members(self._index_recs, self._member_bytes))
        self._index_recs, self._index_units, self._index_ok = [], 0, not self._exact
        self._crc = 0
        self._size = 0
        self.raw.flush()

    def flush(self):
        self._end_gzip_stream()
        self._write_gzip_header()

    def close(self):
        if self._closed:
            return
        self._end_gzip_stream(closing=True)
        self.stop()
        if self.exception:
            self.raw.close()
            self._closed = True
            raise self.exception
        if self.closefd:
            self.raw.close()
        self._closed = True
        self._release_buffers()

    def _release_buffers(self):
        """The collecting and output buffers go back to the process-wide pool (no batch and no file write is running)."""
        self._small_view = None
        for b in (self._small, self._small_other, self._packed[0], self._packed[1]):
            if b is not None:
                _lib.give_buffer(b)
        self._small = self._small_other = None
        self._packed = [None, None]

    @property
    def closed(self):
        return self._closed

    def _alive(self):
        return self.running and self._calling_thread.is_alive()

    def _compress_and_write(self):
        compressor = self.compressors[0]
        nq = self.threads
        while True:
            # take blocks in the order write() dealt them; block only for the first one
            batch, origins = [], []
            q = self.input_queues[self._drain_index % nq]
            try:
                item = q.get(timeout=0.05)
            except queue.Empty:
                if not self._alive():
                    return
                continue
            if item is None:                         # stop()'s wake-up call
                q.task_done()
                if not self._alive():
                    return
                continue
            batch.append(item)
            origins.append(q)
            self._drain_index += 1
            while len(batch) < self._batch_blocks:
                q = self.input_queues[self._drain_index % nq]
                try:
                    item = q.get_nowait()
                except queue.Empty:
                    break
                if item is None:
                    q.task_done()
                    break
                batch.append(item)
                origins.append(q)
                self._drain_index += 1
            try:
                if self._members:
                    joined = b"".join(bytes(d) for d, _ in batch)
                    out = self._contexts()[0].gzip_members(joined, self._member_size, self.level)
                    self._members_written += len(out)
                    results = [(out, 0)] + [(b"", 0)] * (len(batch) - 1)
                else:
                    results = compressor.compress_and_crc_batch(batch)
            except Exception as exc:
                for q in origins:
                    q.task_done()
                self._set_error_and_empty_queue(exc)
                return
            for (data, _), (compressed, crc) in zip(batch, results):
                self._crc = zlib_ng.crc32_combine(self._crc, crc, len(data))
                self._size += len(data)
            self.raw.write(b"".join(c for c, _ in results))          # one write per batch, blocks in order
            for q in origins:
                q.task_done()

    def _set_error_and_empty_queue(self, error, q=None):
        with self.lock:
            self.exception = error
            self.running = False
            for q in self.input_queues:
                while True:
                    try:
                        q.get(timeout=0.05)
                        q.task_done()
                    except queue.Empty:
                        break

    def writable(self):
        return True
