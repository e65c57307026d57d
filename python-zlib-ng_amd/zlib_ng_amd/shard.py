"""Block sharding across ranks (multi-GPU leg, one process per GPU).  No torch: RCCL is driven through the C ABI.

The reference's only parallel strategy is round-robin blocks over threads with an in-order writer
(gzip_ng_threaded.py:316-321, :382-398).  Here every rank owns a contiguous range of blocks (`shard_range`), primes the
first block of its range with the 32 KiB of input in front of it (the previous rank's tail, gzip_ng_threaded.py:317) and
compresses the range on its own GPU into one contiguous slice.  One exchange step reassembles the member stream:
  * `Comm.layout`           all-gather of {slice bytes, CRC-32 of the rank's input, input bytes} -- 24 bytes per rank; gives
                            every rank the offset of its slice, the total size and the CRC-32 / length of the whole input
                            for the trailer (crc32_combine is associative: the folds of the ranks fold again);
  * `Comm.allgather_stream` exact-size exchange of the slices: grouped ncclSend / ncclRecv over all xGMI links at once,
                            every slice lands at its offset; runs on a stream of its own so that it overlaps later kernels.
Where nobody needs the whole stream in one memory the layout alone is enough: every rank writes its slice at its own offset.
`exchange_layout` takes any all-gather of three integers per rank (`Comm.layout` on GPUs; the CPU tests hand in a gloo one).
Only integer arithmetic and framing bytes live here; no payload byte is computed on the host.
"""
import ctypes as C
import socket
import struct
import time

from . import _lib


def shard_range(n_blocks, rank, world):
    """Contiguous block range [lo, hi) of `rank` (replaces `index % threads`, gzip_ng_threaded.py:320)."""
    lo = (n_blocks * rank) // world
    hi = (n_blocks * (rank + 1)) // world
    return lo, hi


def combine_crcs(crcs_and_lens):
    """Fold (crc32, uncompressed_len) pairs in order, as _write does (gzip_ng_threaded.py:394)."""
    L = _lib.load()
    crc = 0
    for c, n in crcs_and_lens:
        crc = L.zngamd_crc32_combine(crc, c & 0xFFFFFFFF, n)
    return crc


def layout_from_records(records, rank):
    """records[r] = (slice bytes, crc32 of rank r's input, input bytes of rank r), in rank order ->
    (offset of `rank`'s slice, total bytes, per-rank sizes, CRC-32 of the whole input, whole input length)."""
    sizes = [int(r[0]) for r in records]
    return (sum(sizes[:rank]), sum(sizes), sizes, combine_crcs([(int(r[1]), int(r[2])) for r in records]),
            sum(int(r[2]) for r in records))


def exchange_layout(local_len, crc, ulen, rank, allgather):
    """`allgather((len, crc, ulen))` -> the records of all ranks in rank order; see layout_from_records."""
    return layout_from_records(allgather((int(local_len), int(crc) & 0xFFFFFFFF, int(ulen))), rank)


def gzip_frame(body_len_total, crc, size, level):
    """Header / trailer bytes of the reference's threaded writer (gzip_ng_threaded.py:269-284, :324-338)."""
    xfl = 2 if level == 9 else 4 if level == 1 else 0
    header = struct.pack("BBBBIBB", 0x1f, 0x8b, 8, 0, 0, 0xff, xfl)
    trailer = b"\x03\x00" + struct.pack("<II", crc & 0xFFFFFFFF, size & 0xFFFFFFFF)
    return header, trailer


_RDV_MAGIC = b"ZNGA"


def rendezvous_bytes(rank, world, addr, port, payload=None, timeout=900.0, tries=8):
    """Rank 0 hands `payload` (bytes) to the other ranks over TCP (one short connection each); every rank returns it.
    What a launcher without a key-value store needs to pass the 128-byte RCCL unique id around.  Rank 0 listens on the first
    free port of port .. port+tries-1; the others go round those ports until one answers with the magic word."""
    if world == 1:
        return payload
    if rank == 0:
        srv, err = None, None
        for k in range(tries):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, port + k))
                srv = s
                break
            except OSError as e:
                err = e
                s.close()
        if srv is None:
            raise err
        srv.listen(world)
        srv.settimeout(timeout)
        served = 0
        while served < world - 1:
            conn, _ = srv.accept()
            try:
                conn.settimeout(5.0)
                if _recv_exact(conn, 4) != _RDV_MAGIC:              # not one of ours
                    continue
                conn.sendall(_RDV_MAGIC + struct.pack("<I", len(payload)) + payload)
                served += 1
            except OSError:
                pass
            finally:
                conn.close()
        srv.close()
        return payload
    deadline = time.time() + timeout
    k = 0
    while True:
        try:
            s = socket.create_connection((addr, port + k % tries), timeout=5.0)
            try:
                s.sendall(_RDV_MAGIC)
                if _recv_exact(s, 4) == _RDV_MAGIC:
                    n = struct.unpack("<I", _recv_exact(s, 4))[0]
                    return _recv_exact(s, n)
            finally:
                s.close()
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError(f"rendezvous: nobody answered on {addr}:{port}..{port + tries - 1}")
        k += 1
        time.sleep(0.05)


def _recv_exact(sock, n):
    out = b""
    while len(out) < n:
        chunk = sock.recv(n - len(out))
        if not chunk:
            raise ConnectionError("rendezvous: connection closed early")
        out += chunk
    return out


class Comm:
    """RCCL communicator of the engine (zngamd_comm_*): one per process, bound to a context's GPU."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        L = _lib.load()
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        r = L.zngamd_comm_unique_id(buf)
        if r != _lib.OK:
            raise RuntimeError(f"zng_amd: RCCL is not usable here (zngamd_comm_unique_id -> {r})")
        return bytes(buf)

    def __init__(self, ctx, uid, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.L = ctx.L
        L = self.L
        L.zngamd_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.zngamd_comm_destroy.argtypes = [C.c_void_p]
        L.zngamd_comm_last_error.argtypes = [C.c_void_p]
        L.zngamd_comm_last_error.restype = C.c_char_p
        L.zngamd_comm_layout.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.zngamd_comm_allgather_stream.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64]
        L.zngamd_comm_wait.argtypes = [C.c_void_p]
        L.zngamd_comm_barrier.argtypes = [C.c_void_p]
        L.zngamd_comm_max_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.zngamd_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self.h = C.c_void_p()
        r = L.zngamd_comm_create(ctx.h, bytes(uid), rank, world, C.byref(self.h))
        if r != _lib.OK:
            raise RuntimeError(f"zng_amd: zngamd_comm_create -> {r}: {ctx.err()}")

    def _chk(self, r):
        if r != _lib.OK:
            raise RuntimeError(f"zng_amd comm: {r}: {self.L.zngamd_comm_last_error(self.h).decode('utf-8', 'replace')}")

    def layout(self, local_len, crc, ulen):
        """-> (offset of this rank's slice, total bytes, per-rank sizes, CRC-32 of the whole input, whole input length)"""
        sizes = (C.c_uint64 * self.world)()
        off, total, wl = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        wc = C.c_uint32(0)
        self._chk(self.L.zngamd_comm_layout(self.h, int(local_len), int(crc) & 0xFFFFFFFF, int(ulen), sizes, C.byref(off), C.byref(total),
                                            C.byref(wc), C.byref(wl)))
        return off.value, total.value, [int(x) for x in sizes], wc.value, wl.value

    def allgather_stream(self, d_local, sizes, d_stream, stream_cap):
        """Start the exchange of the slices (device pointers as integers); wait() blocks until it is done."""
        arr = (C.c_uint64 * self.world)(*sizes)
        self._chk(self.L.zngamd_comm_allgather_stream(self.h, C.c_void_p(d_local), arr, C.c_void_p(d_stream), int(stream_cap)))

    def count(self):
        """ranks of the communicator as RCCL reports them (ncclCommCount)"""
        n = C.c_int(0)
        self._chk(self.L.zngamd_comm_count(self.h, C.byref(n)))
        return n.value

    def wait(self):
        self._chk(self.L.zngamd_comm_wait(self.h))

    def barrier(self):
        self._chk(self.L.zngamd_comm_barrier(self.h))

    def max(self, value):
        v = C.c_double(float(value))
        self._chk(self.L.zngamd_comm_max_f64(self.h, C.byref(v)))
        return v.value

    def close(self):
        if self.h:
            self.L.zngamd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
