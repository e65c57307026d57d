"""CPU: the C-ABI library loads and exports every symbol include/zng_amd.h declares, the product fails
loudly without a GPU, and the host-side logic (header maths, framing, sharding + all-gather reassembly
with world_size 2 on gloo) is right.  No compute call is made here."""
import ctypes as C
import os
import re
import socket
import struct
import sys
import zlib

import pytest

from conftest import PKG_DIR, ROOT


def test_library_exports_every_declared_symbol():
    from zlib_ng_amd import _lib
    L = _lib.load()
    header = open(os.path.join(ROOT, "include", "zng_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(zngamd_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 28
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared
    assert L.zngamd_version().startswith(b"zng_amd")
    # the library must not depend on the oracle or on torch
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in needed and "torch" not in needed


def test_header_constants_match_binding():
    from zlib_ng_amd import _lib
    header = open(os.path.join(ROOT, "include", "zng_amd.h")).read()
    consts = dict(re.findall(r"#define\s+(ZNGAMD_[A-Z_]+)\s+\(?(-?\d+)u?\)?", header))
    assert int(consts["ZNGAMD_SLOT_STRIDE"]) == _lib.SLOT_STRIDE
    assert int(consts["ZNGAMD_UNIT_MAX"]) == _lib.UNIT_MAX
    assert int(consts["ZNGAMD_E_OVERFLOW"]) == _lib.E_OVERFLOW
    assert int(consts["ZNGAMD_BUF_ERROR"]) == _lib.BUF_ERROR


def test_scalar_entry_points_without_gpu():
    from zlib_ng_amd import _lib, zlib_ng
    L = _lib.load()
    a, b = b"hello ", b"world, this is crc32_combine"
    assert L.zngamd_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert zlib_ng.crc32_combine(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)
    assert [L.zngamd_level_ok(v) for v in (-2, -1, 0, 9, 10, 42)] == [0, 1, 1, 1, 0, 0]
    blocks = (_lib.Block * 3)(_lib.Block(0, 0, 0, 0, 0), _lib.Block(0, 131072, 0, 0, 0), _lib.Block(0, 131073, 0, 0, 0))
    assert L.zngamd_count_units(blocks, 3) == 1 + 1 + 2


def test_no_gpu_means_loud_failure():
    """On a box without a GPU the product raises; it never falls back to a CPU codec."""
    from zlib_ng_amd import _lib, zlib_ng
    if _lib.load().zngamd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no usable GPU"):
        zlib_ng.compress(b"abc")
    with pytest.raises(RuntimeError, match="no usable GPU"):
        zlib_ng.crc32(b"abc")
    for name in ("zlib_ng", "gzip_ng", "gzip_ng_threaded", "_lib", "shard"):
        src = open(os.path.join(PKG_DIR, "zlib_ng_amd", name + ".py")).read()
        assert "oracle" not in src.replace("no oracle", ""), name
        assert "import zlib\n" not in src and "from zlib " not in src, name


def test_framing_helpers():
    from zlib_ng_amd import shard, zlib_ng
    # zlib header bytes for the usual levels (RFC 1950): 78 01 / 78 5e / 78 9c / 78 da
    assert [zlib_ng._zlib_header(lv, 15).hex() for lv in (1, 3, 6, 9, -1)] == ["7801", "785e", "789c", "78da", "789c"]
    assert zlib_ng._zlib_header(6, 9)[0] == 0x18
    for lv in (1, 6, 9):
        co = zlib.compressobj(lv, zlib.DEFLATED, 31)
        assert zlib_ng._gzip_header(lv) == (co.compress(b"") + co.flush())[:10]
    h, t = shard.gzip_frame(0, 0x12345678, 2 ** 32 + 5, 9)
    assert h == bytes.fromhex("1f8b0800" "00000000" "ff02") and t == b"\x03\x00" + struct.pack("<II", 0x12345678, 5)
    assert shard.shard_range(10, 0, 3) == (0, 3) and shard.shard_range(10, 2, 3) == (6, 10)
    cover = [shard.shard_range(32768, r, 8) for r in range(8)]
    assert cover[0][0] == 0 and cover[-1][1] == 32768 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def _gloo_worker(rank, world, port, q, path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, PKG_DIR)
    from zlib_ng_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank "compresses" its contiguous block range (stand-in codec: the system zlib; the exchange logic
    # under test is independent of who produced the bytes) with the reference's dictionary chaining: the first block of
    # rank 1's range is primed with the tail of rank 0's input
    data = bytes((i * 7 + (i >> 9)) & 0xFF for i in range(40 * 1024))
    bs = 4096
    nb = len(data) // bs
    lo, hi = shard.shard_range(nb, rank, world)
    parts, crcs = [], []
    for b in range(lo, hi):
        zd = data[max(0, b * bs - 32768):b * bs]
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0, zd) if zd else zlib.compressobj(6, zlib.DEFLATED, -15)
        parts.append(co.compress(data[b * bs:(b + 1) * bs]) + co.flush(zlib.Z_SYNC_FLUSH))
        crcs.append((zlib.crc32(data[b * bs:(b + 1) * bs]), bs))
    mine = b"".join(parts)
    my_crc = shard.combine_crcs(crcs)

    def allgather(rec):                       # the transport under the layout exchange: gloo here, RCCL (Comm.layout) on GPUs
        out = [None] * world
        dist.all_gather_object(out, rec)
        return out
    off, total, sizes, crc, usize = shard.exchange_layout(len(mine), my_crc, (hi - lo) * bs, rank, allgather)
    ok = sizes[rank] == len(mine) and off == sum(sizes[:rank]) and total == sum(sizes) and crc == zlib.crc32(data) and usize == len(data)
    # (1) the whole stream on every rank: slices land at their offsets (what Comm.allgather_stream does with ncclSend/ncclRecv)
    stream = torch.zeros(total, dtype=torch.uint8)
    mx = max(sizes)
    padded = torch.zeros(mx, dtype=torch.uint8)
    padded[:len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
    got = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(got, padded)
    o = 0
    for r in range(world):
        stream[o:o + sizes[r]] = got[r][:sizes[r]]
        o += sizes[r]
    header, trailer = shard.gzip_frame(total, crc, usize, 6)
    import gzip
    ok = ok and gzip.decompress(header + bytes(stream.numpy()) + trailer) == data
    # (2) the same file without moving any payload between ranks: every rank writes its slice at its own offset
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
    os.pwrite(fd, mine, len(header) + off)
    if rank == 0:
        os.pwrite(fd, header, 0)
        os.pwrite(fd, trailer, len(header) + total)
    os.close(fd)
    dist.barrier()
    ok = ok and gzip.decompress(open(path, "rb").read()) == data
    # (3) the launcher-free hand-over of the communicator id: rank 0 serves 128 bytes over TCP
    uid = bytes(range(128)) if rank == 0 else None
    ok = ok and shard.rendezvous_bytes(rank, world, "127.0.0.1", port + 1, uid) == bytes(range(128))
    q.put((rank, ok, total))
    dist.destroy_process_group()


def test_two_rank_allgather_reassembly_gloo(tmp_path):
    import torch.multiprocessing as mp
    path = str(tmp_path / "written_by_two_ranks.gz")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q, path)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res) and res[0][2] == res[1][2]


def _gloo_worker8(rank, world, port, q, path):
    """world 8, uneven slices, ranks WITHOUT any block (more ranks than blocks in their range): the layout arithmetic and the
    write-at-own-offset assembly of shard.py, over gloo."""
    import torch.distributed as dist
    sys.path.insert(0, PKG_DIR)
    from zlib_ng_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng_bytes = bytes((i * 131 + (i >> 7) * 17) & 0xFF for i in range(5 * 8192 + 1234))      # 6 blocks (the last one short) over 8 ranks
    bs = 8192
    nb = -(-len(rng_bytes) // bs)
    lo, hi = shard.shard_range(nb, rank, world)
    parts, crcs = [], []
    for b in range(lo, hi):
        blk = rng_bytes[b * bs:(b + 1) * bs]
        zd = rng_bytes[max(0, b * bs - 32768):b * bs]
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0, zd) if zd else zlib.compressobj(6, zlib.DEFLATED, -15)
        parts.append(co.compress(blk) + co.flush(zlib.Z_SYNC_FLUSH))
        crcs.append((zlib.crc32(blk), len(blk)))
    mine = b"".join(parts)

    def allgather(rec):
        out = [None] * world
        dist.all_gather_object(out, rec)
        return out
    off, total, sizes, crc, usize = shard.exchange_layout(len(mine), shard.combine_crcs(crcs), sum(n for _, n in crcs), rank, allgather)
    ok = sizes[rank] == len(mine) and off == sum(sizes[:rank]) and total == sum(sizes) and crc == zlib.crc32(rng_bytes) and usize == len(rng_bytes)
    ok = ok and sizes.count(0) >= 2 and len(set(s for s in sizes if s)) > 1          # empty and uneven slices were exercised
    offs = (C.c_uint64 * world)()
    ok = ok and shard._lib.load().zngamd_comm_offsets((C.c_uint64 * world)(*sizes), world, offs) == total and list(offs) == [sum(sizes[:r]) for r in range(world)]
    header, trailer = shard.gzip_frame(total, crc, usize, 6)
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
    if mine:
        os.pwrite(fd, mine, len(header) + off)
    if rank == world - 1:                       # (any rank may write the frame: it follows from the layout alone)
        os.pwrite(fd, header, 0)
        os.pwrite(fd, trailer, len(header) + total)
    os.close(fd)
    dist.barrier()
    import gzip
    ok = ok and gzip.decompress(open(path, "rb").read()) == rng_bytes
    q.put((rank, ok, total))
    dist.destroy_process_group()


def test_eight_rank_layout_with_empty_and_uneven_slices_gloo(tmp_path):
    import torch.multiprocessing as mp
    path = str(tmp_path / "written_by_eight_ranks.gz")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker8, args=(r, 8, port, q, path)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res) and len({t for _, _, t in res}) == 1


def test_rendezvous_skips_a_busy_port():
    """shard.rendezvous_bytes: rank 0 takes the first free port of the range, the other ranks find it; a foreign listener on
    the first port (accepts, says nothing useful) does not confuse them."""
    import threading
    from zlib_ng_amd import shard
    squat = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    squat.bind(("127.0.0.1", 0))
    squat.listen(8)
    port = squat.getsockname()[1]
    stop = threading.Event()

    def foreign():
        squat.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = squat.accept()
                c.sendall(b"HTTP/1.0 400\r\n\r\n")
                c.close()
            except OSError:
                pass
    ft = threading.Thread(target=foreign, daemon=True)
    ft.start()
    payload = os.urandom(128)
    got = {}

    def run(rank):
        got[rank] = shard.rendezvous_bytes(rank, 3, "127.0.0.1", port, payload if rank == 0 else None, timeout=30.0)
    ts = [threading.Thread(target=run, args=(r,)) for r in (1, 2, 0)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(40)
    stop.set()
    ft.join(2)
    squat.close()
    assert got == {0: payload, 1: payload, 2: payload}


def test_result_object_grows_and_is_cut_in_place():
    """_lib._Out: the result of a call is an uninitialised bytes object held through a bare pointer; it is grown and finally
    cut to size in place (no zero fill, no final copy) -- what the streaming bindings build their output in."""
    import ctypes as C
    from zlib_ng_amd import _lib
    o = _lib._Out(10)
    C.memmove(o.addr(), b"0123456789", 10)
    o.resize(1 << 20)                                   # grows (the address may change, the bytes stay)
    C.memmove(o.addr().value + 10, b"ab", 2)
    r = o.take(12)
    assert r == b"0123456789ab" and type(r) is bytes and sys.getrefcount(r) == 2
    assert _lib._Out(16).take(0) == b""
    big = _lib._Out(64)
    C.memset(big.addr(), 0x41, 64)
    assert big.take(64) == b"A" * 64                    # filled completely: handed over as it is
    del o, big                                           # (objects that were never taken are released with their holder)


def test_slice_offsets_of_the_exchange():
    """zngamd_comm_offsets: where every rank's slice lies in the assembled stream (what zngamd_comm_allgather_stream places by and
    what a positional write uses) -- exclusive prefix sums in rank order, for 1 .. 9 ranks, with empty slices and sizes beyond 4 GiB."""
    import ctypes as C
    import random
    from zlib_ng_amd import _lib
    L = _lib.load()
    L.zngamd_comm_offsets.restype = C.c_uint64
    L.zngamd_comm_offsets.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
    rng = random.Random(5)
    for world in range(1, 10):
        for _ in range(20):
            sizes = [rng.choice([0, 1, 7, 1 << 20, (5 << 30) + 3, rng.randrange(1 << 33)]) for _ in range(world)]
            a, o = (C.c_uint64 * world)(*sizes), (C.c_uint64 * world)()
            total = L.zngamd_comm_offsets(a, world, o)
            assert total == sum(sizes)
            assert list(o) == [sum(sizes[:r]) for r in range(world)]
            assert L.zngamd_comm_offsets(a, world, None) == total


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher (WORLD_SIZE unset): the parent, which never touches the GPU, starts two FRESH
    child processes with distinct ranks; on this GPU-less box each one stops at "needs a GPU" with exit code 2 and the parent
    passes that on (the SCALE run of the driver may use exactly this command)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    started = [ln for ln in r.stderr.splitlines() if "started rank" in ln]
    assert len(started) == 2 and "rank 0 of 2" in started[0] and "rank 1 of 2" in started[1]
    pids = {ln.split("pid")[-1].strip() for ln in started}
    assert len(pids) == 2
    for rank in (0, 1):
        assert f"rank {rank} of 2 needs a GPU" in r.stderr
        assert any(f"rank {rank} (pid" in ln and "exited with 2" in ln for ln in r.stderr.splitlines())


def test_index_members_round_trip_and_are_skipped_by_any_gzip_reader():
    """The file format of the writer's segment index (host logic, no GPU): empty gzip members with a 'ZA' FEXTRA subfield and a
    locator behind ONE data member -- what _lib.index_members writes, _lib.parse_index_tail finds from the file's end, and the
    system gzip reads as nothing at all."""
    import gzip
    import io
    import struct
    import zlib
    import numpy as np
    from zlib_ng_amd import _lib
    rng = np.random.default_rng(7)
    nu = 2 * _lib.INDEX_PER_MEMBER + 13                      # three index members
    rec = np.zeros(nu, _lib._index_rec_dtype())
    rec["out_len"] = 131072
    rec["out_len"][-1] = 5000
    rec["in_len"] = rng.integers(100, 60000, nu)
    e = rng.integers(1, 2000, (nu, 65)).astype(np.uint16)
    nseg = (rec["out_len"].astype(np.int64) + 2047) >> 11
    e[np.arange(65)[None, :] > nseg[:, None]] = 0
    e[5] = 0                                                 # a unit of stored blocks: all zeros
    rec["e"] = e
    payload = bytes(int(rec["in_len"].astype(np.int64).sum()))     # stands for the units' deflate bytes (never decoded here)
    data_member = bytes.fromhex("1f8b0800" "00000000" "00ff") + payload + b"\x03\x00" + struct.pack("<II", 0, 0)
    tail = _lib.index_members([rec[:1000], rec[1000:]], len(data_member))
    plain = bytes.fromhex("1f8b0800" "00000000" "00ff" "0300" "00000000" "00000000")
    for blob in (data_member + tail, data_member + tail + plain, b"x" * 777 + data_member + tail + plain):
        start = len(blob) - len(data_member) - len(tail) - (len(plain) if blob.endswith(plain) else 0)
        fp = io.BytesIO(blob)
        fp.seek(3)
        got = _lib.parse_index_tail(fp, start, len(blob))
        assert fp.tell() == 3 and got is not None
        uin, uout, rows = got
        assert (uin == rec["in_len"]).all() and (uout == rec["out_len"]).all() and rows.shape == (nu, _lib.INDEX_STRIDE)
        want = np.cumsum(e.astype(np.uint32), axis=1, dtype=np.uint32)
        want[np.arange(65)[None, :] > nseg[:, None]] = 0
        want[5] = 0
        assert (rows[:, :65] == want).all()
    # every index member and the locator are complete, empty gzip members: the system's reader sees no data in them
    assert gzip.decompress(tail) == b"" and gzip.decompress(tail + plain) == b""
    d = zlib.decompressobj(31)
    assert d.decompress(tail[:tail.index(b"\x1f\x8b", 4)]) == b"" and d.eof
    # any doubt and there is no index: a damaged locator, sizes that do not add up, a truncated file
    blob = data_member + tail + plain
    for at, what in ((len(blob) - 20 - 30, "locator"), (len(data_member) + 17, "record header")):
        bad = bytearray(blob); bad[at] ^= 0x40
        assert _lib.parse_index_tail(io.BytesIO(bytes(bad)), 0, len(bad)) is None, what
    assert _lib.parse_index_tail(io.BytesIO(blob[:-25]), 0, len(blob) - 25) is None
    assert _lib.parse_index_tail(io.BytesIO(b"y" + blob), 0, len(blob) + 1) is None      # the data member does not start where the caller says
