#!/usr/bin/env python3
"""bench.py -- MB/s compress+decompress, 128 KiB blocks, level 6 (BASELINE.json metric).

One "step" = one pass of the hot path over this rank's shard of synthetic text already resident in HBM:
  (1) compress: every 128 KiB block, primed with the previous 32 KiB of input as dictionary
      (reference semantics: gzip_ng_threaded.py:299-322 + zlib_ngmodule.c:1696-1782), through the five
      deflate kernels, gathered into one contiguous raw-deflate slice.  With N > 1 ranks every rank owns a contiguous
      block range of ONE stream (its first block is primed with the 32 KiB of input in front of the range: the previous
      rank's tail) and the slices are exchanged over RCCL through the engine's own entry points (zngamd_comm_*: layout
      all-gather, then exact-size grouped ncclSend / ncclRecv; issued on a stream of its own: it overlaps leg 2 and, through a
      second slice buffer, the next step's leg 1; the last step's exchange is waited for inside the timed region), which
      leaves the whole member stream on every rank (BENCH_EXCHANGE=layout: sizes only, for ranks that write their slice
      by offset).  --scaling weak (default): every rank brings --size-mib of its own; strong: --size-mib is the whole job;
  (2) decompress: two-pass inflate (member scan, then one wavefront per member) of a pre-built stream
      of independent indexed gzip members of the same text (BASELINE.json configs[2]).
value = uncompressed bytes of all ranks / (max over ranks of the step time): the rate at which data goes
through compress and then decompress.  Inputs are in HBM when the timed region starts.

    python bench.py                      # 1 GPU, 4 GiB shard
    python bench.py --gpus N             # N ranks started by bench.py itself (fresh child processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))

BLOCK = 131072
HBM_PEAK_GBS = 8000.0


def host_cores():
    """CPUs this process may really use: the scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except Exception:
        pass
    return n


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this process has not touched
    the GPU or imported torch yet, and never does), one per GPU, with the environment a launcher would give them; relay rank 0's
    JSON line; exit non-zero if any rank does.  Nothing here replaces a running program (no exec)."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free port for the ranks' rendezvous (the RCCL id travels on port + 1 ...)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
        print(f"bench.py: started rank {r} of {n} as pid {procs[-1].pid}", file=sys.stderr)
    rc, line, failed_at, signalled = 0, b"", None, 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if r == 0:
                line = procs[0].stdout.read()
            if code != 0:
                print(f"bench.py: rank {r} (pid {procs[r].pid}) exited with {code}", file=sys.stderr)
                rc = rc or code
                failed_at = failed_at or time.monotonic()
        if alive:
            time.sleep(0.05)
            # a rank has failed: the others either fail by themselves within moments (same cause) or would wait for it in a
            # collective for ever -- after a grace period end exactly those PIDs
            if failed_at and signalled == 0 and time.monotonic() - failed_at > 5.0:
                for q in sorted(alive):
                    procs[q].terminate()
                signalled = 1
            elif failed_at and signalled == 1 and time.monotonic() - failed_at > 15.0:
                for q in sorted(alive):
                    procs[q].kill()
                signalled = 2
    if rc == 0 and line.strip():
        sys.stdout.write(line.decode().strip().splitlines()[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    sys.exit(rc if rc > 0 else (1 if rc else 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-mib", type=int, default=4096, help="uncompressed MiB per GPU (weak scaling)")
    ap.add_argument("--unique-mib", type=int, default=64, help="MiB of distinct text, tiled to --size-mib")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --size-mib per GPU (the driver's contract); strong: --size-mib is the whole stream, cut over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-foreign", action="store_true", help="skip the foreign-member inflate leg (outside the timed region)")
    ap.add_argument("--cpu-sample-mib", type=int, default=0, help="0 = sized for ~10-30 s")
    ap.add_argument("--no-api", action="store_true", help="skip the drop-in API leg (host buffers over PCIe; outside the timed region)")
    ap.add_argument("--no-heldout", action="store_true", help="skip the held-out ratio leg (small deflate calls outside the timed region; counter passes average per launch)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                  # does not return
    # stdout carries exactly ONE line, the JSON result: file descriptor 1 is pointed at stderr for the run (RCCL prints a version
    # banner on the first communicator, libraries print what they like) and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    from zlib_ng_amd import _lib, corpus, devmem, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch through torch.distributed.run",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if _lib.load().zngamd_device_count() <= local:
        print(f"bench.py: rank {rank} of {world} needs a GPU (the engine has no CPU path)", file=sys.stderr)
        sys.exit(2)
    # device memory, copies, fills and compares all go through the engine's C ABI (zlib_ng_amd.devmem): no tensor library
    ctx = _lib.Context(device=local)
    L, h = ctx.L, ctx.h
    dempty = lambda nbytes: devmem.empty(ctx, nbytes)
    i32 = lambda buf, n=None: buf.cpu(np.int32) if n is None else buf[:4 * n].cpu(np.int32)
    # BENCH_FORCE_EXCHANGE=1 runs the exchange leg even with one rank (rehearsal of the N > 1 path on one GPU).
    # Default exchange (north_star): the member stream is reassembled on every rank over RCCL (shard.Comm = zngamd_comm_*, no
    # torch in the data path), asynchronously so that it overlaps the inflate leg.  BENCH_EXCHANGE=layout exchanges only the
    # layout of the stream (three integers per rank): what ranks that write their slices by offset need.
    exchange = world > 1 or os.environ.get("BENCH_FORCE_EXCHANGE") == "1"
    exchange_stream = exchange and os.environ.get("BENCH_EXCHANGE", "stream") != "layout"
    comm = None
    if exchange:
        # the 128-byte RCCL id travels from rank 0 over a TCP socket next to the launcher's port (no torch.distributed)
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29533")) + 1 + int(os.environ.get("BENCH_PORT_OFFSET", "0"))
        uid = shard.rendezvous_bytes(rank, world, addr, port, shard.Comm.unique_id() if rank == 0 else None)
        comm = shard.Comm(ctx, uid, rank, world)

    # ---- synthetic stream, this rank's block range resident in HBM ---------------------------------
    # All ranks draw the SAME seeded text, tiled: the job is one stream of (weak: world x, strong: 1 x) --size-mib, rank r
    # owns the contiguous block range r of it, and what lies in front of that range is known on every rank.
    total_size = (args.size_mib << 20) * (world if args.scaling == "weak" else 1)
    total_blocks = total_size // BLOCK
    blo, bhi = shard.shard_range(total_blocks, rank, world)
    nblocks = bhi - blo
    size = nblocks * BLOCK
    uniq = min(args.unique_mib << 20, args.size_mib << 20)
    uniq -= uniq % BLOCK
    host = corpus.text(uniq, seed=1)
    base = devmem.from_host(ctx, host)
    HALO = 32768
    start = blo * BLOCK                                     # byte offset of my range in the whole stream
    # d_buf = [32 KiB halo: the bytes in front of my range][my range][64 zero bytes]; the stream is the tile repeated
    d_buf = dempty(HALO + size + 64)
    idx0 = (start - HALO) % uniq
    pos = 0
    while pos < HALO + size:
        o = (idx0 + pos) % uniq
        k = min(uniq - o, HALO + size - pos)
        d_buf[pos:pos + k] = base[o:o + k]
        pos += k
    d_buf[HALO + size:] = 0
    d_in = d_buf[HALO:]
    ctx.sync()

    BLOCK_FLAGS = 0 if os.environ.get("BENCH_EXACT_FRAMING") == "1" else _lib.FLAG_SEG2K
    blocks = (_lib.Block * nblocks)()
    for b in range(nblocks):        # offsets inside d_buf; only the very first block of the whole stream has no dictionary
        # (flags: what the default writer of r06 asks for -- gzip_ng_threaded.py in this package: 2 KiB segments in blocks of any
        # size, which for these full-size blocks is what they have anyway; BENCH_EXACT_FRAMING=1: no flag, and no index leg)
        blocks[b] = _lib.Block(HALO + b * BLOCK, BLOCK, 32768 if (b or blo) else 0, BLOCK_FLAGS, 0)
    n_units = L.zngamd_count_units(blocks, nblocks)
    assert n_units == nblocks
    d_ulen = dempty(4 * n_units)
    d_ucrc = dempty(4 * n_units)
    # the compressed slice, twice where the slices are exchanged: the exchange of step i reads one while step i + 1 compresses into the
    # other (the transfer -- about 10 GB arriving per rank and step at N = 8 -- then overlaps the inflate leg AND the next step's
    # compress instead of the inflate leg alone)
    d_comps = [dempty(size // 2 + (64 << 20)) for _ in range(2 if exchange_stream else 1)]   # text compresses ~3x
    d_comp = d_comps[0]
    d_stream = dempty(world * (size // 2 + (8 << 20)) + (64 << 20)) if exchange_stream else None
    d_out = dempty(size + 64)
    ptr = lambda t: t.vp()

    def chk(r, what):
        if r != 0:
            raise RuntimeError(f"{what} failed: {r} {ctx.err()}")

    # pre-built multi-member stream for the inflate leg (outside the timed region)
    d_members_stream = dempty(size // 2 + nblocks * 400 + (64 << 20))
    ms_len, ms_n = C.c_uint64(0), C.c_uint32(0)
    chk(L.zngamd_gzip_members_dev(h, ptr(d_in), size, BLOCK, args.level, ptr(d_members_stream),
                                  d_members_stream.numel() - 64, C.byref(ms_len), C.byref(ms_n)), "gzip_members_dev")
    d_members_stream[ms_len.value:ms_len.value + 64] = 0
    d_mtab = dempty(nblocks * C.sizeof(_lib.Member))
    d_mstat = dempty(4 * nblocks)
    ctx.sync()

    comp_total = C.c_uint64(0)
    gathered = {}

    step_no = [0]
    pending = [False]

    def step():
        nonlocal d_comp
        # (1) compress straight into the one contiguous stream: every unit's size is known before it is packed, so the packer
        #     writes at the unit's final byte offset (no slots, no gather) (+ exchange of the slices)
        d_comp = d_comps[step_no[0] % len(d_comps)]
        step_no[0] += 1
        chk(L.zngamd_deflate_blocks_packed_dev(h, ptr(d_buf), HALO + size, blocks, nblocks, args.level, ptr(d_comp), d_comp.numel() - 64,
                                               ptr(d_ulen), ptr(d_ucrc), None, C.byref(comp_total)), "deflate_blocks_packed_dev")
        if exchange_stream and pending[0]:
            comm.wait()                   # the exchange of the step in front (it ran beside that step's inflate and this step's compress)
            pending[0] = False
        if exchange:
            # CRC-32 of my range from the per-block values (the writer thread's fold, gzip_ng_threaded.py:394), then the layout of
            # the one stream: 24 bytes per rank over RCCL
            crc = C.c_uint32(0)
            chk(L.zngamd_crc32_fold_dev(h, ptr(d_ucrc), n_units, BLOCK, BLOCK, C.byref(crc)), "crc32_fold_dev")
            gathered["layout"] = comm.layout(comp_total.value, crc.value, size)
            if exchange_stream:  # the slices travel (grouped ncclSend / ncclRecv on the communicator's stream) while this rank inflates
                comm.allgather_stream(d_comp.data_ptr(), gathered["layout"][2], d_stream.data_ptr(), d_stream.numel() - 64)
                pending[0] = True
        # (2) two-pass inflate of the pre-built member stream
        nm, tot = C.c_uint32(0), C.c_uint64(0)
        chk(L.zngamd_gzip_scan_dev(h, ptr(d_members_stream), ms_len.value, ptr(d_mtab), nblocks, C.byref(nm),
                                   C.byref(tot)), "gzip_scan_dev")
        chk(L.zngamd_gzip_inflate_members_dev(h, ptr(d_members_stream), ms_len.value, ptr(d_mtab), nm.value,
                                              ptr(d_out), size, ptr(d_mstat)), "gzip_inflate_members_dev")

    def barrier():
        ctx.sync()
        if exchange_stream and pending[0]:
            comm.wait()                   # the last step's exchange ends inside the timed region
            pending[0] = False
        if exchange:
            comm.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        step()
    ctx.profiling(True)
    ctx.kernel_times(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kt = ctx.kernel_times(reset=True)
    ctx.profiling(False)
    if exchange:
        dt = comm.max(dt)
    # the segment index of the compress leg's last call (what the writer puts behind its data member): taken now, before any
    # other deflate call of this context replaces it
    d_index = ctx.deflate_index(n_units) if (rank == 0 and blo == 0 and BLOCK_FLAGS and not args.no_foreign) else None

    # ---- correctness of what was timed -------------------------------------------------------------
    # Every collective of the verification runs BEFORE the first assert, and the verdicts are exchanged afterwards: a rank whose
    # check fails must not leave the others waiting for it in RCCL for ever (it would look like a hang, not like a failure).
    kblk = min(nblocks, 2048)
    pre = comm.layout(int(i32(d_ulen, kblk).astype(np.int64).sum()), 0, 0)[2] if exchange_stream else None   # compressed bytes of every rank's first kblk blocks
    rccl_ranks = comm.count() if comm is not None else None
    comp_bytes = int(i32(d_ulen).astype(np.int64).sum())

    def verify():
        assert not i32(d_mstat).any(), "inflate reported member errors"
        assert d_out[:size].equal(d_in[:size]), "inflate output differs from the input"
        assert comp_bytes == comp_total.value
        # The WHOLE dict-chained compressed stream -- with N > 1 the stream assembled from the slices of all ranks, on rank 0 -- is
        # closed with an empty final block, inflated on the device by the chunk-parallel decoder (sync-flush points) and compared
        # with the input; its CRC-32 must be the one the layout exchange folded; a prefix also goes through the system zlib.
        if exchange:
            off, total, sizes, whole_crc, whole_len = gathered["layout"]
            assert sizes[rank] == comp_bytes and off == sum(sizes[:rank]) and total == sum(sizes) and whole_len == total_size
            # the trailer CRC-32 folded from the ranks against the CRC-32 of the whole input, folded on the host from the tile's
            if total_size % uniq == 0:
                tile_crc, want = zlib.crc32(host), 0
                for _ in range(total_size // uniq):
                    want = ctx.crc32_combine(want, tile_crc, uniq)
                assert want == whole_crc, "trailer CRC-32 folded from the ranks differs from the CRC-32 of the whole input"
        if exchange_stream:
            # (a) my slice lies at its offset in my copy of the assembled stream
            assert d_stream[off:off + comp_bytes].equal(d_comp[:comp_bytes]), "my slice is not at its offset in the assembled stream"
            # (b) every rank decodes the head of a slice ANOTHER rank compressed (rank r: slice r + 1), taken from its own copy of
            # the assembled stream: up to 2 048 blocks behind a stored block that holds the 32 KiB of input in front of them (the
            # dictionary that slice was primed with on the other GPU), compared with the input they must decode to
            q = (rank + 1) % world
            qlo = shard.shard_range(total_blocks, q, world)[0]
            qoff = sum(sizes[:q])

            def tile_bytes(first, count):                       # bytes [first, first + count) of the whole stream (negative: halo of block 0)
                outb, pos = dempty(count), 0
                while pos < count:
                    o = (first + pos) % uniq
                    k = min(uniq - o, count - pos)
                    outb[pos:pos + k] = base[o:o + k]
                    pos += k
                return outb
            d_v = dempty(5 + HALO + pre[q] + 66)
            d_v[:5] = bytes([0, 0x00, 0x80, 0xFF, 0x7F])        # stored block, not final, 32 768 bytes
            d_v[5:5 + HALO] = tile_bytes(qlo * BLOCK - HALO, HALO)
            d_v[5 + HALO:5 + HALO + pre[q]] = d_stream[qoff:qoff + pre[q]]
            d_v[5 + HALO + pre[q]:] = 0
            d_v[5 + HALO + pre[q]] = 3                           # empty final block
            d_vo = dempty(HALO + kblk * BLOCK + 64)
            vlen, vused = C.c_uint64(0), C.c_uint64(0)
            rc = L.zngamd_inflate_raw_dev(h, ptr(d_v), 5 + HALO + pre[q] + 2, ptr(d_vo), HALO + kblk * BLOCK, C.byref(vlen), C.byref(vused))
            assert rc == _lib.STREAM_END and vlen.value == HALO + kblk * BLOCK, (rc, vlen.value, vused.value, ctx.err())
            assert d_vo[HALO:HALO + kblk * BLOCK].equal(tile_bytes(qlo * BLOCK, kblk * BLOCK)), \
                f"rank {rank}: the head of slice {q} in the assembled stream does not inflate to its input"
            del d_v, d_vo
            # (c) jobs of at most 4 GiB: rank 0 inflates the WHOLE assembled stream on the device and compares it with the input
            if rank == 0 and total_size <= (4 << 30) and total_size % uniq == 0:
                d_stream[total:total + 66] = 0
                d_stream[total] = 3
                d_big = dempty(total_size + 64) if world > 1 else d_out
                rc = L.zngamd_inflate_raw_dev(h, ptr(d_stream), total + 2, ptr(d_big), total_size, C.byref(vlen), C.byref(vused))
                assert rc == _lib.STREAM_END and vlen.value == total_size and vused.value == total + 2, (rc, vlen.value, vused.value, ctx.err())
                assert all(d_big[t * uniq:(t + 1) * uniq].equal(base) for t in range(total_size // uniq)), "the assembled stream does not inflate to the input"
                c = C.c_uint32(0)
                chk(L.zngamd_crc32_dev(h, 0, ptr(d_big), total_size, C.byref(c)), "crc32_dev")
                assert c.value == whole_crc, "CRC-32 of the inflated stream differs from the trailer value"
                del d_big
        if blo == 0:
            d_comp[comp_bytes:comp_bytes + 66] = 0
            d_comp[comp_bytes] = 3
            vlen, vused = C.c_uint64(0), C.c_uint64(0)
            d_out.zero_()
            rc = L.zngamd_inflate_raw_dev(h, ptr(d_comp), comp_bytes + 2, ptr(d_out), size, C.byref(vlen), C.byref(vused))
            assert rc == _lib.STREAM_END and vlen.value == size and vused.value == comp_bytes + 2, (rc, vlen.value, vused.value, ctx.err())
            assert d_out[:size].equal(d_in[:size]), "the compressed stream does not inflate to the input"
            nchk = min(nblocks, 64)
            ul = i32(d_ulen, nchk).astype(np.int64)
            pref = bytes(d_comp[:int(ul.sum())].cpu())
            assert zlib.decompressobj(-15).decompress(pref) == bytes(d_in[:nchk * BLOCK].cpu()), \
                "compressed stream does not inflate to the input (system zlib)"

    verr = None
    try:
        verify()
    except BaseException as e:          # noqa: BLE001 -- reported below, after the ranks have agreed on the verdict
        verr = e
    if exchange:
        bad = comm.max(1.0 if verr is not None else 0.0)
        if bad:
            if verr is not None:
                import traceback
                traceback.print_exception(type(verr), verr, verr.__traceback__, file=sys.stderr)
            print(f'bench.py: rank {rank}: verification failed on ' + ('this rank' if verr is not None else 'another rank'), file=sys.stderr)
            comm.close()
            sys.exit(1)
    elif verr is not None:
        raise verr

    # ---- foreign members: the same text as 128 KiB gzip members written by the SYSTEM zlib (no index, ordinary dynamic
    # headers), decoded one wavefront per member; outside the timed region, reported beside the headline inflate leg ----
    foreign = None
    reps = size // uniq
    if rank == 0 and not args.no_foreign and size % uniq == 0:
        import struct
        from concurrent.futures import ThreadPoolExecutor
        hv = memoryview(host)

        def zmember(b):
            co = zlib.compressobj(args.level, zlib.DEFLATED, -15)
            raw = co.compress(hv[b * BLOCK:(b + 1) * BLOCK]) + co.flush()
            return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + raw + struct.pack("<II", zlib.crc32(hv[b * BLOCK:(b + 1) * BLOCK]), BLOCK)
        with ThreadPoolExecutor(host_cores()) as ex:
            mem = list(ex.map(zmember, range(uniq // BLOCK)))
        tile = b"".join(mem)
        tl = len(tile)
        d_tile = devmem.from_host(ctx, tile)
        d_for = dempty(reps * tl + 64)
        for t_ in range(reps):
            d_for[t_ * tl:(t_ + 1) * tl] = d_tile
        d_for[reps * tl:] = 0
        ft = (_lib.Member * nblocks)()
        offs = np.concatenate([[0], np.cumsum([len(x) for x in mem])])
        per = uniq // BLOCK
        for b in range(nblocks):
            o = (b // per) * tl + int(offs[b % per])
            ln = len(mem[b % per])
            ft[b] = _lib.Member(o + 10, ln - 18, b * BLOCK, BLOCK, 0, 0, 0)
        d_ft = devmem.from_host(ctx, bytes(ft))
        d_out.zero_()
        f_ms = []
        for it in range(2):
            ctx.profiling(True); ctx.kernel_times(reset=True)
            chk(L.zngamd_gzip_inflate_plain_members_dev(h, ptr(d_for), reps * tl, ptr(d_ft), nblocks, ptr(d_out), size, ptr(d_mstat)),
                "gzip_inflate_plain_members_dev")
            f_ms.append(ctx.kernel_times(reset=True)["inflate"][0])
            ctx.profiling(False)
        assert not i32(d_mstat).any(), "foreign members: errors"
        assert d_out[:size].equal(d_in[:size]), "foreign members: output differs"
        fbytes = reps * tl + size
        foreign = {"bound": "hbm", "kernel": "za_k_inflate_serial_members", "members": nblocks,
                   "writer": "system zlib " + zlib.ZLIB_RUNTIME_VERSION + f" level {args.level}, plain gzip members of 128 KiB",
                   "ms": round(min(f_ms), 3), "decompress_MBps": round(size / (min(f_ms) * 1e-3) / 1e6, 1),
                   "achieved": round(fbytes / (min(f_ms) * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(fbytes / (min(f_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "alg_bytes_per_launch": int(fbytes)}
        del d_for, d_tile, d_ft

    # ---- what the drop-in writer writes, decoded: the compress leg's OWN stream (one raw deflate stream of dict-chained, sync-flushed
    # 128 KiB blocks: gzip_ng_threaded.py:299-338 -- what gzip_ng_threaded.open(..., "rb") gets for a file of this package's default
    # writer), (a) with the writer's segment index, as the reader decodes a file that carries it (zngamd_inflate_units_indexed_dev:
    # the units side by side, a lane per 2 KiB segment, markers for what reaches in front of a unit), (b) without it -- a pipe, a file
    # of the reference's writer, exact_framing=True: sync points found by a scan, self-synchronising sweeps inside the blocks
    # (zngamd_inflate_raw_dev); outside the timed region ----
    chained = chained_ix = None
    if rank == 0 and blo == 0 and not args.no_foreign:
        d_comp[comp_bytes:comp_bytes + 66] = 0
        d_comp[comp_bytes] = 3                                   # empty final block
        if d_index is not None:
            uin, uout = i32(d_ulen).astype(np.uint32), np.full(n_units, BLOCK, np.uint32)
            x_wall, x_kern = [], []
            for it in range(3):
                d_out.zero_()
                ctx.sync()
                ctx.profiling(True); ctx.kernel_times(reset=True)
                t = time.perf_counter()
                rc, xl = ctx.inflate_units_indexed_dev(d_comp.ptr, comp_bytes + 2, uin, uout, d_index.ptr, d_out.ptr, size)
                x_wall.append((time.perf_counter() - t) * 1e3)
                x_kern.append(sum(v[0] for v in ctx.kernel_times(reset=True).values()))
                ctx.profiling(False)
                assert rc == _lib.STREAM_END and xl == size, (rc, xl, ctx.err())
            assert d_out[:size].equal(d_in[:size]), "indexed chained stream: output differs"
            xb = comp_bytes + size
            xms = min(x_wall)
            chained_ix = {"bound": "hbm", "kernel": "za_k_inflate_units_marked + za_k_chunk_compose / _chain / _resolve",
                          "stream": f"the compress leg's own output: ONE raw deflate stream, {nblocks} dict-chained blocks of 128 KiB ending in sync flushes; "
                                    f"the writer's segment index beside it (138 bytes per block in a file's trailing members)",
                          "ms": round(xms, 3), "kernel_ms": round(min(x_kern), 3),
                          "decompress_MBps": round(size / (xms * 1e-3) / 1e6, 1),
                          "achieved": round(xb / (xms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(xb / (xms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "alg_bytes_per_launch": int(xb),
                          "note": "ms = the whole device-resident call (host-side tables and the result read-back included), best of 3"}
        vlen, vused = C.c_uint64(0), C.c_uint64(0)
        c_wall, c_kern = [], []
        for it in range(3):
            d_out.zero_()
            ctx.sync()
            ctx.profiling(True); ctx.kernel_times(reset=True)
            t = time.perf_counter()
            rc = L.zngamd_inflate_raw_dev(h, ptr(d_comp), comp_bytes + 2, ptr(d_out), size, C.byref(vlen), C.byref(vused))
            c_wall.append((time.perf_counter() - t) * 1e3)
            c_kern.append(sum(v[0] for v in ctx.kernel_times(reset=True).values()))
            ctx.profiling(False)
            assert rc == _lib.STREAM_END and vlen.value == size and vused.value == comp_bytes + 2, (rc, vlen.value, vused.value, ctx.err())
        assert d_out[:size].equal(d_in[:size]), "chained stream: output differs"
        cbytes = comp_bytes + size
        cms = min(c_wall)
        chained = {"bound": "hbm", "kernel": "za_k_chunk_decode + za_k_chunk_compose / _chain / _resolve (+ za_k_scan_sync)",
                   "stream": f"the same stream WITHOUT its index (what a pipe, or a file of the reference's writer, offers): {nblocks} dict-chained blocks of 128 KiB ending in sync flushes",
                   "ms": round(cms, 3), "kernel_ms": round(min(c_kern), 3),
                   "decompress_MBps": round(size / (cms * 1e-3) / 1e6, 1),
                   "achieved": round(cbytes / (cms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(cbytes / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "alg_bytes_per_launch": int(cbytes),
                   "note": "ms = the whole device-resident call (its host-side planning between the kernels included), best of 3"}

    # ---- BGZF: standard 'BC' members (at most 65 280 bytes of input each, FEXTRA subfield BC = member size - 1) written by the
    # system zlib, one wavefront per member; outside the timed region ----
    bgzf = None
    if rank == 0 and not args.no_foreign and size % uniq == 0:
        import struct
        from concurrent.futures import ThreadPoolExecutor
        hv = memoryview(host)
        BG = 65280
        nbg = (uniq + BG - 1) // BG

        def bmember(b):
            piece = hv[b * BG:min(uniq, (b + 1) * BG)]
            co = zlib.compressobj(args.level, zlib.DEFLATED, -15)
            raw = co.compress(piece) + co.flush()
            bsize = 18 + len(raw) + 8 - 1
            return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + raw +
                    struct.pack("<II", zlib.crc32(piece), len(piece)))
        with ThreadPoolExecutor(host_cores()) as ex:
            mem = list(ex.map(bmember, range(nbg)))
        tile = b"".join(mem)
        tl = len(tile)
        d_tile = devmem.from_host(ctx, tile)
        d_bg = dempty(reps * tl + 64)
        for t_ in range(reps):
            d_bg[t_ * tl:(t_ + 1) * tl] = d_tile
        d_bg[reps * tl:] = 0
        nmem = reps * nbg
        bt = (_lib.Member * nmem)()
        offs = np.concatenate([[0], np.cumsum([len(x) for x in mem])])
        for m in range(nmem):
            r_, b = divmod(m, nbg)
            ilen = min(uniq, (b + 1) * BG) - b * BG
            bt[m] = _lib.Member(r_ * tl + int(offs[b]) + 18, len(mem[b]) - 26, r_ * uniq + b * BG, ilen, 0, 0, 0)
        d_bt = devmem.from_host(ctx, bytes(bt))
        d_bstat = dempty(4 * nmem)
        d_out.zero_()
        b_ms = []
        for it in range(2):
            ctx.profiling(True); ctx.kernel_times(reset=True)
            chk(L.zngamd_gzip_inflate_plain_members_dev(h, ptr(d_bg), reps * tl, ptr(d_bt), nmem, ptr(d_out), size, ptr(d_bstat)),
                "gzip_inflate_plain_members_dev (BGZF)")
            b_ms.append(ctx.kernel_times(reset=True)["inflate"][0])
            ctx.profiling(False)
        assert not i32(d_bstat).any(), "BGZF members: errors"
        assert d_out[:size].equal(d_in[:size]), "BGZF members: output differs"
        bbytes = reps * tl + size
        bgzf = {"bound": "hbm", "kernel": "za_k_inflate_serial_members", "members": nmem,
                "writer": "system zlib " + zlib.ZLIB_RUNTIME_VERSION + f" level {args.level}, BGZF members ('BC' subfield) of 65 280 bytes",
                "ms": round(min(b_ms), 3), "decompress_MBps": round(size / (min(b_ms) * 1e-3) / 1e6, 1),
                "achieved": round(bbytes / (min(b_ms) * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(bbytes / (min(b_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "alg_bytes_per_launch": int(bbytes)}
        del d_bg, d_tile, d_bt, d_bstat

    # ---- per-leg numbers -----------------------------------------------------------------------------
    steps = args.steps
    deflate_ms = sum(kt[k][0] for k in ("chains", "search", "optparse", "parse", "plan", "pack", "gather")) / steps
    inflate_ms = (kt["scan"][0] + kt["inflate"][0]) / steps
    # The roofline of the step's dominant LEG: deflate = the kernels chains + search + parse + plan + pack + gather
    # together move N_in + C_out algorithmic bytes per unit -- no single one of them can be credited with those bytes -- so the
    # leg is priced as one: algorithmic bytes of one launch set (the whole shard) / the sum of its kernels' average launch times.
    # (`dominant_kernel` names the longest single kernel, with the bytes its own role makes it move and its instruction issue.)
    legs = {"deflate": deflate_ms, "inflate": inflate_ms}
    dom_leg = max(legs, key=legs.get)
    dom = max(("chains", "search", "optparse", "parse", "plan", "pack", "inflate"), key=lambda k: kt[k][0])
    launches = max(1, kt[dom][1])
    avg_ms = kt[dom][0] / launches
    launches_per_step = launches / steps
    if dom_leg == "deflate":
        leg_kernels = ("chains", "search", "optparse", "parse", "plan", "pack", "gather")
        leg_bytes = size + comp_bytes
        leg_name = "deflate pipeline: za_k_chains x3 (tables A, B, C) + za_k_search + za_k_optparse + za_k_parse + za_k_plan + za_k_offsets + za_k_pack (packed: no gather)"
    else:
        leg_kernels = ("scan", "inflate")
        leg_bytes = size + ms_len.value
        leg_name = "inflate: za_k_scan_members + za_k_inflate_members"
    leg_ms = sum(kt[k][0] for k in leg_kernels) / steps
    leg_sets = max(1.0, kt[leg_kernels[0] if dom_leg == "deflate" else "inflate"][1] / steps)   # launch sets per step (1: every kernel takes the whole shard)
    achieved = leg_bytes / (leg_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": leg_name,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "avg_launch_ms": round(leg_ms / leg_sets, 4), "alg_bytes_per_launch": int(leg_bytes / leg_sets),
                "units_per_launch": int(nblocks / leg_sets)}
    # bytes the dominant kernel's own role makes it move per input byte N (by design, not measured): chains read N + dictionary,
    # write 2 B links; search reads links and input, writes 4 B entries; parse reads the entries, writes tokens; ...
    own = {"chains": 3 * (1.0 + 2.0), "search": 2.0 + 2.0 + 2.0 + 1.0 + 4.0, "optparse": 4.0 + 4.0 + 4.0, "parse": 4.0 + 0.8, "pack": 1.6 + comp_bytes / size, "plan": 0.02,
           "inflate": (ms_len.value + size) / size}
    dom_bytes = own[dom] * size / launches_per_step
    # the longest single kernel: NOT a roofline fraction of the path (its bytes are intermediates) -- the bytes its own role makes it
    # move, and, since it is bound by instruction issue, its vector instructions against the chip's issue rate (`valu`, below)
    roofline_dom = {"kernel": "za_k_" + ("inflate_members" if dom == "inflate" else dom),
                    "avg_launch_ms": round(avg_ms, 4), "own_bytes_per_launch": int(dom_bytes),
                    "own_bytes_GBps": round(dom_bytes / (avg_ms * 1e-3) / 1e9, 2),
                    "note": "bytes this kernel's own role makes it read and write (links, entries, tokens included), by design; no fraction of a roofline"}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic_step, traffic_src, pmc_per_unit = {}, None, {}
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))                      # HBM bytes per unit from the PMC passes (profiles/run_pmc.sh)
            pmc_per_unit = pj
            traffic_src = "PMC passes of build '%s' taken at %s units per launch (profiles/pmc_traffic.json), scaled to %d units" % (
                pj.get("build", "?"), pj.get("units_per_launch", "?"), nblocks)
            for k in ("chains", "search", "optparse", "parse", "plan", "pack", "gather", "scan_members", "inflate_members"):
                if pj.get("za_k_" + k):
                    # (a timer class may hold a second kernel: za_k_dpstats ran under "optparse" until it moved into the search)
                    traffic_step[k] = int((pj["za_k_" + k] + (pj.get("za_k_dpstats", 0) if k == "optparse" else 0)) * nblocks)
                elif k == "gather":    # packed deflate: what is left under this timer is za_k_offsets (4 B read + 8 B written per unit: below the counters' floor)
                    traffic_step[k] = int(pj.get("za_k_offsets", 12) * nblocks)
            names = {"scan": "scan_members", "inflate": "inflate_members"}
            tr = [traffic_step.get(names.get(k, k)) for k in leg_kernels]
            if all(t is not None for t in tr if True) and any(tr):
                roofline["traffic"] = int(sum(t or 0 for t in tr) / leg_sets)
                roofline["traffic_source"] = traffic_src
            per_unit = pj.get(roofline_dom["kernel"])
            if per_unit:
                roofline_dom["traffic"] = int(per_unit * nblocks / launches_per_step)
            # bound "valu": vector wave-instructions per unit from the SQ pass of the same build; peak = one wave-instruction per
            # SIMD every 4 cycles (what all but add / logic / right shift issue at on gfx950: profiles/ubench_issue2.hip) at 2.4 GHz
            vi = (pj.get("valu_wave_insts_per_unit") or {}).get(roofline_dom["kernel"])
            if vi:
                peak_g = 1024 * 2.4 / 4.0
                ach = vi * nblocks / launches_per_step / (avg_ms * 1e-3) / 1e9
                roofline_dom["valu"] = {"bound": "valu", "lane_instructions_per_input_byte": round(vi * 64 / BLOCK, 1),
                                        "achieved": round(ach, 1), "peak": round(peak_g, 1), "unit": "G wave-instructions/s",
                                        "frac": round(ach / peak_g, 3)}
        except Exception:
            pass

    free_b, total_b = devmem.mem_info(ctx)
    out = {
        "metric": "MB/s compress+decompress, 128 KiB blocks level 6",
        "value": round(total_size / dt * steps / 1e6, 1), "unit": "MB/s",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"one stream of {total_size >> 20} MiB seeded Zipf-word text ({uniq >> 20} MiB distinct, tiled), {size >> 20} MiB per GPU "
                               f"(contiguous block ranges, 32 KiB halo), 128 KiB blocks, level {args.level}: dict-chained deflate packed at final offsets"
                               f"{' + RCCL exchange of the slices (zngamd_comm_*: layout all-gather, grouped send/recv)' if exchange_stream else ' + layout exchange (sizes) over RCCL' if exchange else ''}, then two-pass inflate of "
                               f"{nblocks} independent gzip members written by this engine ('ZA' chunk index, flat dynamic headers)",
                   "block": BLOCK, "level": args.level, "bytes_per_gpu": size},
        "compress_MBps": round(total_size / (deflate_ms * 1e-3) / 1e6, 1),
        "decompress_MBps": round(total_size / (inflate_ms * 1e-3) / 1e6, 1),
        "ratio": round(size / comp_bytes, 4),
        "kernel_ms_per_step": {k: round(v[0] / steps, 3) for k, v in kt.items() if v[1]},
        "roofline": roofline,
        "dominant_kernel": roofline_dom,
        "roofline_deflate_pipeline": {"bound": "hbm", "kernels": "chains(A,B,C)+search+optparse+parse+plan+offsets+pack",
                                      "achieved": round((size + comp_bytes) / max(deflate_ms, 1e-9) / 1e6, 2), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round((size + comp_bytes) / max(deflate_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 5)},
        "device_memory_in_use_GB": round((total_b - free_b) / 1e9, 1),
        "roofline_inflate": {"bound": "hbm", "kernel": "za_k_inflate_members",
                             "achieved": round((ms_len.value + size) / max(kt["inflate"][0] / steps, 1e-9) / 1e6, 2),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round((ms_len.value + size) / max(kt["inflate"][0] / steps, 1e-9) / 1e6 / HBM_PEAK_GBS, 5),
                             "alg_bytes_per_launch": int(ms_len.value + size), "traffic": traffic_step.get("inflate_members"),
                             "traffic_source": traffic_src},
        "hbm_traffic_bytes_per_step": traffic_step or None,
        # what bounds every kernel here: vector wave-instructions per 128 KiB unit (SQ pass of the build named in traffic_source)
        "valu_wave_insts_per_unit": {k: int(v) for k, v in (pmc_per_unit.get("valu_wave_insts_per_unit") or {}).items()} or None,
    }
    if foreign is not None and pmc_per_unit.get("za_k_inflate_serial_members"):
        foreign["traffic"] = int(pmc_per_unit["za_k_inflate_serial_members"] * nblocks)
        foreign["traffic_source"] = traffic_src

    # ---- CPU baseline on this box's host cores (rank 0, N = 1 only) ------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = host_cores()
        # bounded sample of the same workload: the distinct text tiled like the GPU shard, 8 MiB per core
        sample = args.cpu_sample_mib << 20 if args.cpu_sample_mib else min(size, (16 << 20) * cores)
        sample -= sample % BLOCK
        arr = np.ascontiguousarray(np.tile(host, (sample + uniq - 1) // uniq)[:sample])
        td, ti, cb = O.bench_blocks(arr, BLOCK, args.level, cores)
        out["cpu_oracle"] = {"value": round(sample / (td + ti) / 1e6, 1), "unit": "MB/s", "cores": cores,
                             "kind": "port",
                             "sample": f"first {sample >> 20} MiB of the same tiled text, 128 KiB blocks + 32 KiB dictionary, "
                                       f"level {args.level}, oracle C codec (the parity checker, not a competitor) on {cores} threads",
                             "compress_MBps": round(sample / td / 1e6, 1),
                             "decompress_MBps": round(sample / ti / 1e6, 1), "ratio": round(sample / cb, 4)}
        # THE CPU baseline: the system zlib on the same sample, same protocol (deflateReset -> SetDictionary -> deflate(Z_SYNC_FLUSH)
        # per block, zlib_ngmodule.c:1725-1742), one block per task on all host cores (the pool's calls release the GIL) -- the
        # standard library every box has; a real zlib-ng, the reference's own CPU path, is probed for below
        from concurrent.futures import ThreadPoolExecutor
        mv = memoryview(arr)

        def zc(b):
            co = zlib.compressobj(args.level, zlib.DEFLATED, -15, 8, 0, bytes(mv[max(0, b * BLOCK - 32768):b * BLOCK])) \
                if b else zlib.compressobj(args.level, zlib.DEFLATED, -15)
            return co.compress(mv[b * BLOCK:(b + 1) * BLOCK]) + co.flush(zlib.Z_SYNC_FLUSH)

        def zd(bc):
            b, c = bc
            do = zlib.decompressobj(-15, zdict=bytes(mv[max(0, b * BLOCK - 32768):b * BLOCK])) if b else zlib.decompressobj(-15)
            return len(do.decompress(c))
        nb = sample // BLOCK
        with ThreadPoolExecutor(cores) as ex:
            t = time.perf_counter(); comp = list(ex.map(zc, range(nb))); tzc = time.perf_counter() - t
            t = time.perf_counter(); n_out = sum(ex.map(zd, enumerate(comp))); tzd = time.perf_counter() - t
        assert n_out == sample
        out["cpu_baseline"] = {"value": round(sample / (tzc + tzd) / 1e6, 1), "unit": "MB/s", "cores": cores,
                               "kind": "zlib " + zlib.ZLIB_RUNTIME_VERSION,
                               "sample": f"first {sample >> 20} MiB of the same tiled text, 128 KiB blocks + 32 KiB dictionary, "
                                         f"level {args.level}, system zlib {zlib.ZLIB_RUNTIME_VERSION} on {cores} threads (a cgroup share, not a socket)",
                               "compress_MBps": round(sample / tzc / 1e6, 1), "decompress_MBps": round(sample / tzd / 1e6, 1),
                               "ratio": round(sample / sum(map(len, comp)), 4)}
        out["cpu_zlib"] = dict(out["cpu_baseline"], library="zlib " + zlib.ZLIB_RUNTIME_VERSION)      # (the field's older name)
        # a real zlib-ng on this host would be the true reference CPU path; say plainly if there is none
        try:
            from zlib_ng import zlib_ng as _real                              # the reference's wheel, if the box has one
            zc_ng = lambda b: _real.compress(mv[b * BLOCK:(b + 1) * BLOCK], args.level, -15)
            with ThreadPoolExecutor(cores) as ex:
                t = time.perf_counter(); comp = list(ex.map(zc_ng, range(nb))); tn = time.perf_counter() - t
            out["cpu_zlib_ng"] = {"available": True, "compress_MBps": round(sample / tn / 1e6, 1), "cores": cores,
                                  "note": "independent blocks, no dictionary"}
        except Exception:
            out["cpu_zlib_ng"] = {"available": False, "note": "no zlib-ng wheel or library on this host: CPU column = zlib 1.2.x (cpu_baseline) + the oracle port (cpu_oracle)"}
    # ---- what the level compresses like OFF the bench corpus (rank 0, N = 1; outside the timed region): real files of this box
    # (zlib_ng_amd.corpus.heldout: Python sources, two ELF binaries, C headers; 4 MiB each) through the HIP path, 128 KiB units with
    # the previous 32 KiB as dictionary and a sync flush each -- the bench's own protocol -- beside the system zlib at the SAME level
    if rank == 0 and world == 1 and not args.no_heldout:
        from zlib_ng_amd import corpus as _corpus
        rh = {}
        for cname, cdata in _corpus.heldout(4 << 20).items():
            nbk = (len(cdata) + BLOCK - 1) // BLOCK
            blks = [(b * BLOCK, min(BLOCK, len(cdata) - b * BLOCK), 32768 if b else 0, 0) for b in range(nbk)]
            row = {}
            for lv in sorted({1, args.level, 9}):
                outs, _crcs, ovf = ctx.deflate_blocks(cdata, blks, lv, BLOCK + BLOCK // 8 + 600)
                assert not ovf
                stream = b"".join(outs)
                assert zlib.decompressobj(-15).decompress(stream + b"\x03\x00") == cdata, f"held-out {cname} level {lv} does not inflate"
                zt = 0
                for b in range(nbk):
                    zdct = cdata[max(0, b * BLOCK - 32768):b * BLOCK]
                    co = zlib.compressobj(lv, zlib.DEFLATED, -15, 8, 0, zdct) if zdct else zlib.compressobj(lv, zlib.DEFLATED, -15, 8, 0)
                    zt += len(co.compress(cdata[b * BLOCK:(b + 1) * BLOCK]) + co.flush(zlib.Z_SYNC_FLUSH))
                row[f"level_{lv}"] = {"ours": round(len(cdata) / len(stream), 4), "zlib_same_level": round(len(cdata) / zt, 4),
                                      "size_vs_zlib": round(len(stream) / zt, 4)}
            rh[cname] = row
        out["ratio_heldout"] = rh
        out["ratio_heldout_note"] = ("HIP path (zngamd_deflate_blocks) on files of this box, bench protocol; zlib " + zlib.ZLIB_RUNTIME_VERSION +
                                     " at the same level beside it; gate: size_vs_zlib <= 1.02 (tests/test_gpu_ratio_heldout.py)")
    if foreign is not None:
        out["roofline_inflate_foreign"] = foreign
    if chained is not None:
        # HBM bytes of the chunk pipeline's kernels for one decode of the stream: the counter passes give every kernel's average
        # launch (per unit of the launch size they were taken at); a decode launches each of them once per batch of compressed input
        # (ZNGAMD_CHUNK_BATCH_MIB, 512 by default), so average x batches = the kernel's share of one decode
        ck = ("za_k_scan_sync", "za_k_chunk_decode", "za_k_chunk_compose", "za_k_chunk_chain", "za_k_chunk_resolve")
        if all(pmc_per_unit.get(k) for k in ck) and pmc_per_unit.get("units_per_launch") == nblocks:      # (the shape the passes were taken at)
            batch = int(os.environ.get("ZNGAMD_CHUNK_BATCH_MIB", "512")) << 20
            nbatch = max(1, -(-comp_bytes // batch))
            chained["traffic"] = int(sum(pmc_per_unit[k] for k in ck) * nblocks * nbatch)
            chained["traffic_source"] = traffic_src + f"; {nbatch} batches per decode"
    # compress + the decode of the very stream the compress leg wrote (what gzip_ng_threaded.open("rb") gets for a file of the default
    # writer), beside `value`, whose decode leg reads indexed members: with the writer's index when the stream has one
    primary = chained_ix if chained_ix is not None else chained
    if primary is not None:
        out["value_chained"] = round(total_size / ((deflate_ms + primary["ms"]) * 1e-3) / 1e6, 1)
        out["value_chained_note"] = ("MB/s of compress + decode of the compress leg's own single-member stream " +
                                     ("with the writer's segment index" if chained_ix is not None else "without an index") +
                                     " (kernel time of the deflate leg + the whole device-resident decode call)")
        if chained_ix is not None and pmc_per_unit.get("za_k_inflate_units_marked"):
            ck2 = ("za_k_inflate_units_marked", "za_k_chunk_compose", "za_k_chunk_chain", "za_k_chunk_resolve")
            chained_ix["traffic"] = int(sum(pmc_per_unit.get(k, 0) for k in ck2) * nblocks)
            chained_ix["traffic_source"] = traffic_src
        out["roofline_inflate_chained"] = primary
        if chained_ix is not None and chained is not None:
            out["roofline_inflate_chained_noindex"] = chained
            out["value_chained_noindex"] = round(total_size / ((deflate_ms + chained["ms"]) * 1e-3) / 1e6, 1)
    if bgzf is not None:
        if pmc_per_unit.get("za_k_inflate_serial_members"):
            # (the counter passes average this kernel's launches over the foreign-member and the BGZF leg: both decode the same text)
            bgzf["traffic"] = int(pmc_per_unit["za_k_inflate_serial_members"] * nblocks)
        out["roofline_inflate_bgzf"] = bgzf
    # ---- the drop-in API over host buffers (PCIe, Python call overhead and fresh result objects included): NEVER `value`,
    # outside the timed region, rank 0 at N = 1 only.  One-shot calls on 256 MiB of the same text; the reference's own streaming
    # benchmark (benchmark_scripts/gzipwrite128kblocks.py:6-12, gzipread128kblocks.py:5-9): a gzip file written and read back
    # through gzip_ng_threaded.open in 128 KiB calls.
    if rank == 0 and world == 1 and not args.no_api:
        from zlib_ng_amd import zlib_ng, gzip_ng_threaded
        api_n = min(256 << 20, uniq * max(1, (256 << 20) // uniq))
        blob = bytes(np.tile(host, (api_n + uniq - 1) // uniq)[:api_n])

        def best_of(fn, reps=3):
            ts = []
            for _ in range(reps):
                t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
                del r
            return min(ts)
        comp_blob = zlib_ng.compress(blob, args.level, 31)
        assert zlib.decompress(comp_blob, 31) == blob
        t_c = best_of(lambda: zlib_ng.compress(blob, args.level, 31))
        assert zlib_ng.decompress(comp_blob, 31) == blob
        t_d = best_of(lambda: zlib_ng.decompress(comp_blob, 31))
        CALL = 128 * 1024
        mvb = memoryview(blob)

        import tempfile
        tmpdir = tempfile.mkdtemp(prefix="zng_bench_")
        gz_path = os.path.join(tmpdir, "api.gz")

        # the reference's write benchmark writes to os.devnull (gzipwrite128kblocks.py:7), its read benchmark reads a file: 1 GiB each
        # here (the 256 MiB four times over), the file written once outside the timing and checked with the system zlib
        STREAM_REPS = 4

        def w128(dst, members=False):
            with gzip_ng_threaded.open(dst, "wb", compresslevel=args.level, threads=8, block_size=CALL, indexed_members=members) as f:
                for _ in range(STREAM_REPS):
                    for o in range(0, api_n, CALL):
                        f.write(mvb[o:o + CALL])
        w128(gz_path)
        with open(gz_path, "rb") as fh:
            d = zlib.decompressobj(31)
            got = 0
            while True:
                piece = fh.read(64 << 20)
                if not piece:
                    break
                outp = memoryview(d.decompress(piece))
                while len(outp):                             # against the text, which repeats every api_n bytes
                    o = got % api_n
                    k = min(len(outp), api_n - o)
                    assert outp[:k] == mvb[o:o + k], "threaded writer: output differs"
                    outp = outp[k:]
                    got += k
            assert got == STREAM_REPS * api_n and d.eof
        t_w = best_of(lambda: w128(os.devnull), 3)

        def r128():
            got = 0
            with gzip_ng_threaded.open(gz_path, "rb", threads=8, block_size=CALL) as f:
                while True:
                    b = f.read(CALL)
                    if not b:
                        break
                    got += len(b)
            assert got == STREAM_REPS * api_n
        r128()
        t_r = best_of(r128, 2)
        # the same writer with indexed_members=True: independent members with this engine's chunk index, read back by the same reader
        # (one wavefront per member instead of the chunk pipeline); a gzip file for the system zlib as well
        w128(gz_path, True)
        with open(gz_path, "rb") as fh:
            assert fh.read(4) == b"\x1f\x8b\x08\x04"
        import gzip as _gz
        with _gz.open(gz_path, "rb") as fh:
            assert fh.read(api_n) == blob, "member writer: output differs"
        t_mw = best_of(lambda: w128(os.devnull, True), 3)
        r128()
        t_mr = best_of(r128, 2)
        # ... and through gzip_ng.open, which is what those two scripts open (one window over the whole stream; the engine's batches
        # run beside the caller): the same 1 GiB to os.devnull, and read back from the file gzip_ng.open wrote
        from zlib_ng_amd import gzip_ng

        def wopen(dst):
            with gzip_ng.open(dst, "wb", compresslevel=args.level) as f:
                for _ in range(STREAM_REPS):
                    for o in range(0, api_n, CALL):
                        f.write(mvb[o:o + CALL])
        wopen(gz_path)

        def ropen():
            got = 0
            with gzip_ng.open(gz_path, "rb") as f:
                while True:
                    b = f.read(CALL)
                    if not b:
                        break
                    got += len(b)
            assert got == STREAM_REPS * api_n
        ropen()
        t_ow = best_of(lambda: wopen(os.devnull), 2)
        t_or = best_of(ropen, 2)
        with gzip_ng.open(gz_path, "rb") as f:
            assert f.read(api_n) == blob, "gzip_ng.open: output differs"
        os.remove(gz_path)
        os.rmdir(tmpdir)
        # the sizes of the reference's own micro-benchmark (benchmark_scripts/benchmark.py:16-27, :56-75): latency of one call
        small = {}
        for sz in (1 << 10, 16 << 10, 64 << 10):
            piece = blob[:sz]
            zc_piece = zlib.compress(piece, args.level)
            assert zlib_ng.decompress(zlib_ng.compress(piece, args.level)) == piece and zlib_ng.decompress(zc_piece) == piece
            for _ in range(3):
                zlib_ng.compress(piece, args.level); zlib_ng.decompress(zc_piece)
            reps = 40
            t = time.perf_counter()
            for _ in range(reps):
                zlib_ng.compress(piece, args.level)
            tc_s = (time.perf_counter() - t) / reps
            t = time.perf_counter()
            for _ in range(reps):
                zlib_ng.decompress(zc_piece)
            td_s = (time.perf_counter() - t) / reps
            small[f"{sz >> 10}KiB"] = {"compress_us": round(tc_s * 1e6, 1), "decompress_us": round(td_s * 1e6, 1)}
        out["api"] = {"compress_MBps": round(api_n / t_c / 1e6, 1), "decompress_MBps": round(api_n / t_d / 1e6, 1),
                      "small_calls": small,
                      "threaded_write_MBps": round(STREAM_REPS * api_n / t_w / 1e6, 1), "threaded_read_MBps": round(STREAM_REPS * api_n / t_r / 1e6, 1),
                      "threaded_members_write_MBps": round(STREAM_REPS * api_n / t_mw / 1e6, 1), "threaded_members_read_MBps": round(STREAM_REPS * api_n / t_mr / 1e6, 1),
                      "open_write_MBps": round(STREAM_REPS * api_n / t_ow / 1e6, 1), "open_read_MBps": round(STREAM_REPS * api_n / t_or / 1e6, 1),
                      "note": f"host buffers, PCIe and fresh result objects included; {api_n >> 20} MiB of the same text, level {args.level}: zlib_ng.compress / "
                              "decompress (gzip container) one-shot; gzip_ng_threaded.open(threads=8, block_size=128 KiB) in 128 KiB calls as the reference's own "
                              f"benchmark scripts do it: {STREAM_REPS * api_n >> 20} MiB written to os.devnull, the same stream read back from a temporary file "
                              "(open_*: the same through gzip_ng.open, which is what the scripts themselves open; threaded_members_*: the threaded writer with "
                              "indexed_members=True and the same reader on its file); best of 2-3"}
        del blob, comp_blob
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
