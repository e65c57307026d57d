"""Rehearsal of the multi-GPU exchange on one GPU: the same zngamd_comm_* calls bench.py --gpus N makes (RCCL communicator from a
unique id, layout all-gather, exchange of the slices, barrier, max) with world size 1: the collectives run through RCCL, the
own slice takes the device copy it takes at any N, only the send / receive pairs have no partner.  The layout arithmetic for N > 1 is covered on the CPU (tests/test_cpu_library.py, gloo, world size 2)."""
import ctypes as C
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

B = 131072


def test_comm_world_one_exchange(ctx):
    from zlib_ng_amd import _lib, corpus, shard
    L, h = ctx.L, ctx.h
    data = corpus.text(24 * B, seed=9).tobytes()
    nb = len(data) // B
    blocks = [(b * B, B, 32768 if b else 0, 0) for b in range(nb)]
    outs, crcs, ovf = ctx.deflate_blocks(data, blocks, 6, B + B // 8 + 600)
    assert not ovf
    body = b"".join(outs)
    comm = shard.Comm(ctx, shard.Comm.unique_id(), 0, 1)
    try:
        crc = shard.combine_crcs([(c, B) for c in crcs])
        off, total, sizes, whole_crc, whole_len = comm.layout(len(body), crc, len(data))
        assert (off, total, sizes, whole_len) == (0, len(body), [len(body)], len(data))
        assert whole_crc == zlib.crc32(data)

        def dmalloc(n):
            p = C.c_void_p()
            assert L.zngamd_dmalloc(h, n, C.byref(p)) == 0, ctx.err()
            return p
        d_local, d_stream = dmalloc(len(body) + 64), dmalloc(len(body) + 4096)
        try:
            assert L.zngamd_h2d(h, d_local, C.cast(C.c_char_p(body), C.c_void_p), len(body)) == 0
            comm.allgather_stream(d_local.value, sizes, d_stream.value, len(body) + 4096)
            comm.wait()
            back = np.empty(len(body), np.uint8)
            assert L.zngamd_d2h(h, back.ctypes.data_as(C.c_void_p), d_stream, len(body)) == 0
            assert back.tobytes() == body
            with pytest.raises(RuntimeError):          # a stream buffer that is too small is refused, nothing is sent
                comm.allgather_stream(d_local.value, sizes, d_stream.value, len(body) - 1)
        finally:
            L.zngamd_dfree(h, d_local)
            L.zngamd_dfree(h, d_stream)
        comm.barrier()
        assert comm.max(3.25) == 3.25
    finally:
        comm.close()
    header, trailer = shard.gzip_frame(total, whole_crc, whole_len, 6)
    import gzip
    assert gzip.decompress(header + body + trailer) == data


def test_bench_rehearses_the_exchange_on_one_gpu():
    """bench.py with BENCH_FORCE_EXCHANGE=1: the N > 1 code path (communicator, layout, slice exchange on its own stream, the
    assembled stream inflated on the device and compared with the whole input) with one rank."""
    import json
    env = dict(os.environ, BENCH_FORCE_EXCHANGE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size-mib", "256", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-foreign"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "RCCL exchange" in line["config"]["workload"]
