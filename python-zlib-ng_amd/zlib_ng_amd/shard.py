"""Block sharding across ranks (multi-GPU leg).

The reference's only parallel strategy is round-robin blocks over threads with an in-order writer
(gzip_ng_threaded.py:316-321, :382-398).  Here every rank owns a contiguous range of blocks and compresses it on its
own GPU.  The member stream is reassembled with one padded all-gather of the variable-size slices (`allgather_stream`: RCCL
over xGMI with backend "nccl", every rank sends its slice over all its links at once; the same code runs on "gloo" for the
CPU tests).  Where nobody needs the whole stream in one memory, the ranks only have to agree on its LAYOUT -- where each
slice starts, the total size, CRC-32 / length of the whole input for the trailer: three integers per rank
(`exchange_layout`) -- and every rank writes its slice at its own offset.  Only torch.distributed plumbing and integer
arithmetic live here; no payload byte is computed on the host.
"""
import torch
import torch.distributed as dist

from . import _lib


def shard_range(n_blocks, rank, world):
    """Contiguous block range [lo, hi) of `rank` (replaces `index % threads`, gzip_ng_threaded.py:320)."""
    lo = (n_blocks * rank) // world
    hi = (n_blocks * (rank + 1)) // world
    return lo, hi


def exchange_layout(local_len, crc=0, ulen=0, group=None, device=None):
    """All-gather of (compressed bytes, CRC-32 of the uncompressed shard, uncompressed bytes) of every rank.  Returns
    (offset of this rank's slice in the stream, total compressed size, per-rank sizes, CRC-32 of the whole input,
    total uncompressed size): everything the writer's header / trailer and a positional write of the slice need
    (gzip_ng_threaded.py:382-398 folds the CRCs the same way, block by block)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([int(local_len), int(crc) & 0xFFFFFFFF, int(ulen)], dtype=torch.int64, device=device)
    parts = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    rows = [[int(v) for v in p.tolist()] for p in parts]
    sizes = [r[0] for r in rows]
    whole_crc = combine_crcs([(r[1], r[2]) for r in rows])
    return sum(sizes[:rank]), sum(sizes), sizes, whole_crc, sum(r[2] for r in rows)


def allgather_stream_start(local, local_len, group=None, scratch=None):
    """Start the exchange of the slices: sizes are all-gathered synchronously (8 bytes per rank), the payload all-gather
    (one collective into a contiguous [world x largest slice] buffer: no per-rank staging copies) is issued asynchronously
    so that independent work (the inflate leg) can overlap it.  Returns a handle for allgather_stream_finish."""
    world = dist.get_world_size(group)
    dev = local.device
    mine = torch.tensor([int(local_len)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    hs = [int(s.item()) for s in sizes]
    mx = max(max(hs), 1)
    if local.numel() < mx:
        raise ValueError("local buffer shorter than the largest slice")
    scratch = scratch if scratch is not None else {}
    need = world * mx
    if scratch.get("cap", 0) < need:
        scratch["buf"] = torch.empty(need + need // 8, dtype=torch.uint8, device=dev)
        scratch["cap"] = scratch["buf"].numel()
    flat = scratch["buf"][:need]
    work = dist.all_gather_into_tensor(flat, local[:mx].contiguous(), group=group, async_op=True)
    parts = [flat[r * mx:(r + 1) * mx] for r in range(world)]
    return {"work": work, "parts": parts, "sizes": hs, "scratch": scratch}


def allgather_stream_finish(h, compact=True):
    """Wait for the payload.  The slices lie in rank order (= block order: ranges are contiguous) at a stride of the
    largest slice; compact=True closes the gaps (one device copy per rank) and returns (stream tensor, total length,
    per-rank sizes); compact=False returns (list of slice views, total length, per-rank sizes) -- a writer can hand
    those to one positional / vectored write without touching the bytes again."""
    h["work"].wait()
    total = sum(h["sizes"])
    if not compact:
        return [p[:s] for p, s in zip(h["parts"], h["sizes"])], total, h["sizes"]
    scratch = h["scratch"]
    if scratch.get("scap", 0) < total:
        scratch["stream"] = torch.empty(total + total // 8 + 64, dtype=torch.uint8, device=h["parts"][0].device)
        scratch["scap"] = scratch["stream"].numel()
    off = 0
    stream = scratch["stream"]
    for r, s in enumerate(h["sizes"]):
        stream[off:off + s] = h["parts"][r][:s]
        off += s
    return stream, off, h["sizes"]


def allgather_stream(local, local_len, group=None, scratch=None):
    """All ranks contribute local[:local_len] (uint8, 1-D); every rank gets the slices concatenated in
    rank order.  Returns (stream tensor, total length, per-rank sizes).  `scratch` caches buffers."""
    return allgather_stream_finish(allgather_stream_start(local, local_len, group, scratch))


def combine_crcs(crcs_and_lens):
    """Fold (crc32, uncompressed_len) pairs in order, as _write does (gzip_ng_threaded.py:394)."""
    L = _lib.load()
    crc = 0
    for c, n in crcs_and_lens:
        crc = L.zngamd_crc32_combine(crc, c & 0xFFFFFFFF, n)
    return crc


def gzip_frame(body_len_total, crc, size, level):
    """Header / trailer bytes of the reference's threaded writer (gzip_ng_threaded.py:269-284, :324-338)."""
    import struct
    xfl = 2 if level == 9 else 4 if level == 1 else 0
    header = struct.pack("BBBBIBB", 0x1f, 0x8b, 8, 0, 0, 0xff, xfl)
    trailer = b"\x03\x00" + struct.pack("<II", crc & 0xFFFFFFFF, size & 0xFFFFFFFF)
    return header, trailer
