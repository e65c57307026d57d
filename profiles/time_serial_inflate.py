"""Times the sequential wavefront decoder (ordinary single-member gzip) and the BGZF one-launch path."""
import gzip, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib
ctx = _lib.default_context()
for name in ("test.fastq.gz", "test.fastq.bgzip.gz"):
    raw = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    exp = gzip.decompress(raw)
    ctx.gunzip(raw, len(exp))
    t = time.perf_counter(); code, out, nm = ctx.gunzip(raw, len(exp)); dt = time.perf_counter() - t
    assert code == 0 and out == exp
    t = time.perf_counter(); zlib.decompress(raw, 47) if nm == 1 else gzip.decompress(raw); dz = time.perf_counter() - t
    print(f"{name}: {nm} members, {len(exp)/dt/1e6:.1f} MB/s on the GPU path ({dt*1e3:.1f} ms), system zlib {len(exp)/dz/1e6:.1f} MB/s")
raw = open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz"), "rb").read()
ctx.profiling(True); ctx.kernel_times(True)
t = time.perf_counter(); ctx.gunzip(raw, 3578369); dt = time.perf_counter() - t
print("wall %.1f ms; kernel ms:" % (dt * 1e3), {k: round(v[0], 2) for k, v in ctx.kernel_times(True).items() if v[1]})

# chunk-parallel path on a larger stream: 256 MiB of text in the reference's threaded-writer framing
import struct
from zlib_ng_amd import corpus, shard
data = corpus.text(64 << 20, seed=1).tobytes() * 4
B = 131072
blocks = [(off, B, min(32768, off), 0) for off in range(0, len(data), B)]
outs, crcs, ovf = ctx.deflate_blocks(data, blocks, 6, B + B // 10)
crc = shard.combine_crcs([(c, B) for c in crcs])
hdr, trl = shard.gzip_frame(0, crc, len(data), 6)
blob = hdr + b"".join(outs) + trl
ctx.gunzip(blob, len(data))
ctx.profiling(True); ctx.kernel_times(True)
t = time.perf_counter(); code, out, nm = ctx.gunzip(blob, len(data)); dt = time.perf_counter() - t
kt = ctx.kernel_times(True)
assert code == 0 and out == data
print("threaded-writer framing, %d MiB, %d blocks: %.1f MB/s wall incl. PCIe (%.1f ms); kernel ms: %s" % (
    len(data) >> 20, len(blocks), len(data) / dt / 1e6, dt * 1e3, {k: round(v[0], 2) for k, v in kt.items() if v[1]}))

# ordinary gzip (no sync points): chunk starts from the block finder
for lvl in (1, 6, 9):
    d = data[:128 << 20]
    blob = gzip.compress(d, lvl)
    ctx.gunzip(blob, len(d))
    ctx.profiling(True); ctx.kernel_times(True)
    t = time.perf_counter(); code, out, nm = ctx.gunzip(blob, len(d)); dt = time.perf_counter() - t
    kt = ctx.kernel_times(True)
    assert code == 0 and out == d
    t = time.perf_counter(); zlib.decompress(blob, 47); dz = time.perf_counter() - t
    print("ordinary gzip -%d, %d MiB: %.1f MB/s wall incl. PCIe (%.1f ms), system zlib 1 core %.1f MB/s; kernel ms: %s" % (
        lvl, len(d) >> 20, len(d) / dt / 1e6, dt * 1e3, len(d) / dz / 1e6, {k: (round(v[0], 2), v[1]) for k, v in kt.items() if v[1]}))
