"""BGZF-style file (64 KiB members with the 'BC' block-size subfield, what bgzip / htslib write): one wavefront per member."""
import os, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, gzip_ng
ctx = _lib.default_context()
data = corpus.fastq(256 << 20, seed=2).tobytes()
t = time.perf_counter()
parts = []
for o in range(0, len(data), 65280):
    blk = data[o:o + 65280]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(blk) + co.flush()
    bsize = 12 + 6 + len(raw) + 8 - 1
    parts.append(b"\x1f\x8b\x08\x04" + bytes(4) + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + raw
                 + struct.pack("<II", zlib.crc32(blk), len(blk)))
blob = b"".join(parts)
print("built %d members, %d MiB compressed in %.1f s" % (len(parts), len(blob) >> 20, time.perf_counter() - t))
ctx.gunzip(blob[:1 << 20] if False else blob, len(data))
ctx.profiling(True); ctx.kernel_times(True); ctx.decode_paths(True)
t = time.perf_counter(); code, out, nm = ctx.gunzip(blob, len(data)); dt = time.perf_counter() - t
kt = ctx.kernel_times(True)
assert code == 0 and out == data and nm == len(parts)
print("gunzip: %.0f MB/s wall incl. PCIe (%.0f ms); kernel ms %s; paths %s" % (len(data) / dt / 1e6, dt * 1e3,
      {k: round(v[0], 1) for k, v in kt.items() if v[1]}, ctx.decode_paths(True)))
path = "/tmp/x.bgzf.gz"
open(path, "wb").write(blob)
t = time.perf_counter(); n = 0
with gzip_ng.open(path, "rb") as f:
    while True:
        p = f.read(32 << 20)
        if not p: break
        n += len(p)
print("gzip_ng.open streaming: %.0f MB/s" % (n / (time.perf_counter() - t) / 1e6))
import gzip
part = b"".join(parts[:len(parts) // 8])
t = time.perf_counter(); n2 = len(gzip.decompress(part)); print("system gzip: %.0f MB/s" % (n2 / (time.perf_counter() - t) / 1e6))
os.remove(path)
