import sys, time, zlib
sys.path.insert(0, "/root/repo/python-zlib-ng_amd")
from zlib_ng_amd import zlib_ng, corpus
data = corpus.text(96 << 20, seed=1).tobytes()
for kind, blob, wb in (("zlib", zlib.compress(data, 6), 15), ("gzip", __import__("gzip").compress(data, 6), 31)):
    for feed in (1 << 20, 256 << 10):
        d = zlib_ng.decompressobj(wb)
        t = time.perf_counter(); n = 0; h = 0
        for i in range(0, len(blob), feed):
            out = d.decompress(blob[i:i + feed]); n += len(out); h = zlib.crc32(out, h)
        out = d.flush(); n += len(out); h = zlib.crc32(out, h)
        dt = time.perf_counter() - t
        assert n == len(data) and h == zlib.crc32(data) and d.eof, (n, len(data))
        print("%s decompressobj fed %d KiB at a time: %.0f MB/s" % (kind, feed >> 10, n / dt / 1e6))
