"""One-off compatibility sweep of the zlib_ng face against the system zlib over a parameter grid."""
import itertools, os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "python-zlib-ng_amd"))
from zlib_ng_amd import corpus, zlib_ng, gzip_ng
import gzip
datas = {"empty": b"", "one": b"x", "small": b"hello world " * 50, "text": corpus.text(700000, seed=1).tobytes(),
         "zeros": bytes(300000), "rand": os.urandom(200000)}
bad = 0
def check(label, fn):
    global bad
    try:
        ok = fn()
    except Exception as e:
        ok = repr(e)
    if ok is not True:
        bad += 1
        print("FAIL", label, ok)
for (dn, d), level, wb in itertools.product(datas.items(), (-1, 0, 1, 3, 6, 9), (9, 12, 15, -9, -15, 25, 31)):
    check(("compress", dn, level, wb), lambda: zlib.decompress(zlib_ng.compress(d, level, wb), wb) == d)
    check(("decompress", dn, level, wb), lambda: zlib_ng.decompress(zlib.compress(d, level, wb) if hasattr(zlib, "compress") and False else
                                                                  (lambda c: c.compress(d) + c.flush())(zlib.compressobj(level, zlib.DEFLATED, wb)), wb) == d)
    for strategy in (0, 1, 2, 3, 4):
        def obj():
            co = zlib_ng.compressobj(level, zlib_ng.DEFLATED, wb, 8, strategy)
            parts = [co.compress(d[:len(d) // 3]), co.flush(zlib_ng.Z_SYNC_FLUSH), co.compress(d[len(d) // 3:2 * len(d) // 3]),
                     co.flush(zlib_ng.Z_FULL_FLUSH), co.compress(d[2 * len(d) // 3:]), co.flush()]
            return zlib.decompress(b"".join(parts), wb) == d
        check(("compressobj", dn, level, wb, strategy), obj)
    def dobj():
        c = zlib.compressobj(level, zlib.DEFLATED, wb)
        blob = c.compress(d) + c.flush()
        do = zlib_ng.decompressobj(wb)
        out = b"".join(do.decompress(blob[i:i + 50000]) for i in range(0, max(len(blob), 1), 50000)) + do.flush()
        return out == d and do.eof
    check(("decompressobj", dn, level, wb), dobj)
for (dn, d), level in itertools.product(datas.items(), (0, 1, 6, 9)):
    check(("gzip_ng.compress", dn, level), lambda: gzip.decompress(gzip_ng.compress(d, level)) == d)
    check(("gzip_ng.decompress", dn, level), lambda: gzip_ng.decompress(gzip.compress(d, level)) == d)
    check(("crc32", dn), lambda: zlib_ng.crc32(d) == zlib.crc32(d) and zlib_ng.adler32(d) == zlib.adler32(d))
print("failures:", bad)
