import os, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
import numpy as np
from oracle import oracle as O
from zlib_ng_amd import _lib
ctx = _lib.default_context()
fastq = gzip.open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz")).read()
for name, data in (("fastq128k", fastq[:131072]), ("zeros", bytes(70000)), ("tiny5", b"hello"), ("fq5000", fastq[:5000])):
    for level in (7, 9):
        exp, exp_crc, dbg = O.deflate_unit(data, b"", level, 0, debug=True)
        try:
            got, crcs, ovf = ctx.deflate_blocks(data, [(0, len(data), 0, 0)], level, len(data) + 1024)
            msg = "ok" if got[0] == exp else "bytes differ"
        except Exception as e:
            msg = str(e)
        n = len(data)
        best = np.frombuffer(ctx.debug_fetch(1, 0, 4 * n), np.uint32)
        bad = np.flatnonzero(best != dbg["best"])
        print(name, level, msg, "best mismatches", len(bad), bad[:12])
        for i in bad[:6]:
            print("   pos", i, "got %08x exp %08x" % (best[i], dbg["best"][i]))
