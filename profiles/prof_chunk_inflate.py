"""Decodes one ordinary `gzip -6` member (128 MiB of text) three times: run under rocprofv3 --kernel-trace --stats
to see the split between block finder, count pass, marker decode, window propagation and resolve."""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus
ctx = _lib.default_context()
d = corpus.text(64 << 20, seed=1).tobytes() * 2
blob = gzip.compress(d, 6)
for _ in range(3):
    code, out, nm = ctx.gunzip(blob, len(d))
    assert code == 0 and out == d
print(ctx.decode_paths())
