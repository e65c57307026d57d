"""Chunk-parallel inflate of one-shot compress() output (256 MiB of text, gzip container) five times: run under
`rocprofv3 --kernel-trace --stats` for the kernels' shares (profiles/README.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import zlib_ng, corpus
blob = bytes(corpus.text(64 << 20, seed=5)) * 4
comp = zlib_ng.compress(blob, 6, 31)
for i in range(5):
    out = zlib_ng.decompress(comp, 31)
    assert len(out) == len(blob)
    del out
