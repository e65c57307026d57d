import os, sys, time
os.environ["ZNGAMD_TRACE"]="1"
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "python-zlib-ng_amd"))
import numpy as np
from zlib_ng_amd import zlib_ng, corpus
blob = bytes(np.tile(corpus.text(64<<20, 1), 4))
for i in range(3):
    t=time.perf_counter(); c=zlib_ng.compress(blob, 6, 31); print("compress %.2f ms"%((time.perf_counter()-t)*1e3), file=sys.stderr)
for i in range(2):
    t=time.perf_counter(); d=zlib_ng.decompress(c, 31); print("decompress %.2f ms"%((time.perf_counter()-t)*1e3), file=sys.stderr)
