"""cProfile of gzip_ng.open written in 128 KiB calls (where the host time of the small-call pattern goes)."""
import cProfile, pstats, os, sys, time, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus, gzip_ng
CALL = 131072
base = corpus.text(64 << 20, seed=1).tobytes()
_lib.default_context()
blocks = [base[o:o + CALL] for o in range(0, len(base), CALL)]
def run():
    with gzip_ng.open(os.devnull, "wb", compresslevel=6) as f:
        for _ in range(4):
            for b in blocks:
                f.write(b)
run()
t = time.perf_counter(); run(); dt = time.perf_counter() - t
print("plain: %.0f MB/s" % (4 * len(base) / dt / 1e6))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue())
