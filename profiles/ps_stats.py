"""Counters of the self-synchronising sweep (profiling build, see ps_stats.sh) on single streams of several kinds."""
import ctypes, gzip, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, zlib_ng, corpus
L = _lib.load()
ctx = _lib.default_context()
names = ["sweeps", "lanes_kept", "sync_passes", "t_sync", "t_emit", "t_resolve", "sweeps_empty", "seq_rounds", "blocks", "t_tables",
         "ring_refills", "out_bytes", "eob_in_sweep", "lanes_exact", "resolve_rounds", "matches"]
def stats():
    a = (ctypes.c_ulonglong * 32)()
    assert L.zngamd_debug_ps_stats(a) == 0
    return dict(zip(names, list(a)))
fq = gzip.open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz")).read()
text = corpus.text(1 << 20, seed=5).tobytes()
for label, d in (("fastq 4K", fq[:4096]), ("fastq 16K", fq[:16384]), ("fastq 128K", fq[:131072]), ("fastq 1M", fq[:1 << 20]), ("text 128K", text[:131072]), ("text 1M", text),
                 ("random 128K", os.urandom(131072)), ("zeros 1M", bytes(1 << 20))):
    for lvl in (1, 6, 9):
        z = zlib.compress(d, lvl)
        assert zlib_ng.decompress(z) == d
        stats()
        t = time.perf_counter(); zlib_ng.decompress(z); dt = time.perf_counter() - t
        s = stats()
        us = lambda k: s[k] / 100.0      # 100 MHz wall clock -> microseconds
        print(f"{label} L{lvl}: {len(z)} B in, wall {dt*1e3:.2f} ms | sweeps {s['sweeps']} (empty {s['sweeps_empty']}), passes/sweep "
              f"{s['sync_passes']/max(1,s['sweeps']):.2f}, exact lanes/sweep {s['lanes_exact']/max(1,s['sweeps']):.1f}, kept {s['lanes_kept']/max(1,s['sweeps']):.1f}, "
              f"out/sweep {s['out_bytes']/max(1,s['sweeps']):.0f}, matches {s['matches']}, resolve rounds {s['resolve_rounds']} | "
              f"us: sync {us('t_sync'):.0f} emit {us('t_emit'):.0f} resolve {us('t_resolve'):.0f} tables {us('t_tables'):.0f} ({s['blocks']} blocks) | "
              f"seq rounds {s['seq_rounds']}, ring refills {s['ring_refills']}, eob in sweep {s['eob_in_sweep']}")
