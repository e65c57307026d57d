"""The sweep counters (profiling build, ps_stats.sh with PS_SCRIPT=profiles/ps_stats_small.py) on the small streams of the small-call bars."""
import ctypes, gzip, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, zlib_ng
L = _lib.load()
ctx = _lib.default_context()
names = ["sweeps", "lanes_kept", "sync_passes", "t_sync", "t_emit", "t_resolve", "sweeps_empty", "seq_rounds", "blocks", "t_tables",
         "ring_refills", "out_bytes", "eob_in_sweep", "lanes_exact", "resolve_rounds", "matches"]
def stats():
    a = (ctypes.c_ulonglong * 32)()
    assert L.zngamd_debug_ps_stats(a) == 0
    d = dict(zip(names, list(a)))
    d["hdr"] = [a[i] / 100.0 for i in range(20, 25)]
    return d
fq = gzip.open(os.path.join(ROOT, "tests", "golden", "test.fastq.gz")).read()
for size in (1 << 10, 4 << 10, 16 << 10, 64 << 10):
    d = fq[:size]
    z = zlib.compress(d, 6)
    assert zlib_ng.decompress(z) == d
    stats()
    ctx.profiling(True); ctx.kernel_times(True)
    zlib_ng.decompress(z)
    kt = ctx.kernel_times(True); ctx.profiling(False)
    s = stats()
    us = lambda k: s[k] / 100.0      # 100 MHz wall clock -> microseconds
    print(f"fastq {size} B L6: {len(z)} B in, kernel {kt['inflate'][0]*1e3:.0f} us | sweeps {s['sweeps']} (empty {s['sweeps_empty']}), passes/sweep "
          f"{s['sync_passes']/max(1,s['sweeps']):.2f}, kept {s['lanes_kept']/max(1,s['sweeps']):.1f}, out by sweeps {s['out_bytes']}, matches {s['matches']}, resolve rounds {s['resolve_rounds']} | "
          f"us: sync {us('t_sync'):.0f} emit {us('t_emit'):.0f} resolve {us('t_resolve'):.0f} tables {us('t_tables'):.0f} ({s['blocks']} blocks) | "
          f"seq rounds {s['seq_rounds']}, ring refills {s['ring_refills']}, eob in sweep {s['eob_in_sweep']} | header us: staging {s['hdr'][0]:.1f}, code-length table {s['hdr'][1]:.1f}, lengths {s['hdr'][2]:.1f}, literal/length table {s['hdr'][3]:.1f}, distance table {s['hdr'][4]:.1f}")
