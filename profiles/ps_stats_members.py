"""Counters of the self-synchronising sweeps (profiling build, see ps_stats.sh with PS_SCRIPT=profiles/ps_stats_members.py) on a
file of plain gzip members of 128 KiB written by the system zlib: the workload of the bench line's foreign-member leg."""
import ctypes, os, sys, time, zlib, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-zlib-ng_amd"))
from zlib_ng_amd import _lib, corpus
L = _lib.load()
ctx = _lib.default_context()
names = ["sweeps", "lanes_kept", "sync_passes", "t_sync", "t_emit", "t_resolve", "sweeps_empty", "seq_rounds", "blocks", "t_tables",
         "ring_refills", "out_bytes", "eob_in_sweep", "lanes_exact", "resolve_rounds", "matches",
         "count_rounds_wave", "count_cycles", "count_matches", "count_runs", "count_rounds_lanes"]
def stats():
    a = (ctypes.c_ulonglong * 32)()
    assert L.zngamd_debug_ps_stats(a) == 0
    return dict(zip(names, list(a)))
B = 131072
text = corpus.text(64 << 20, seed=1).tobytes()
for lvl in (1, 6, 9):
    mem = []
    for b in range(len(text) // B):
        co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
        raw = co.compress(text[b * B:(b + 1) * B]) + co.flush()
        mem.append(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + raw + struct.pack("<II", zlib.crc32(text[b * B:(b + 1) * B]), B))
    blob = b"".join(mem)
    code, out, nm = ctx.gunzip(blob, len(text))
    print("first decode: code", code, "equal", out == text)
    stats(); ctx.profiling(True); ctx.kernel_times(True)
    t = time.perf_counter(); ctx.gunzip(blob, len(text)); dt = time.perf_counter() - t
    kt = ctx.kernel_times(True); ctx.profiling(False)
    s = stats()
    us = lambda k: s[k] / 100.0
    n = max(1, s["sweeps"])
    print(f"zlib -{lvl}: {nm} members, {len(blob)} B in, wall {dt*1e3:.1f} ms, kernels {kt} | sweeps/member {s['sweeps']/nm:.1f} (empty {s['sweeps_empty']}), "
          f"passes/sweep {s['sync_passes']/n:.2f}, exact lanes/sweep {s['lanes_exact']/n:.1f}, kept {s['lanes_kept']/n:.1f}, out/sweep {s['out_bytes']/n:.0f}, "
          f"matches/sweep {s['matches']/n:.0f}, resolve rounds/sweep {s['resolve_rounds']/n:.1f} | wave-us per sweep: sync {us('t_sync')/n:.1f} emit {us('t_emit')/n:.1f} "
          f"resolve {us('t_resolve')/n:.1f}; tables {us('t_tables')/max(1,s['blocks']):.1f} per block ({s['blocks']/nm:.1f} blocks/member); seq rounds/member {s['seq_rounds']/nm:.1f} | "
          f"counting passes run {s['count_runs']/n:.2f} per sweep, {s['count_rounds_wave']/max(1,s['count_runs']):.1f} rounds each as the wave runs them "
          f"({s['count_rounds_lanes']/max(1,s['count_runs'])/64:.1f} per lane on average over 64), {s['count_cycles']/max(1,s['count_rounds_wave']):.0f} core cycles per round, "
          f"{s['count_cycles']/max(1,s['count_runs']):.0f} per pass")
