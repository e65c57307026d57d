"""What the member decoder's wave waits for, counted on the CPU from the oracle's token streams (no GPU): test infrastructure.

For units of 128 KiB (indexed members: flat headers, 2 KiB segments, level 6) of several corpora:
  phase A   a lane decodes one 2 KiB segment in ROUNDS (za_k_inflate_members: up to six literals and the match behind them per round,
            a run of literals longer than six takes more rounds), the wave moves from 48-byte row to row of compressed input together:
            rounds of the mean lane, of the slowest lane, and of the wave as it runs (sum over rows of the slowest lane in the row);
            and what a token-balanced cut (segments of equal TOKEN count instead of equal output) would leave.
  phase B   queue entries per segment (a match, or a cut run of literals), groups of 64, and the share of groups in which no
            entry reads what an earlier entry of the same group writes (what an encoder-side "independent" bit could mark).
usage: python3 profiles/sim_inflate_balance.py [units per corpus]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "python-zlib-ng_amd")
from oracle import oracle as O
from zlib_ng_amd import corpus

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_BASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DIST_EXTRA = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
U, SEG = 131072, 2048


def unit_tokens(data, zd):
    """-> per segment: list of (is_match, bits, out_len, dist)"""
    c, crc, dbg = O.deflate_unit(data, zd, level=6, flags=2, debug=True)
    lens = dbg["lens"]
    nseg = (len(data) + SEG - 1) // SEG
    segs = []
    for s in range(nseg):
        t = dbg["tokens"][s * SEG:s * SEG + int(dbg["seg_ntok"][s])]
        toks = []
        for tk in t.tolist():
            if tk & 0x80000000:
                lc, dc = (tk >> 26) & 31, (tk >> 16) & 31
                ln = LEN_BASE[lc] + ((tk >> 21) & 31)
                dist = DIST_BASE[dc] + (tk & 0x1FFF)
                toks.append((1, int(lens[257 + lc]) + LEN_EXTRA[lc] + int(lens[288 + dc]) + DIST_EXTRA[dc], ln, dist))
            else:
                for i in range(((tk >> 24) & 3) + 1):
                    toks.append((0, int(lens[(tk >> (8 * i)) & 0xFF]), 1, 0))
        segs.append(toks)
    return segs, dbg


def rounds_of(toks):
    """the decoder's rounds over a token list: -> list of (bits consumed, entries pushed)"""
    out, i, gap = [], 0, 0
    while i < len(toks):
        bits, nl = 0, 0
        while nl < 6 and i < len(toks) and not toks[i][0]:
            bits += toks[i][1]; nl += 1; i += 1
        took = False
        if i < len(toks) and toks[i][0]:
            bits += toks[i][1]; i += 1; took = True
        g2 = gap + nl
        ent = 1 if (took or g2 >= 27) else 0
        gap = 0 if ent else g2
        out.append((bits, ent))
    return out


def wave_rounds(seg_rounds, start_bits):
    """lanes move row by row (48 bytes of input) together: per row the wave runs as many rounds as its slowest lane"""
    per_lane_rows = []
    for rs, sb in zip(seg_rounds, start_bits):
        pos = sb & 127                      # rows start at the 16-byte aligned address below the stream's first byte
        rows = {}
        for bits, _ in rs:
            rows[pos // 384] = rows.get(pos // 384, 0) + 1
            pos += bits
        per_lane_rows.append(rows)
    nrow = 1 + max((max(r) if r else 0) for r in per_lane_rows)
    return sum(max(r.get(k, 0) for r in per_lane_rows) for k in range(nrow))


def main():
    nunits = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    corp = {"text": corpus.text(nunits * U, 1).tobytes(), "fastq": corpus.fastq(nunits * U, 2).tobytes(), "mixed": corpus.mixed(nunits * U, 5).tobytes()}
    print("%-6s %8s %8s %8s %8s %8s | %8s %8s | %8s %8s %8s" % ("corpus", "mean", "slowest", "wave", "wave/mn", "slow/mn", "tb-slow", "tb-wave", "entries", "groups", "indep %"))
    for name, buf in corp.items():
        acc = np.zeros(8)
        for u in range(0, len(buf), U):
            data = buf[u:u + U]
            if len(data) < U:
                break
            segs, dbg = unit_tokens(data, b"")
            rs = [rounds_of(t) for t in segs]
            nr = np.array([len(r) for r in rs])
            starts = dbg["seg_bits"][:len(segs)].tolist()
            w = wave_rounds(rs, starts)
            # token-balanced: the unit's tokens dealt to 64 lanes in order, the same number of ROUNDS' worth each (by tokens)
            flat = [t for s in segs for t in s]
            per = (len(flat) + 63) // 64
            tb = [rounds_of(flat[k * per:(k + 1) * per]) for k in range(64)]
            tbn = np.array([len(r) for r in tb])
            bits_cum = np.cumsum([0] + [t[1] for t in flat])
            tb_starts = [int(starts[0] + bits_cum[min(k * per, len(flat))]) for k in range(64)]
            tbw = wave_rounds(tb, tb_starts)
            # phase B: entries and groups
            ent = grp = ind = 0
            for s, toks in enumerate(segs):
                # queue entries of the segment with destination ranges: a match entry writes [dst, dst + len), reads [dst - dist, ...)
                es, pos, gap = [], s * SEG, 0
                i = 0
                while i < len(toks):
                    nl = 0
                    while nl < 6 and i < len(toks) and not toks[i][0]:
                        nl += 1; i += 1; pos += 1
                    took = i < len(toks) and toks[i][0]
                    g2 = gap + nl
                    if took:
                        _, _, ln, d = toks[i]; i += 1
                        es.append((pos - g2, pos, ln, d)); pos += ln; gap = 0
                    elif g2 >= 27:
                        es.append((pos - g2, pos, 0, 0)); gap = 0
                    else:
                        gap = g2
                ent += len(es)
                for g in range(0, len(es), 64):
                    group = es[g:g + 64]
                    grp += 1
                    lo = group[0][0]
                    ok = True
                    for (l0, dst, ln, d) in group:
                        if ln and dst - d + min(ln, d) > lo and dst - d < dst:   # source ends above the group's first written byte
                            if dst - d + min(ln, d) > lo and (dst - d) < dst and (dst - d + min(ln, d)) > lo and dst > lo:
                                # reads something this group writes (anything at or above lo that lies below its own destination)
                                if dst - d + min(ln, d) > lo:
                                    ok = False
                                    break
                    ind += ok
            acc += [nr.mean(), nr.max(), w, 1, tbn.max(), tbw, ent, grp]
            acc[3] = 0
            ind_total = getattr(main, "_ind", {}).get(name, 0) + ind
            main._ind = dict(getattr(main, "_ind", {}), **{name: ind_total})
        k = len(buf) // U
        mean, slow, wv, _, tbs, tbwv, ent, grp = acc / k
        print("%-6s %8.1f %8.1f %8.1f %8.2f %8.2f | %8.1f %8.1f | %8.0f %8.1f %8.1f" % (name, mean, slow, wv, wv / mean, slow / mean, tbs, tbwv, ent, grp, 100.0 * main._ind[name] / max(1, grp * k)))


if __name__ == "__main__":
    main()
