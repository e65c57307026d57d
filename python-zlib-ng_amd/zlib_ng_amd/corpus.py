"""Seeded synthetic corpora for benchmarks and tests (code, not data files).

text():  Zipf-word text (50 000-word vocabulary, s = 1.1) -- the "synthetic text" of BASELINE.json's
         configs; zlib level 6 compresses it about 3x.
fastq(): 4-line FASTQ-like records (ACGTN + Phred 33..73), mirroring the reference's tests/data.
mixed(): Silesia-like mixture for the level-9 ratio run.
"""
import numpy as np


def _vocab(rng, nwords):
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    lp = 1.0 / np.arange(1, 27) ** 0.8
    lp /= lp.sum()
    lens = np.clip(rng.geometric(0.22, nwords) + 1, 2, 14).astype(np.int64)
    flat = letters[rng.choice(26, size=int(lens.sum() + nwords), p=lp)]
    starts = np.zeros(nwords, np.int64)
    np.cumsum(lens[:-1] + 1, out=starts[1:])
    flat[starts + lens] = 32            # every word is stored with a trailing space
    return flat, starts, lens + 1


def text(nbytes, seed=1, nwords=50000, zipf_s=1.1):
    """nbytes of Zipf-word text as a numpy uint8 array."""
    rng = np.random.default_rng(seed)
    flat, starts, wl = _vocab(rng, nwords)
    p = 1.0 / np.arange(1, nwords + 1) ** zipf_s
    p /= p.sum()
    out = np.empty(nbytes, np.uint8)
    filled = 0
    mean = float((wl * p).sum())
    while filled < nbytes:
        want = min(nbytes - filled, 16 << 20)
        m = int(want / mean * 1.05) + 16
        idx = rng.choice(nwords, size=m, p=p)
        ln = wl[idx]
        ends = np.cumsum(ln)
        tot = int(ends[-1])
        pos = np.arange(tot, dtype=np.int64) - np.repeat(ends - ln, ln) + np.repeat(starts[idx], ln)
        chunk = flat[pos]
        nl = np.flatnonzero(chunk == 32)[11::12]     # a newline every 12 words
        chunk[nl] = 10
        take = min(tot, nbytes - filled)
        out[filled:filled + take] = chunk[:take]
        filled += take
    return out


def fastq(nbytes, seed=2, read_len=150):
    rng = np.random.default_rng(seed)
    rec = 60 + 2 * read_len
    nrec = nbytes // rec + 2
    parts = []
    bases = np.frombuffer(b"ACGTN", np.uint8)
    for i in range(0, nrec, 4096):
        k = min(4096, nrec - i)
        seq = bases[rng.choice(5, size=(k, read_len), p=[0.27, 0.23, 0.23, 0.26, 0.01])]
        qual = (33 + np.clip(rng.normal(34, 6, size=(k, read_len)), 0, 40)).astype(np.uint8)
        for j in range(k):
            hdr = b"@chr%d_%d_%d_0_0_0_0_0:0:0_0:0:0_%d/1\n" % (rng.integers(1, 23), rng.integers(1, 2 ** 28),
                                                              rng.integers(1, 2 ** 28), i + j)
            parts.append(hdr + seq[j].tobytes() + b"\n+\n" + qual[j].tobytes() + b"\n")
    return np.frombuffer(b"".join(parts)[:nbytes], np.uint8).copy()


def mixed(nbytes, seed=5):
    """Equal parts: text, XML-ish, FASTQ, int32 random walk, sparse zeros, opcode soup, random."""
    rng = np.random.default_rng(seed)
    part = nbytes // 7
    t = text(part, seed + 1)
    words = text(part, seed + 2)
    xml = np.frombuffer((b"<row id=\"%d\"><name>" % 7).ljust(24, b"x"), np.uint8)
    x = np.concatenate([np.concatenate([xml, words[i:i + 40], np.frombuffer(b"</name></row>\n", np.uint8)])
                        for i in range(0, part, 78)])[:part]
    f = fastq(part, seed + 3)
    walk = np.cumsum(rng.integers(-50, 51, size=part // 4), dtype=np.int64).astype("<i4").view(np.uint8)
    sparse = np.where(rng.random(part) < 0.03, rng.integers(1, 256, part), 0).astype(np.uint8)
    ops = np.frombuffer(bytes([0x48, 0x89, 0x8b, 0xe8, 0xc3, 0x55, 0x5d, 0x0f, 0x83, 0xff, 0x00, 0x01]), np.uint8)
    soup = ops[rng.choice(len(ops), size=part, p=np.array([12, 11, 10, 9, 8, 8, 8, 7, 7, 7, 7, 6]) / 100)]
    rnd = np.frombuffer(rng.bytes(nbytes - 6 * part), np.uint8)
    return np.concatenate([t, x, f, walk[:part], sparse, soup, rnd])


def heldout(max_bytes=4 << 20):
    """Held-out real files for the ratio gates (tests/test_oracle_ratio_heldout.py, bench.py `ratio_heldout`): data the level
    table was NOT designed on a generator for.  Assembled at run time from files any box with Python has; a corpus whose
    files are missing is left out.  -> {name: bytes}, each at most max_bytes."""
    import glob
    import os
    import sys
    import sysconfig

    def cat(paths):
        out, size = [], 0
        for f in paths:
            try:
                with open(f, "rb") as fh:
                    b = fh.read(max_bytes - size)
            except OSError:
                continue
            out.append(b)
            size += len(b)
            if size >= max_bytes:
                break
        return b"".join(out)

    res = {}
    py = cat(sorted(glob.glob(os.path.join(sysconfig.get_path("stdlib"), "*.py"))))
    if len(py) >= 1 << 20:
        res["python_sources"] = py
    exe = cat([os.path.realpath(sys.executable)])
    if len(exe) >= 1 << 20:
        res["python_elf"] = exe
    for cand in ("/usr/lib/x86_64-linux-gnu/libc.so.6", "/lib/x86_64-linux-gnu/libc.so.6", "/lib64/libc.so.6",
                 "/usr/lib64/libc.so.6", "/usr/lib/libc.so.6"):
        if os.path.isfile(cand):
            res["libc_elf"] = cat([cand])
            break
    hdr = cat(sorted(glob.glob("/usr/include/**/*.h", recursive=True)))
    if len(hdr) >= 1 << 20:
        res["c_headers"] = hdr
    return res
