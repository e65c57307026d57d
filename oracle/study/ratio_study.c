/* oracle/study -- TEST INFRASTRUCTURE ONLY: parameter studies for the ZA codec spec (round 5).
 * Includes the oracle's own stages and adds experimental variants of stage 1-3; prints compressed sizes
 * beside the system zlib's under the bench protocol (128 KiB units, 32 KiB dictionary = previous input,
 * sync flush per unit).   make -C oracle study;  oracle/study/ratio_study <corpus files...> < parameter lines
 *
 * A parameter line (what DESIGN.md 3.6 quotes; '#' lines are echoed):
 *   name  a_bytes a_bits a_chain  b_bytes b_bits b_chain  nice lazy cap  min_match too_far3 too_far4  seg blk  limL limD  parse ref
 *         [c_bytes c_bits c_chain  prop dp_sub dp_lit too_far5 dp_mlen]
 * a / b / c: up to three link tables (context bytes, bucket bits, steps walked; 0 bytes = off); parse 0 = greedy / one-step lazy,
 * 1 = backward dynamic programme (dp_sub shorter lengths tried; dp_lit 0 flat costs, 1 bytes of the whole unit, 2 bytes of the
 * match-less positions + log-odds terms, 3 + length / distance code statistics, 4.. = 2 with matches shorter than dp_lit counted as
 * literals, env WKD = only if farther back than that); blk = Huffman block size (0 = the unit); ref = index of the zlib level the
 * percentage is taken against (0..4 = -1 -3 -4 -6 -9).  The spec that came out of it: "L6 5 14 2 3 14 1 16 0 16 3 4096 32768 2048 0 10 9 1 3
 * 12 14 1 0 4 2 32768 12" plus the weak-match rule (WKD=256, dp_lit 4) and sampled statistics, which only the oracle itself has.
 */
#include "../oracle_deflate.c"
#include <stdio.h>
#include <zlib.h>

typedef struct {
    int a_bytes, a_bits, a_chain;     /* primary chain: context bytes, bucket bits, steps */
    int b_bytes, b_bits, b_chain;     /* secondary chain (0 bytes = off) */
    int nice, lazy, cap;
    int min_match;                    /* 3 or 4 */
    int too_far3, too_far4;           /* drop len==3 (len==4) matches farther than this */
    int seg;                          /* token boundary granularity */
    int blk;                          /* Huffman block = this many input bytes (0 = whole unit) */
    int limL, limD;
    int parse;                        /* 0 greedy/lazy per level, 1 = backward DP */
    int c_bytes, c_bits, c_chain;     /* third table */
    int prop;                         /* candidate (len-1, dist) from best[p-1] */
    int dp_sub;                       /* DP: how many shorter lengths are tried */
    int dp_lit;                       /* DP literal cost model: 0 = flat 8.3, 1 = from the unit's byte histogram */
    int too_far5;
    int dp_mlen;
} sp;

static uint32_t hashn(const uint8_t *p, int nb, int bits)
{
    uint32_t lo = ld32(p);
    if (nb == 3) lo &= 0xFFFFFF;
    uint32_t h = lo * 2654435761u;
    if (nb > 4) { uint32_t hi = nb >= 8 ? ld32(p + 4) : (ld32(p + 4) & (0xFFFFFFFFu >> (8 * (8 - nb)))); h ^= hi * 2246822519u; }
    if (nb > 8) { uint32_t hi = nb >= 12 ? ld32(p + 8) : (ld32(p + 8) & (0xFFFFFFFFu >> (8 * (12 - nb)))); h = (h ^ (h >> 15)) * 2246822519u ^ hi * 3266489917u; }
    if (nb > 12) { uint32_t hi = ld32(p + 12); h = (h ^ (h >> 13)) * 3266489917u ^ hi * 668265263u; }
    return h >> (32 - bits);
}
static void chains_n(const uint8_t *data, int dict_len, int n, int nb, int bits, uint16_t *prevdist)
{
    int32_t *head = malloc(sizeof(int32_t) << bits);
    for (int i = 0; i < (1 << bits); i++) head[i] = -1;
    for (int p = -dict_len; p < n; p++) {
        int i = p + dict_len;
        if (p + nb > n) { prevdist[i] = 0; continue; }
        int32_t P = ZA_WIN + p;
        uint32_t h = hashn(data + p, nb, bits);
        int32_t q = head[h];
        int32_t d = q >= 0 ? P - q : 0;
        prevdist[i] = (uint16_t)((d >= 1 && d <= ZA_WIN) ? d : 0);
        head[h] = P;
    }
    free(head);
}

static void walk(const uint8_t *data, int dict_len, const uint16_t *prev, int p, int depth, int cap, int nice,
                 int *best_len, int *best_dist)
{
    int q = p;
    while (depth-- > 0) {
        int d = prev[q + dict_len];
        if (d == 0) break;
        q -= d;
        int dist = p - q;
        if (dist > ZA_WIN) break;
        int len = 0;
        while (len < cap && data[q + len] == data[p + len]) len++;
        if (len > *best_len || (len == *best_len && dist < *best_dist)) {
            *best_len = len; *best_dist = dist;
            if (len >= nice) break;
        }
    }
}

static long study_unit(const uint8_t *data, int dict_len, int n, const sp *S)
{
    uint16_t *pa = malloc(2 * (size_t)(dict_len + n)), *pb = malloc(2 * (size_t)(dict_len + n)), *pc = malloc(2 * (size_t)(dict_len + n));
    uint32_t *best = malloc(4 * (size_t)n);
    chains_n(data, dict_len, n, S->a_bytes, S->a_bits, pa);
    if (S->b_bytes) chains_n(data, dict_len, n, S->b_bytes, S->b_bits, pb);
    if (S->c_bytes) chains_n(data, dict_len, n, S->c_bytes, S->c_bits, pc);
    for (int p = 0; p < n; p++) {
        int seg_end = (p / S->seg + 1) * S->seg; if (seg_end > n) seg_end = n;
        int maxlen = seg_end - p; if (maxlen > 258) maxlen = 258;
        best[p] = 0;
        if (maxlen < S->min_match) continue;
        int cap = S->cap < maxlen ? S->cap : maxlen;
        int nice = S->nice < cap ? S->nice : cap;
        int bl = S->min_match - 1, bd = 0;
        walk(data, dict_len, pa, p, S->a_chain, cap, nice, &bl, &bd);
        if (S->b_bytes && bl < nice) walk(data, dict_len, pb, p, S->b_chain, cap, nice, &bl, &bd);
        if (S->c_bytes && bl < nice) walk(data, dict_len, pc, p, S->c_chain, cap, nice, &bl, &bd);
        if (S->prop && p > 0 && best[p - 1]) {
            int pl = (int)(best[p - 1] >> 16) - 1, pd = (int)(best[p - 1] & 0xFFFF);
            if (pl > cap) pl = cap;
            if (pl > bl) { bl = pl; bd = pd; }
        }
        if (bl < S->min_match) continue;
        if (bl == cap) while (bl < maxlen && data[p - bd + bl] == data[p + bl]) bl++;
        if (bl == 3 && bd > S->too_far3) continue;
        if (bl == 4 && bd > S->too_far4) continue;
        if (bl == 5 && bd > S->too_far5) continue;
        best[p] = ((uint32_t)bl << 16) | (uint32_t)bd;
    }
    /* parse + cost, per Huffman block */
    int blk = S->blk ? S->blk : n;
    uint64_t bits = 0;
    uint8_t *choice = malloc((size_t)n + 1);   /* DP: chosen length at p (1 = literal) */
    uint16_t *choice2 = malloc(2 * ((size_t)n + 1));
    if (S->parse == 1) {
        /* backward DP per segment, integer costs in 1/4 bit */
        uint32_t *cost = malloc(sizeof(uint32_t) * ((size_t)n + 1));
        int litc[256];
        for (int i = 0; i < 256; i++) litc[i] = 33;
        int mbias = 0; int lenc[29], dstc[30]; for (int i = 0; i < 29; i++) lenc[i] = -1; for (int i = 0; i < 30; i++) dstc[i] = -1;
        if (S->dp_lit >= 1) {
            /* dp_lit 1: bytes of the whole unit; 2: bytes at unmatched positions + P(literal) from the match statistics */
            uint32_t h[256] = {0}; uint64_t tot = 0, nm = 0; int wk = S->dp_lit >= 4 ? S->dp_lit : 1; int wkd = getenv("WKD") ? atoi(getenv("WKD")) : 0;
#define WEAK(i) ((int)(best[i] >> 16) < wk && ((best[i] >> 16) == 0 || (int)(best[i] & 0xFFFF) > wkd))
            for (int i = 0; i < n; i++) {
                if (S->dp_lit == 1 || WEAK(i)) { h[data[i]]++; tot++; }
                int li = (int)(best[i] >> 16), lp = i ? (int)(best[i - 1] >> 16) : 0;
                if (!WEAK(i) && li != lp - 1) nm++;
            }
            if (S->dp_lit >= 2) {
                /* smooth: every byte value gets a floor share so unseen bytes are not free of charge / infinitely dear */
                for (int i = 0; i < 256; i++) { h[i] = h[i] * 16 + 1 + (uint32_t)(tot / 64); }
                tot = 0; for (int i = 0; i < 256; i++) tot += h[i];
            }
            /* P(lit) = U / (U + nm): extra cost log2((U+nm)/U) on every literal, log2((U+nm)/nm) on every match */
            uint64_t U = 0; for (int i = 0; i < n; i++) U += WEAK(i);
            int lbias = 0;
            if (S->dp_lit >= 2) {
                uint64_t a = U + nm; if (U == 0) U = 1; if (nm == 0) nm = 1;
                uint64_t r = (a << 8) / U; int lg = 0; while (r >= 512) { r >>= 1; lg++; }
                lbias = 4 * lg + (r >= 431 ? 3 : r >= 362 ? 2 : r >= 304 ? 1 : 0);
                r = (a << 8) / nm; lg = 0; while (r >= 512) { r >>= 1; lg++; }
                mbias = 4 * lg + (r >= 431 ? 3 : r >= 362 ? 2 : r >= 304 ? 1 : 0);
                if (lbias > 24) lbias = 24; if (mbias > 24) mbias = 24;
            }
            if (S->dp_lit == 3) {
                uint32_t hl[29] = {0}, hd[30] = {0}; uint64_t tl = 0;
                for (int i = 0; i < n; i++) {
                    int li = (int)(best[i] >> 16), lp = i ? (int)(best[i - 1] >> 16) : 0;
                    if (li && li != lp - 1) { hl[len_code(li)]++; hd[dist_code((int)(best[i] & 0xFFFF))]++; tl++; }
                }
                uint64_t t2 = 0;
                for (int i = 0; i < 29; i++) { hl[i] = hl[i] * 16 + 1 + (uint32_t)(tl / 8); t2 += hl[i]; }
                for (int i = 0; i < 29; i++) { uint64_t r = (t2 << 8) / hl[i]; int lg = 0; while (r >= 512) { r >>= 1; lg++; }
                    lenc[i] = 4 * lg + (r >= 431 ? 3 : r >= 362 ? 2 : r >= 304 ? 1 : 0); }
                t2 = 0;
                for (int i = 0; i < 30; i++) { hd[i] = hd[i] * 16 + 1 + (uint32_t)(tl / 8); t2 += hd[i]; }
                for (int i = 0; i < 30; i++) { uint64_t r = (t2 << 8) / hd[i]; int lg = 0; while (r >= 512) { r >>= 1; lg++; }
                    dstc[i] = 4 * lg + (r >= 431 ? 3 : r >= 362 ? 2 : r >= 304 ? 1 : 0); }
            }
            for (int i = 0; i < 256; i++) {
                uint64_t r = h[i] ? ((uint64_t)tot << 8) / h[i] : ((uint64_t)1 << 20);
                int lg = 0; uint64_t t = r; while (t >= 512) { t >>= 1; lg++; }
                int frac = t >= 431 ? 3 : t >= 362 ? 2 : t >= 304 ? 1 : 0;
                int c = 4 * lg + frac + (S->dp_lit == 1 ? 2 : lbias);
                if (c < 12) c = 12; if (c > 52) c = 52;
                litc[i] = c;
            }
        }
        for (int s0 = 0; s0 < n; s0 += S->seg) {
            int e = s0 + S->seg > n ? n : s0 + S->seg;
            cost[e] = 0;
            for (int p = e - 1; p >= s0; p--) {
                uint32_t c = (uint32_t)litc[data[p]] + cost[p + 1]; int ch = 1;
                int len = (int)(best[p] >> 16), dist = (int)(best[p] & 0xFFFF);
                if (len) {
                    int dc = dist_code(dist);
                    uint32_t dcost = dstc[dc] >= 0 ? (uint32_t)(dstc[dc] + 4 * dist_extra[dc]) : 4 * (5 + dist_extra[dc]);
                    int lo = len - S->dp_sub; if (lo < S->min_match) lo = S->min_match;
                    for (int l = len; l >= lo; l--) {
                        if (l == 3 && dist > S->too_far3) break;
                        int lc = len_code(l);
                        uint32_t mc = (S->dp_lit == 3 ? (uint32_t)(lenc[lc] + mbias + 4 * len_extra[lc]) : S->dp_lit >= 2 ? (uint32_t)(S->dp_mlen + mbias + 4 * len_extra[lc]) : 4 * (6 + len_extra[lc]) + (lc < 8 ? 0 : 4)) + dcost + cost[p + l];
                        if (mc < c) { c = mc; ch = l; }
                    }
                }
                cost[p] = c; choice2[p] = (uint16_t)ch;
            }
        }
        free(cost);
    }
    for (int b0 = 0; b0 < n; b0 += blk) {
        int b1 = b0 + blk > n ? n : b0 + blk;
        uint32_t hist[320]; memset(hist, 0, sizeof hist);
        int p = b0;
        /* blocks are aligned to segments when blk is a multiple of seg */
        while (p < b1) {
            int seg_end = (p / S->seg + 1) * S->seg; if (seg_end > n) seg_end = n;
            uint32_t bb = best[p]; int len = (int)(bb >> 16);
            if (S->parse == 1) {
                int ch = choice2[p];
                if (ch == 1 || len == 0) { hist[data[p]]++; p++; continue; }
                int l = ch;
                int dist = (int)(bb & 0xFFFF);
                hist[257 + len_code(l)]++; hist[288 + dist_code(dist)]++; p += l; continue;
            }
            if (len >= S->min_match) {
                if (S->lazy && len < S->lazy && p + 1 < seg_end && (int)(best[p + 1] >> 16) > len) { hist[data[p]]++; p++; continue; }
                int dist = (int)(bb & 0xFFFF);
                hist[257 + len_code(len)]++; hist[288 + dist_code(dist)]++; p += len;
            } else { hist[data[p]]++; p++; }
        }
        hist[256] = 1;
        uint32_t fl[288], fd[32]; uint8_t lens[320]; memset(lens, 0, sizeof lens);
        memcpy(fl, hist, 4 * 288); memcpy(fd, hist + 288, 4 * 32);
        int cntd = 0; for (int i = 0; i < 30; i++) cntd += fd[i] != 0;
        if (cntd < 2 && fd[0] == 0) { fd[0] = 1; cntd++; }
        if (cntd < 2) fd[1] = 1;
        huff_lengths(fl, 286, S->limL, lens); huff_lengths(fd, 30, S->limD, lens + 288);
        int hlit = 286; while (hlit > 257 && lens[hlit - 1] == 0) hlit--;
        int hdist = 30; while (hdist > 1 && lens[288 + hdist - 1] == 0) hdist--;
        uint8_t seq[320]; memcpy(seq, lens, hlit); memcpy(seq + hlit, lens + 288, hdist);
        uint16_t cltok[320]; int ncl = rle_lengths(seq, hlit + hdist, cltok);
        uint32_t clf[19]; memset(clf, 0, sizeof clf); for (int i = 0; i < ncl; i++) clf[cltok[i] & 0xFF]++;
        uint8_t cl_lens[19]; huff_lengths(clf, 19, 7, cl_lens);
        int hclen = 19; while (hclen > 4 && cl_lens[cl_order[hclen - 1]] == 0) hclen--;
        uint64_t dd = 0, df = 0;
        for (int i = 0; i < 286; i++) { uint32_t f = hist[i]; if (!f) continue; int ex = i >= 257 ? len_extra[i - 257] : 0;
            int fx = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; dd += (uint64_t)f * (lens[i] + ex); df += (uint64_t)f * (fx + ex); }
        for (int i = 0; i < 30; i++) { uint32_t f = hist[288 + i]; dd += (uint64_t)f * (lens[288 + i] + dist_extra[i]); df += (uint64_t)f * (5 + dist_extra[i]); }
        uint64_t hd = 17 + 3 * (uint64_t)hclen;
        for (int i = 0; i < ncl; i++) { int s = cltok[i] & 0xFF; hd += cl_lens[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0); }
        uint64_t cd = hd + dd, cf = 3 + df, cs = 8 * ((uint64_t)(b1 - b0) + 5) + 7;
        uint64_t c = cd; if (cf <= c) c = cf; if (cs <= c) c = cs;
        bits += c;
    }
    bits += 3 + 7 + 32;  /* sync marker, avg pad */
    free(pa); free(pb); free(pc); free(best); free(choice); free(choice2);
    return (long)((bits + 7) / 8);
}

static long zlib_unit(const uint8_t *data, int dict_len, int n, int level)
{
    static __thread z_stream zs; static __thread int init_level = -99;
    if (init_level != level) { if (init_level != -99) deflateEnd(&zs); memset(&zs, 0, sizeof zs); deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY); init_level = level; }
    else deflateReset(&zs);
    if (dict_len) deflateSetDictionary(&zs, data - dict_len, (uInt)dict_len);
    static __thread uint8_t out[200000];
    zs.next_in = (Bytef *)data; zs.avail_in = (uInt)n; zs.next_out = out; zs.avail_out = sizeof out;
    deflate(&zs, Z_SYNC_FLUSH);
    return (long)(sizeof out - zs.avail_out);
}

static uint8_t *load(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc((size_t)sz + 64); memset(b + sz, 0, 64);
    if (fread(b, 1, (size_t)sz, f) != (size_t)sz) exit(1);
    fclose(f); *n = (size_t)sz; return b;
}

typedef struct { const uint8_t *d; size_t n; const sp *S; int zl; int t, nt; long sum; } job;
static void *worker(void *v)
{
    job *j = v; long s = 0; size_t nu = (j->n + ZA_MAX_UNIT - 1) / ZA_MAX_UNIT;
    for (size_t u = (size_t)j->t; u < nu; u += (size_t)j->nt) {
        size_t off = u * ZA_MAX_UNIT; int len = (int)(j->n - off > ZA_MAX_UNIT ? ZA_MAX_UNIT : j->n - off);
        int dict = off > ZA_WIN ? ZA_WIN : (int)off;
        if (j->S) s += study_unit(j->d + off, dict, len, j->S); else s += zlib_unit(j->d + off, dict, len, j->zl);
    }
    j->sum = s; return NULL;
}
static long run(const uint8_t *d, size_t n, const sp *S, int zl)
{
    enum { NT = 8 }; pthread_t th[NT]; job J[NT]; long s = 0;
    for (int t = 0; t < NT; t++) { J[t] = (job){ d, n, S, zl, t, NT, 0 }; pthread_create(&th[t], NULL, worker, &J[t]); }
    for (int t = 0; t < NT; t++) { pthread_join(th[t], NULL); s += J[t].sum; }
    return s;
}

int main(int argc, char **argv)
{
    /* argv: files...  ; parameter sets from stdin, one per line:
       name a_bytes a_bits a_chain b_bytes b_bits b_chain nice lazy cap min_match too_far3 too_far4 seg blk limL limD parse */
    int nf = argc - 1; uint8_t **D = malloc(sizeof *D * nf); size_t *N = malloc(sizeof *N * nf);
    for (int i = 0; i < nf; i++) D[i] = load(argv[i + 1], &N[i]);
    printf("%-34s", "config");
    for (int i = 0; i < nf; i++) { const char *b = strrchr(argv[i + 1], '/'); printf(" %10.10s", b ? b + 1 : argv[i + 1]); }
    printf("\n");
    int zls[] = {1, 3, 4, 6, 9};
    long zs[5][16];
    for (int k = 0; k < 5; k++) {
        char nm[32]; sprintf(nm, "zlib -%d", zls[k]); printf("%-34s", nm);
        for (int i = 0; i < nf; i++) { zs[k][i] = run(D[i], N[i], NULL, zls[k]); printf(" %10.4f", (double)N[i] / zs[k][i]); }
        printf("\n");
    }
    char line[512];
    while (fgets(line, sizeof line, stdin)) {
        if (line[0] == '#' || line[0] == '\n') { fputs(line, stdout); continue; }
        char nm[64]; sp S; int ref = 3; memset(&S, 0, sizeof S);
        int k = sscanf(line, "%63s %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d", nm, &S.a_bytes, &S.a_bits, &S.a_chain, &S.b_bytes, &S.b_bits, &S.b_chain,
                   &S.nice, &S.lazy, &S.cap, &S.min_match, &S.too_far3, &S.too_far4, &S.seg, &S.blk, &S.limL, &S.limD, &S.parse, &ref, &S.c_bytes, &S.c_bits, &S.c_chain, &S.prop, &S.dp_sub, &S.dp_lit, &S.too_far5, &S.dp_mlen);
        if (k < 26) S.too_far5 = 32768;
        if (k < 18) { fprintf(stderr, "bad line: %s", line); continue; }
        printf("%-34s", nm);
        for (int i = 0; i < nf; i++) { long c = run(D[i], N[i], &S, 0); printf(" %6.4f%+5.1f", (double)N[i] / c, 100.0 * ((double)zs[ref][i] / c - 1)); }
        printf("   (%% vs zlib -%d)\n", zls[ref]);
        fflush(stdout);
    }
    return 0;
}
