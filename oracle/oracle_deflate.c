/* oracle/ -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The "ZA codec": this repo's deterministic DEFLATE encoder specification, restated in scalar C.
 * It stands where the reference calls zng_deflateReset / zng_deflateSetDictionary /
 * zng_deflate(Z_SYNC_FLUSH) (zlib_ngmodule.c:1725, :1735, :1742).  zlib-ng's source is absent from
 * /root/reference (un-vendored submodule) and no reference test pins compressed bytes, so the
 * encoder choices below are ours; what is pinned is that the output is valid RFC 1951 that inflates
 * to the input (tests/test_oracle_*.py cross-decode with the system zlib).  The HIP kernels in
 * python-zlib-ng_amd/csrc/ are an independent implementation of the same five stages and must match
 * this file byte for byte, stage by stage.
 *
 *   stage 1  link tables        link[p] = distance to the nearest earlier position whose context hashes into the same
 *                               14-bit bucket, if within 32 768: table A over 5-byte contexts (walked as a chain),
 *                               tables B (3 bytes) and C (12 bytes) as one candidate each
 *   stage 2  match search       best[p] = longest match among the first `chain` entries of chain A and the
 *                               candidates of B and C, nearest wins ties, truncated at the 2 KiB segment end
 *   stage 3a dynamic programme  (levels 4-9) per 2 KiB segment, backwards: literal or match (or the match shortened by
 *                               up to 4) by estimated bit costs; rewrites best[]
 *   stage 3  parse              per 2 KiB segment, greedy over best[]
 *   stage 4  entropy plan       histograms -> length-limited canonical Huffman (Moffat-Katajainen
 *                               in-place lengths + count-based limiting), stored/fixed/dynamic pick
 *   stage 5  bit packing        header, tokens, EOB, then sync-flush marker or final padding
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <time.h>

typedef struct { int chain, nice, cap, use_c, dp, too_far3, too_far4; } za_level;
/* Level table.  Three link tables: A = chains over 5-byte contexts, walked `chain` steps; B = the nearest
 * earlier position with the same 3-byte context (one candidate, every level); C = the nearest earlier position with the same 12-byte
 * context (one candidate, levels 5-9: "skip ahead in chain A to the first long match").  Levels 1-3 take the longest candidate
 * greedily; levels 4-9 choose literals / matches per 2 KiB segment by a backward dynamic programme over estimated bit costs (dp).
 * cap: candidates are compared on their first `cap` bytes only (longest wins, nearest wins ties); the winner is then extended to
 * its true length.  16 on the fast levels (one 16-byte compare per candidate on the GPU), 258 = compare in full, and only those
 * levels stop a walk early at `nice` equal bytes.  too_far3 / too_far4: a match of 3 / 4 bytes farther back than this is dropped.
 * Calibrated against zlib 1.2.11 at the same level on held-out real files (Python sources, C headers, ELF) as well as the synthetic
 * corpora: tests/test_oracle_ratio_heldout.py. */
static const za_level LEVELS[10] = {
    {0, 0, 0, 0, 0, 0, 0},
    {1, 16, 16, 0, 0, 256, 4096}, {2, 16, 16, 0, 0, 256, 4096}, {3, 16, 16, 0, 0, 256, 4096},
    {2, 16, 16, 0, 1, 4096, 32768}, {2, 16, 16, 1, 1, 4096, 32768}, {3, 16, 16, 1, 1, 4096, 32768},
    {4, 32, 258, 1, 1, 4096, 32768}, {8, 64, 258, 1, 1, 4096, 32768}, {12, 128, 258, 1, 1, 4096, 32768}
};

/* The segment size of a unit (token boundaries are forced at segment ends; at most 64 segments): 2 KiB for full units and for
 * indexed members, smaller for the units of small calls -- the smallest power of two >= 32 that covers the unit with 64 segments --
 * so that the kernels, which give a lane to each segment, do not walk a small input with one lane. */
int za_o_seg_shift(int n, int flags)
{
    if ((flags & (ZA_FLAG_FLATHDR | ZA_FLAG_SEG2K)) || n > 65536) return 11;
    int s = 5;
    while ((64 << s) < n) s++;
    return s;
}

/* test-only override of the level table (parameter studies); chain <= 0 switches it off */
static za_level g_override = {0, 0, 258, 1, 1, 4096, 32768};
void za_o_override_level(int chain, int nice, int cap, int use_c, int dp, int too_far3, int too_far4)
{
    g_override = (za_level){ chain, nice, cap, use_c, dp, too_far3, too_far4 };
}

static inline uint32_t ld32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
/* buckets of the three link tables (all ZA_HASH_BITS wide): A over 5 bytes, B over 3, C over 12 */
#define ZA_K1 2654435761u
#define ZA_K2 2246822519u
#define ZA_K3 3266489917u
static const int za_hash_bytes[3] = { ZA_HASH_BYTES_A, ZA_HASH_BYTES_B, ZA_HASH_BYTES_C };
static inline uint32_t hash_of(const uint8_t *p, int table)
{
    uint32_t x;
    if (table == 0) x = (ld32(p) * ZA_K1) ^ ((uint32_t)p[4] * ZA_K2);
    else if (table == 1) x = ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) * (ZA_K1 << 8);     /* three bytes, no fourth read (it could not reach the product's low 32 bits anyway) */
    else {
        x = (ld32(p) * ZA_K1) ^ (ld32(p + 4) * ZA_K2);
        x = ((x ^ (x >> 15)) * ZA_K2) ^ (ld32(p + 8) * ZA_K3);
    }
    return x >> (32 - ZA_HASH_BITS);
}

/* ---------------- stage 1 ---------------- */
static void stage1_chains(const uint8_t *data, int dict_len, int n, int table, uint16_t *prevdist)
{
    /* head[h] = the nearest earlier position of bucket h, or none.  No 16-bit wrap-around: a bucket's link is the true
     * nearest earlier position of the bucket if that lies within 32 768 and not in front of the dictionary, else none --
     * so the links of a position do not depend on whether positions further back than its window were ever seen (what lets
     * the chain kernel carry its tables from one unit of a stream to the next instead of inserting the dictionary again).
     * A position with fewer bytes left in the unit than the table's context is not inserted. */
    static __thread int32_t head[1 << ZA_HASH_BITS];
    const int hb = za_hash_bytes[table];
    for (int i = 0; i < (1 << ZA_HASH_BITS); i++) head[i] = -1;
    for (int p = -dict_len; p < n; p++) {
        int i = p + dict_len;
        if (p + hb > n) { prevdist[i] = 0; continue; }
        int32_t P = ZA_WIN + p;
        uint32_t h = hash_of(data + p, table);
        int32_t q = head[h];
        int32_t d = q >= 0 ? P - q : 0;
        prevdist[i] = (uint16_t)((d >= 1 && d <= ZA_WIN) ? d : 0);
        head[h] = P;
    }
}

/* ---------------- stage 2 ---------------- */
/* one candidate at distance `dist`: its first `cap` bytes against the position's; longer wins, nearer wins ties */
static inline void try_candidate(const uint8_t *data, int p, int dist, int cap, int *best_len, int *best_dist)
{
    const uint8_t *q = data + p - dist;
    int len = 0;
    while (len < cap && q[len] == data[p + len]) len++;
    if (len > *best_len || (len == *best_len && dist < *best_dist)) { *best_len = len; *best_dist = dist; }
}

static uint32_t stage2_search(const uint8_t *data, int dict_len, int n, const uint16_t *linkA, const uint16_t *linkB,
                              const uint16_t *linkC, int p, const za_level *L, int max_dist, int seg)
{
    int seg_end = (p / seg + 1) * seg;
    if (seg_end > n) seg_end = n;
    int maxlen = seg_end - p;
    if (maxlen > ZA_MAX_MATCH) maxlen = ZA_MAX_MATCH;
    if (maxlen < ZA_MIN_MATCH) return 0;
    const int cap = L->cap < maxlen ? L->cap : maxlen;      /* bytes compared per candidate */
    const int nice = L->nice < cap ? L->nice : cap;
    int best_len = ZA_MIN_MATCH - 1, best_dist = 0;
    int q = p, depth = L->chain;
    while (depth-- > 0) {                                     /* table A: a walk */
        int d = linkA[q + dict_len];
        if (d == 0) break;
        q -= d;
        int dist = p - q;
        if (dist > max_dist) break;
        try_candidate(data, p, dist, cap, &best_len, &best_dist);
        if (best_len >= nice) break;
    }
    /* tables B and C: the position's own link, one candidate each.
     * Levels that compare 16 bytes (1-6): a link counts only if the candidate really shares the table's context -- its first 3
     * (B) or 12 (C) bytes -- and then stands for a match of exactly that length (longer wins, nearer wins ties, as before); a
     * winner that came from B or C is extended to its true length like a winner at `cap`.  (What the 16-byte compare of these
     * two candidates bought beyond that was nothing on any corpus: a B candidate that shares five bytes is in chain A, and a
     * C candidate beats the walk only when the walk stopped short.)  C is left out where fewer than 12 bytes remain.
     * Levels that compare in full (7-9): both are compared in full, and skipped behind a nice match. */
    int from_bc = 0;
    for (int t = 1; t <= (L->use_c ? 2 : 1); t++) {
        int d = (t == 1 ? linkB : linkC)[p + dict_len];
        if (d == 0 || d > max_dist) continue;
        if (L->cap > 16) {
            if (best_len >= nice) break;
            try_candidate(data, p, d, cap, &best_len, &best_dist);
            continue;
        }
        const int ctx = t == 1 ? ZA_HASH_BYTES_B : ZA_HASH_BYTES_C;
        if (cap < ctx || memcmp(data + p - d, data + p, (size_t)ctx) != 0) continue;
        if (ctx > best_len || (ctx == best_len && d < best_dist)) { best_len = ctx; best_dist = d; from_bc = 1; }
    }
    if (best_len < ZA_MIN_MATCH) return 0;
    if (best_len == cap || from_bc)                           /* the winner's true length */
        while (best_len < maxlen && data[p - best_dist + best_len] == data[p + best_len]) best_len++;
    if (best_len == 3 && best_dist > L->too_far3) return 0;
    if (best_len == 4 && best_dist > L->too_far4) return 0;
    return ((uint32_t)best_len << 16) | (uint32_t)best_dist;
}

/* ---------------- symbol maps ---------------- */
/* ---------------- symbol maps ---------------- */
static const uint16_t len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t  len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t  dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

static int len_code(int len)     /* 0..28 */
{
    int c = 28;
    while (len_base[c] > len) c--;
    return c;
}
static int dist_code(int dist)   /* 0..29 */
{
    int c = 29;
    while (dist_base[c] > dist) c--;
    return c;
}

/* ---------------- stage 3a: dynamic programme ---------------- */
/* 4 * log2(a / b) in whole quarter bits, a >= b >= 1, a < 2^22 */
static int ilog4(uint32_t a, uint32_t b)
{
    uint32_t q = (a << 8) / b;                       /* the ratio with 8 fractional bits, >= 256 */
    int lg = 0;
    while ((q >> lg) >= 512u) lg++;
    uint32_t t = q >> lg;                             /* 256 .. 511 */
    return 4 * lg + (t >= 304u) + (t >= 362u) + (t >= 431u);     /* 256 * 2^(1/4), 2^(2/4), 2^(3/4) */
}

/* Bit costs in quarter bits, estimated from the search results of the unit alone (no pass over tokens exists yet).  A position
 * is "literal-like" if it has no match, or only a 3-byte match farther back than ZA_DP_WEAK_DIST (such a match costs about what
 * its three literals cost; on data whose every position has one -- a small alphabet at random -- counting them as matches would
 * price literals out of the parse altogether):
 * (all of it counted over a SAMPLE of the unit, the positions p with (p >> 8) & 3 == 0 -- every fourth block of 256: the costs
 * are estimates of estimates, and a quarter of the entries gives the same parse within 0.03 % of size on every corpus tried)
 *   U  = literal-like positions, hU[b] = their bytes: the literals to come are mostly these;
 *   NM = other positions whose length is not one less than the length in front of them: "a new match starts here", the
 *        number of match tokens to come;
 *   a literal b costs log2(T / h'[b]) + log2((U + NM) / U) (h' = 16 hU + 1 + U / 64: unseen bytes are dear, not impossible),
 *   a match costs 3 bits of length code + log2((U + NM) / NM) + its extra bits + 5 bits of distance code + its extra bits. */
static void dp_costs(const uint8_t *data, int n, const uint32_t *best, uint32_t *cost /* [258] */)
{
    uint32_t h[256], U = 0, NM = 0, T = 0;
    memset(h, 0, sizeof h);
    for (int p = 0; p < n; p++) {
        if (((p >> 8) & 3) != 0) continue;              /* a sample: every fourth block of 256 positions */
        uint32_t len = best[p] >> 16, lp = p ? best[p - 1] >> 16 : 0;
        if (len == 0 || (len == 3 && (best[p] & 0xFFFF) > ZA_DP_WEAK_DIST)) { h[data[p]]++; U++; }
        else if (len + 1 != lp) NM++;
    }
    for (int b = 0; b < 256; b++) { h[b] = 16 * h[b] + 1 + (U >> 6); T += h[b]; }
    int lbias = ilog4(U + NM, U ? U : 1), mbias = ilog4(U + NM, NM ? NM : 1);
    if (lbias > 24) lbias = 24;
    if (mbias > 24) mbias = 24;
    for (int b = 0; b < 256; b++) {
        int c = ilog4(T, h[b]) + lbias;
        cost[b] = (uint32_t)(c < 12 ? 12 : c > 52 ? 52 : c);
    }
    cost[256] = (uint32_t)(12 + mbias + 20);          /* length code + distance code */
    cost[257] = 0;
}

static void stage3a_dp(const uint8_t *data, int n, uint32_t *best, const za_level *L, uint32_t *cost_out, int seg)
{
    uint32_t cost[258];
    static __thread uint32_t acc[ZA_SEG + 1];
    dp_costs(data, n, best, cost);
    if (cost_out) memcpy(cost_out, cost, sizeof cost);
    for (int s0 = 0; s0 < n; s0 += seg) {
        int e = s0 + seg > n ? n : s0 + seg;
        acc[e - s0] = 0;
        for (int p = e - 1; p >= s0; p--) {
            uint32_t c = cost[data[p]] + acc[p + 1 - s0];
            int choice = 0;                                    /* 0 = literal */
            int len = (int)(best[p] >> 16), dist = (int)(best[p] & 0xFFFF);
            if (len) {
                uint32_t mc0 = cost[256] + 4u * dist_extra[dist_code(dist)];
                int lo = len - ZA_DP_SUB < ZA_MIN_MATCH ? ZA_MIN_MATCH : len - ZA_DP_SUB;
                for (int l = len; l >= lo; l--) {             /* the longest first: it keeps a tie */
                    if (l == 3 && dist > L->too_far3) break;          /* (what the search would have dropped) */
                    uint32_t mc = mc0 + 4u * len_extra[len_code(l)] + acc[p + l - s0];
                    if (mc < c) { c = mc; choice = l; }
                }
            }
            acc[p - s0] = c;
            best[p] = choice ? ((uint32_t)choice << 16) | (uint32_t)dist : 0u;
        }
    }
}

/* ---------------- stage 4 helpers ---------------- */
static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* Length-limited code lengths.  Moffat & Katajainen, "In-place calculation of minimum-redundancy
 * codes" (1995) on the symbols sorted by (freq, index); depths above `limit` are folded by the
 * classic count-based adjustment (one code leaves length `limit`, one shorter code splits), then
 * lengths are dealt out longest-first to the rarest symbols. */
static void huff_lengths(const uint32_t *freq, int n, int limit, uint8_t *lens)
{
    uint32_t key[288];
    uint32_t A[288];
    int m = 0;
    memset(lens, 0, (size_t)n);
    for (int i = 0; i < n; i++) if (freq[i]) key[m++] = (freq[i] << 9) | (uint32_t)i;
    if (m == 0) return;
    if (m == 1) { lens[key[0] & 511] = 1; return; }
    qsort(key, (size_t)m, sizeof key[0], cmp_u32);
    for (int i = 0; i < m; i++) A[i] = key[i] >> 9;
    {
        int root, leaf, next, avbl, used, dpth;
        A[0] += A[1]; root = 0; leaf = 2;
        for (next = 1; next < m - 1; next++) {
            if (leaf >= m || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint32_t)next; }
            else A[next] = A[leaf++];
            if (leaf >= m || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint32_t)next; }
            else A[next] += A[leaf++];
        }
        A[m - 2] = 0;
        for (next = m - 3; next >= 0; next--) A[next] = A[A[next]] + 1;
        avbl = 1; used = dpth = 0; root = m - 2; next = m - 1;
        while (avbl > 0) {
            while (root >= 0 && (int)A[root] == dpth) { used++; root--; }
            while (avbl > used) { A[next--] = (uint32_t)dpth; avbl--; }
            avbl = 2 * used; dpth++; used = 0;
        }
    }
    int cnt[17];
    memset(cnt, 0, sizeof cnt);
    int over = 0;
    for (int i = 0; i < m; i++) {
        int d = (int)A[i];
        if (d > limit) { d = limit; over = 1; }
        cnt[d]++;
    }
    if (over) {
        uint32_t total = 0;
        for (int i = 1; i <= limit; i++) total += (uint32_t)cnt[i] << (limit - i);
        while (total != (1u << limit)) {
            cnt[limit]--;
            for (int i = limit - 1; i > 0; i--)
                if (cnt[i]) { cnt[i]--; cnt[i + 1] += 2; break; }
            total--;
        }
    }
    int idx = 0;
    for (int l = limit; l >= 1; l--)
        for (int k = 0; k < cnt[l]; k++) lens[key[idx++] & 511] = (uint8_t)l;
}

/* canonical codes (RFC 1951 3.2.2), returned bit-reversed so they can be emitted LSB first */
static void canon_codes(const uint8_t *lens, int n, uint16_t *codes)
{
    int bl_count[16] = {0};
    uint32_t next_code[16];
    for (int i = 0; i < n; i++) bl_count[lens[i]]++;
    bl_count[0] = 0;
    uint32_t code = 0;
    for (int b = 1; b <= 15; b++) { code = (code + (uint32_t)bl_count[b - 1]) << 1; next_code[b] = code; }
    for (int i = 0; i < n; i++) {
        int l = lens[i];
        codes[i] = 0;
        if (!l) continue;
        uint32_t c = next_code[l]++, r = 0;
        for (int b = 0; b < l; b++) if (c & (1u << b)) r |= 1u << (l - 1 - b);
        codes[i] = (uint16_t)r;
    }
}

/* ---------------- bit writer ---------------- */
typedef struct { uint8_t *out; size_t cap; size_t pos; uint64_t acc; int nb; int overflow; } bitwr;
static inline void putbits(bitwr *w, uint32_t v, int n)
{
    w->acc |= (uint64_t)v << w->nb;
    w->nb += n;
    while (w->nb >= 8) {
        if (w->pos < w->cap) w->out[w->pos] = (uint8_t)w->acc; else w->overflow = 1;
        w->pos++; w->acc >>= 8; w->nb -= 8;
    }
}
static inline void flushbyte(bitwr *w) { if (w->nb) putbits(w, 0, 8 - w->nb); }
static inline uint64_t bitpos(const bitwr *w) { return (uint64_t)w->pos * 8 + (uint64_t)w->nb; }

static const uint8_t cl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

/* run-length encode the concatenated code-length sequence; out tokens: sym | extra<<8 */
static int rle_lengths(const uint8_t *seq, int n, uint16_t *tok)
{
    int nt = 0, i = 0;
    while (i < n) {
        int v = seq[i], run = 1;
        while (i + run < n && seq[i + run] == v) run++;
        i += run;
        if (v == 0) {
            while (run >= 3) {
                if (run >= 11) { int r = run > 138 ? 138 : run; tok[nt++] = (uint16_t)(18 | ((r - 11) << 8)); run -= r; }
                else { tok[nt++] = (uint16_t)(17 | ((run - 3) << 8)); run = 0; }
            }
            while (run-- > 0) tok[nt++] = 0;
        } else {
            tok[nt++] = (uint16_t)v; run--;
            while (run >= 3) { int r = run > 6 ? 6 : run; tok[nt++] = (uint16_t)(16 | ((r - 3) << 8)); run -= r; }
            while (run-- > 0) tok[nt++] = (uint16_t)v;
        }
    }
    return nt;
}

/* window limit for the next calls (zlib-container / raw streams with wbits < 15): distances above it
 * are never emitted.  32768 = full window. */
static __thread int g_max_dist = ZA_WIN;
void za_o_set_max_dist(int max_dist) { g_max_dist = (max_dist < 1 || max_dist > ZA_WIN) ? ZA_WIN : max_dist; }

long za_o_deflate_unit(const uint8_t *data, int dict_len, int n, int level, int flags,
                       uint8_t *out, size_t cap, uint32_t *crc, za_o_debug *dbg)
{
    if (level == -1) level = 6;
    if (level < 0 || level > 9 || n < 0 || n > ZA_MAX_UNIT || dict_len < 0 || dict_len > ZA_WIN)
        return ZA_STREAM_ERROR;
    const za_level *L = (g_override.chain > 0 && level > 0) ? &g_override : &LEVELS[level];
    int final = (flags & ZA_FLAG_FINAL) != 0;
    int flat = (flags & ZA_FLAG_FLATHDR) != 0;
    bitwr w = { out, cap, 0, 0, 0, 0 };
    if (crc) *crc = za_o_crc32(0, data, (size_t)n);
    const int sshift = za_o_seg_shift(n, flags), seg = 1 << sshift;
    int nseg = (n + seg - 1) / seg;
    if (dbg && dbg->btype) *dbg->btype = -1;

    if (n == 0) {
        if (final) { putbits(&w, 3, 3); putbits(&w, 0, 7); flushbyte(&w); }       /* 03 00 */
        else { putbits(&w, 0, 3); flushbyte(&w); putbits(&w, 0, 16); putbits(&w, 0xFFFF, 16); }
        return w.overflow ? ZA_BUF_ERROR : (long)w.pos;
    }

    int use_stored = (level == 0);
    uint32_t hist[320];
    uint8_t lens[320];
    uint16_t codes[320];
    uint32_t *tokens = NULL, *best = NULL;
    uint16_t *prevdist = NULL, *linkB = NULL, *linkC = NULL;
    uint32_t seg_ntok[ZA_MAX_SEGS];
    uint32_t seg_bits[ZA_MAX_SEGS + 1];
    uint32_t chunk_idx[ZA_MAX_CHUNKS + 1];
    int btype = 0;
    uint8_t cl_lens[19]; uint16_t cl_codes[19];
    uint16_t cltok[320]; int ncltok = 0, hlit = 257, hdist = 1, hclen = 4;
    memset(hist, 0, sizeof hist);
    memset(lens, 0, sizeof lens);
    memset(seg_ntok, 0, sizeof seg_ntok);
    memset(seg_bits, 0, sizeof seg_bits);
    memset(chunk_idx, 0, sizeof chunk_idx);

    if (!use_stored) {
        prevdist = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(dict_len + n));
        linkB = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(dict_len + n));
        linkC = (uint16_t *)calloc((size_t)(dict_len + n), sizeof(uint16_t));
        best = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
        tokens = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nseg * seg));
        if (!prevdist || !linkB || !linkC || !best || !tokens) { free(prevdist); free(linkB); free(linkC); free(best); free(tokens); return ZA_MEM_ERROR; }
        stage1_chains(data, dict_len, n, 0, prevdist);
        stage1_chains(data, dict_len, n, 1, linkB);
        if (L->use_c) stage1_chains(data, dict_len, n, 2, linkC);
        for (int p = 0; p < n; p++) best[p] = stage2_search(data, dict_len, n, prevdist, linkB, linkC, p, L, g_max_dist, seg);
        if (dbg) {
            if (dbg->prevdist) memcpy(dbg->prevdist, prevdist, sizeof(uint16_t) * (size_t)(dict_len + n));
            if (dbg->linkB) memcpy(dbg->linkB, linkB, sizeof(uint16_t) * (size_t)(dict_len + n));
            if (dbg->linkC) memcpy(dbg->linkC, linkC, sizeof(uint16_t) * (size_t)(dict_len + n));
            if (dbg->best) memcpy(dbg->best, best, sizeof(uint32_t) * (size_t)n);
        }
        /* stage 3a (levels 4-9): the dynamic programme rewrites the entries -- a position it makes a literal loses its match, a
         * match it shortens gets the shorter length -- so that the greedy walk of stage 3 follows its choices */
        if (L->dp) stage3a_dp(data, n, best, L, dbg ? dbg->dp_cost : NULL, seg);
        if (dbg && dbg->best_dp) memcpy(dbg->best_dp, best, sizeof(uint32_t) * (size_t)n);
        /* stage 3: parse every segment on its own */
        for (int s = 0; s < nseg; s++) {
            int p = s * seg, end = p + seg;
            uint32_t *t = tokens + (size_t)s * seg;
            int nt = 0;
            if (end > n) end = n;
            while (p < end) {
                uint32_t b = best[p];
                int len = (int)(b >> 16);
                if (len >= ZA_MIN_MATCH) {
                    int dist = (int)(b & 0xFFFF);
                    /* a match token carries its symbols: bit 31, length code << 26, length extra << 21, distance
                     * code << 16, distance extra (the packer needs no symbol arithmetic) */
                    int lc = len_code(len), dc = dist_code(dist);
                    t[nt++] = 0x80000000u | ((uint32_t)lc << 26) | ((uint32_t)(len - len_base[lc]) << 21) |
                              ((uint32_t)dc << 16) | (uint32_t)(dist - dist_base[dc]);
                    hist[257 + lc]++;
                    hist[288 + dc]++;
                    p += len;
                } else {
                    /* a position without any match: the literal takes up to two more such positions into its token word
                     * (bytes in bits 0..23, count - 1 in bits 24..25), as far as the 32-position chunk reaches in which the
                     * kernel decides it: chunks are counted from the segment start, and a position is decided in the chunk
                     * of its successor (its successor's entry must be at hand) */
                    uint32_t tk = data[p]; int nl = 1;
                    hist[data[p]]++; p++;
                    int lim = s * seg + 32 * ((p - s * seg) / 32 + 1);
                    if (lim > end) lim = end;
                    while (nl < 3 && p < lim && (best[p] >> 16) == 0) { tk |= (uint32_t)data[p] << (8 * nl); hist[data[p]]++; nl++; p++; }
                    t[nt++] = tk | ((uint32_t)(nl - 1) << 24);
                }
            }
            seg_ntok[s] = (uint32_t)nt;
        }
        hist[256] = 1;
        if (dbg) {
            if (dbg->tokens) memcpy(dbg->tokens, tokens, sizeof(uint32_t) * (size_t)(nseg * seg));
            if (dbg->seg_ntok) memcpy(dbg->seg_ntok, seg_ntok, sizeof seg_ntok);
            if (dbg->hist) memcpy(dbg->hist, hist, sizeof hist);
        }
        /* stage 4: entropy plan */
        uint32_t fl[288], fd[32];
        memcpy(fl, hist, sizeof(uint32_t) * 288);
        memcpy(fd, hist + 288, sizeof(uint32_t) * 32);
        {   /* at least two distance codes so every decoder accepts the set */
            int cntd = 0;
            for (int i = 0; i < 30; i++) cntd += fd[i] != 0;
            if (cntd < 2 && fd[0] == 0) { fd[0] = 1; cntd++; }
            if (cntd < 2) fd[1] = 1;
        }
        huff_lengths(fl, 286, ZA_LIMIT_L, lens);
        huff_lengths(fd, 30, ZA_LIMIT_D, lens + 288);
        canon_codes(lens, 286, codes);
        canon_codes(lens + 288, 30, codes + 288);
        hlit = 286; while (hlit > 257 && lens[hlit - 1] == 0) hlit--;
        hdist = 30; while (hdist > 1 && lens[288 + hdist - 1] == 0) hdist--;
        uint8_t seq[320];
        memcpy(seq, lens, (size_t)hlit);
        memcpy(seq + hlit, lens + 288, (size_t)hdist);
        ncltok = rle_lengths(seq, hlit + hdist, cltok);
        uint32_t clf[19];
        memset(clf, 0, sizeof clf);
        for (int i = 0; i < ncltok; i++) clf[cltok[i] & 0xFF]++;
        huff_lengths(clf, 19, 7, cl_lens);
        canon_codes(cl_lens, 19, cl_codes);
        hclen = 19; while (hclen > 4 && cl_lens[cl_order[hclen - 1]] == 0) hclen--;
        /* exact costs in bits */
        uint64_t data_dyn = 0, data_fix = 0;
        for (int i = 0; i < 286; i++) {
            uint32_t f = hist[i];
            if (!f) continue;
            int ex = i >= 257 ? len_extra[i - 257] : 0;
            int fx = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            data_dyn += (uint64_t)f * (uint64_t)(lens[i] + ex);
            data_fix += (uint64_t)f * (uint64_t)(fx + ex);
        }
        for (int i = 0; i < 30; i++) {
            uint32_t f = hist[288 + i];
            data_dyn += (uint64_t)f * (uint64_t)(lens[288 + i] + dist_extra[i]);
            data_fix += (uint64_t)f * (uint64_t)(5 + dist_extra[i]);
        }
        uint64_t hdr_dyn = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
        for (int i = 0; i < ncltok; i++) {
            int s = cltok[i] & 0xFF;
            hdr_dyn += cl_lens[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0);
        }
        if (flat) hdr_dyn = 3 + 5 + 5 + 4 + 3 * 19 + 4 * (uint64_t)(hlit + hdist);
        uint64_t cost_dyn = hdr_dyn + data_dyn, cost_fix = 3 + data_fix;
        uint64_t nchunks = ((uint64_t)n + 65534) / 65535;
        uint64_t cost_sto = 8 * ((uint64_t)n + 5 * nchunks);
        uint64_t bestc = cost_dyn; btype = 2;
        if (cost_fix <= bestc) { bestc = cost_fix; btype = 1; }
        if (cost_sto <= bestc) { bestc = cost_sto; btype = 0; }
        if (btype == 1) {
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            for (i = 0; i < 32; i++) lens[288 + i] = 5;
            canon_codes(lens, 288, codes);
            canon_codes(lens + 288, 30, codes + 288);
        }
        use_stored = (btype == 0);
    }
    if (dbg && dbg->btype) *dbg->btype = btype;
    if (dbg && dbg->lens) memcpy(dbg->lens, lens, sizeof lens);

    if (use_stored) {
        int off = 0;
        while (off < n) {
            int len = n - off > 65535 ? 65535 : n - off;
            int last = final && (off + len == n);
            putbits(&w, (uint32_t)last, 3); flushbyte(&w);
            putbits(&w, (uint32_t)len, 16); putbits(&w, (uint32_t)len ^ 0xFFFFu, 16);
            for (int i = 0; i < len; i++) putbits(&w, data[off + i], 8);
            off += len;
        }
    } else {
        putbits(&w, (uint32_t)final | ((uint32_t)btype << 1), 3);
        if (btype == 2 && flat) {
            /* flat header: code-length code = symbols 0..15 at 4 bits each (16, 17, 18 unused), so code(s) = s;
             * every code length is one 4-bit field at a fixed offset */
            putbits(&w, (uint32_t)(hlit - 257), 5);
            putbits(&w, (uint32_t)(hdist - 1), 5);
            putbits(&w, 15, 4);
            for (int i = 0; i < 19; i++) putbits(&w, cl_order[i] < 16 ? 4u : 0u, 3);
            for (int i = 0; i < hlit + hdist; i++) {
                uint32_t v = i < hlit ? lens[i] : lens[288 + i - hlit];
                putbits(&w, ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3), 4);
            }
        } else if (btype == 2) {
            putbits(&w, (uint32_t)(hlit - 257), 5);
            putbits(&w, (uint32_t)(hdist - 1), 5);
            putbits(&w, (uint32_t)(hclen - 4), 4);
            for (int i = 0; i < hclen; i++) putbits(&w, cl_lens[cl_order[i]], 3);
            for (int i = 0; i < ncltok; i++) {
                int s = cltok[i] & 0xFF, ex = cltok[i] >> 8;
                putbits(&w, cl_codes[s], cl_lens[s]);
                if (s == 16) putbits(&w, (uint32_t)ex, 2);
                else if (s == 17) putbits(&w, (uint32_t)ex, 3);
                else if (s == 18) putbits(&w, (uint32_t)ex, 7);
            }
        }
        for (int s = 0; s < nseg; s++) {
            const uint32_t *t = tokens + (size_t)s * seg;
            seg_bits[s] = (uint32_t)bitpos(&w);
            int op = s * seg, oend = op + seg, nextb = op;               /* output position, next boundary to index (the index's grain is the segment) */
            if (oend > n) oend = n;
            for (uint32_t k = 0; k < seg_ntok[s]; k++) {
                uint32_t tk = t[k];
                for (; nextb <= op; nextb += seg)
                    chunk_idx[nextb >> sshift] = (uint32_t)bitpos(&w) | ((uint32_t)(op - nextb) << 23);
                if (tk & 0x80000000u) {
                    int lc = (int)((tk >> 26) & 31), dc = (int)((tk >> 16) & 31);
                    putbits(&w, codes[257 + lc], lens[257 + lc]);
                    putbits(&w, (tk >> 21) & 31u, len_extra[lc]);
                    putbits(&w, codes[288 + dc], lens[288 + dc]);
                    putbits(&w, tk & 0x1FFFu, dist_extra[dc]);
                    op += len_base[lc] + (int)((tk >> 21) & 31u);
                } else {                                  /* one to three literals */
                    int nl = (int)((tk >> 24) & 3u) + 1;
                    for (int i = 0; i < nl; i++) { uint32_t by = (tk >> (8 * i)) & 0xFFu; putbits(&w, codes[by], lens[by]); }
                    op += nl;
                }
            }
            for (; nextb < oend; nextb += seg)                          /* boundaries behind the last token start */
                chunk_idx[nextb >> sshift] = (uint32_t)bitpos(&w) | ((uint32_t)(oend - nextb) << 23);
        }
        seg_bits[nseg] = (uint32_t)bitpos(&w);
        chunk_idx[nseg] = (uint32_t)bitpos(&w);
        putbits(&w, codes[256], lens[256]);
    }
    if (final) flushbyte(&w);
    else { putbits(&w, 0, 3); flushbyte(&w); putbits(&w, 0, 16); putbits(&w, 0xFFFF, 16); }
    if (dbg && dbg->seg_bits) memcpy(dbg->seg_bits, seg_bits, sizeof seg_bits);
    if (dbg && dbg->chunk_idx) memcpy(dbg->chunk_idx, chunk_idx, sizeof chunk_idx);
    free(prevdist); free(linkB); free(linkC); free(best); free(tokens);
    return w.overflow ? ZA_BUF_ERROR : (long)w.pos;
}

long za_o_deflate_stream(const uint8_t *data, size_t n, int level, int flags, uint8_t *out, size_t cap)
{
    size_t off = 0, op = 0;
    /* a stream of up to one unit's size is cut into units of 16 KiB (the product: zngamd_deflate_stream sets ZNGAMD_FLAG_UNITS16K --
     * latency before size for the small one-shot calls) */
    const size_t U = n <= ZA_MAX_UNIT ? ZA_SMALL_UNIT : ZA_MAX_UNIT;
    if (n == 0) return za_o_deflate_unit(data, 0, 0, level, flags, out, cap, NULL, NULL);
    while (off < n) {
        size_t len = n - off > U ? U : n - off;
        int dict = off > ZA_WIN ? ZA_WIN : (int)off;
        int f = (off + len == n) ? flags : 0;
        long r = za_o_deflate_unit(data + off, dict, (int)len, level, f, out + op, cap - op, NULL, NULL);
        if (r < 0) return r;
        op += (size_t)r; off += len;
    }
    return (long)op;
}

/* ---------------- cpu_baseline helper (bench.py only) ---------------- */
typedef struct {
    const uint8_t *data; size_t n, block; int level, tid, nthreads;
    uint8_t **comp; size_t *comp_len; int phase; int err;
} bench_arg;

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *bench_worker(void *vp)
{
    bench_arg *a = (bench_arg *)vp;
    size_t nblocks = (a->n + a->block - 1) / a->block;
    uint8_t *scratch = (uint8_t *)malloc(a->block + 64);
    for (size_t b = (size_t)a->tid; b < nblocks; b += (size_t)a->nthreads) {
        size_t off = b * a->block, len = a->n - off > a->block ? a->block : a->n - off;
        if (a->phase == 0) {
            size_t cap = len + len / 10 + 512, op = 0, uo = 0;
            uint8_t *dst = (uint8_t *)malloc(cap);
            while (uo < len) {     /* reference block -> units, dictionary = previous 32 KiB of input */
                size_t ul = len - uo > ZA_MAX_UNIT ? ZA_MAX_UNIT : len - uo;
                size_t abs = off + uo;
                int dict = abs > ZA_WIN ? ZA_WIN : (int)abs;
                uint32_t crc;
                long r = za_o_deflate_unit(a->data + abs, dict, (int)ul, a->level, 0, dst + op, cap - op, &crc, NULL);
                if (r < 0) { a->err = (int)r; break; }
                op += (size_t)r; uo += ul;
            }
            a->comp[b] = dst; a->comp_len[b] = op;
        } else {
            size_t used, got;
            int dict = off > ZA_WIN ? ZA_WIN : (int)off;
            int r = za_o_inflate_raw(a->comp[b], a->comp_len[b], scratch, a->block + 64,
                                     a->data + off - dict, (size_t)dict, &used, &got);
            /* stream ends on a sync flush, so the decoder runs out of input: BUF_ERROR is expected */
            if ((r != ZA_BUF_ERROR && r != ZA_STREAM_END) || got != len || memcmp(scratch, a->data + off, len)) a->err = -99;
        }
    }
    free(scratch);
    return NULL;
}

int za_o_bench_blocks(const uint8_t *data, size_t n, size_t block, int level, int threads,
                      double *t_deflate, double *t_inflate, size_t *comp_bytes)
{
    size_t nblocks = (n + block - 1) / block;
    uint8_t **comp = (uint8_t **)calloc(nblocks, sizeof *comp);
    size_t *clen = (size_t *)calloc(nblocks, sizeof *clen);
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    bench_arg *args = (bench_arg *)calloc((size_t)threads, sizeof *args);
    int err = 0;
    for (int phase = 0; phase < 2; phase++) {
        double t0 = now_s();
        for (int t = 0; t < threads; t++) {
            args[t] = (bench_arg){ data, n, block, level, t, threads, comp, clen, phase, 0 };
            pthread_create(&th[t], NULL, bench_worker, &args[t]);
        }
        for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (args[t].err) err = args[t].err; }
        double dt = now_s() - t0;
        if (phase == 0) *t_deflate = dt; else *t_inflate = dt;
    }
    size_t tot = 0;
    for (size_t b = 0; b < nblocks; b++) { tot += clen[b]; free(comp[b]); }
    *comp_bytes = tot;
    free(comp); free(clen); free(th); free(args);
    return err;
}
