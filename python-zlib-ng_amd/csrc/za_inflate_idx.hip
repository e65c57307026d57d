// Pass 2 of the two-pass inflate of indexed gzip members ('Z','A' FEXTRA subfield, version 2).  Product code.
//
// Replaces, for one member per workgroup, the DEFLATE / TRAILER states of GzipReader_read_into_buffer
// (reference src/zlib_ng/zlib_ngmodule.c:2539 zng_inflate, :2556 zng_crc32_z, :2577-2599 CRC32 / ISIZE check).
//
// One 512-thread workgroup per member (<= 128 KiB of output).  The member's whole output lives in LDS while it is
// built; HBM sees the compressed bytes once (16-byte loads) and the output once (coalesced 16-byte stores):
//   setup    the index (one entry per 256 bytes of output: bit offset of the first token that starts at or behind
//            that byte, and how far behind) and the block header are read by all threads.  Members written by this
//            engine carry the dynamic header in its flat form (every code length is a 4-bit field at a known
//            offset), so the tables are built without a serial step: symbol ranks from wave ballots, one table
//            entry per thread and step.  Codes are at most 11 (literal/length) and 9 (distance) bits long, so one
//            table read decodes any symbol; the entries carry base and extra-bit count.
//   phase A  thread c decodes the tokens that start in chunk c from a 64-bit register bit buffer fed by 16-byte
//            global loads (two in flight).  Literals go to LDS; a match leaves a 4-byte record (distance, length,
//            gap to the next match of the chunk) at its destination -- the place it will fill later.
//   phase B  every thread resolves the matches of its chunk in order, 16 bytes per step, LDS to LDS.  A bitmap with
//            one bit per output byte says which bytes are final (phase A sets the literals' bits); a piece is copied as
//            soon as the bits of its source are set, and sets its own.  A match therefore waits for exactly the bytes it
//            reads (the data dependencies of LZ77 are about a hundred levels deep in a 128 KiB member of text; waiting for
//            "everything below the source" instead made the resolution almost sequential).  The lowest pending piece of
//            the member is always ready, so the loop ends.
//   phase C  CRC-32 of the output (slice-by-4 over 256-byte blocks, folded with GF(2) products), compared with the
//            trailer together with ISIZE; the output leaves LDS in 16-byte stores, 8 KiB per step of the workgroup.
// Every chunk must end exactly on the next chunk's first token and the last one on the end-of-block code and the
// member's last byte, so an accepted member is the unique RFC 1951 decode of its stream; anything else (foreign
// header form, stored blocks, an index that does not fit) is reported as ZA_I_INDEX and the caller decodes the
// member with the sequential decoder, which also produces the error verdicts.
#include "za_common.h"

#define ZA_IDX_THREADS 512
#define ZA_IDX_NONE    0xFFFFFFFFu

struct __attribute__((aligned(16))) ZaIdxLds {
    uint8_t out[ZA_MAX_UNIT + 64];             // output byte i at out[shift + i], shift = destination address & 15
    uint16_t lut_l[1 << ZA_LIMIT_L];           // phase C: the four CRC slice tables (4 KiB)
    uint32_t lut_d[1 << ZA_LIMIT_D];
    uint32_t idx[ZA_MAX_CHUNKS + 1];           // index entries as stored: bit offset | overshoot << 23
    uint32_t fin[ZA_MAX_UNIT / 32 + 2];        // one bit per output byte: the byte is final
    uint16_t sym[2][288];                      // symbols in canonical order
    uint32_t cnt[2][16], first[2][16], offs[2][16];
    uint32_t wcnt[6][16];
    uint8_t lens[320];
    uint32_t crcpart[8];
    int err;
};

__device__ __forceinline__ uint32_t za_rev4(uint32_t v) { return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); }

struct __attribute__((aligned(4))) ZaU4 { uint32_t x, y, z, w; };

// 16 bytes at a dword-aligned address; bytes at or behind `lim` read as zero
__device__ __forceinline__ ZaU4 za_idx_load16(const uint8_t *p, const uint8_t *lim)
{
    ZaU4 v;
    if (p + 16 <= lim) v = *(const ZaU4 *)p;
    else {
        uint32_t t[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; k++) if (p + k < lim) t[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
        v.x = t[0]; v.y = t[1]; v.z = t[2]; v.w = t[3];
    }
    return v;
}

#define ZA_IDX_ROUNDS 2      // token rounds between two top-ups of the input queue

// measurement build (-DZA_IDX_STATS, profiles/idx_stats.sh): time per phase, summed over the members (100 MHz ticks)
#ifdef ZA_IDX_STATS
__device__ unsigned long long za_idx_stat[8];
#define ZA_IDX_T(var) const unsigned long long var = wall_clock64()
#define ZA_IDX_ADD(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&za_idx_stat[i], (b) - (a)); } while (0)
#else
#define ZA_IDX_T(var) do { } while (0)
#define ZA_IDX_ADD(i, a, b) do { } while (0)
#endif

__global__ __launch_bounds__(ZA_IDX_THREADS) void za_k_inflate_indexed(const uint8_t *__restrict__ in, uint64_t in_total,
                                                                        const ZaMember *__restrict__ members,
                                                                        uint8_t *__restrict__ out, uint64_t out_cap,
                                                                        const uint32_t *__restrict__ crc_slice4,   // [4][256]
                                                                        const uint32_t *__restrict__ x256_table,   // [512] x^(8*256*k)
                                                                        const uint32_t *__restrict__ x8_table,     // [257] x^(8*k)
                                                                        int32_t *__restrict__ status_out)
{
    __shared__ ZaIdxLds S;
    ZA_IDX_T(t_start);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ZaMember m = members[blockIdx.x];
    const uint32_t n = m.out_len, nchunk = m.nseg;
    // ---- member table entry (uniform) ----
    if (m.in_off + m.in_len + 8 > in_total || m.out_off + (uint64_t)n > out_cap || n == 0 || n > ZA_MAX_UNIT ||
        nchunk != ((n + (1u << ZA_CHUNK_SHIFT) - 1) >> ZA_CHUNK_SHIFT) || m.index_off != 4u * (nchunk + 1u) || m.index_off > m.in_off ||
        m.in_len < 1 || m.in_len > (1u << 20)) {
        if (tid == 0) status_out[blockIdx.x] = ZA_I_INDEX;
        return;
    }
    const uint8_t *src = in + m.in_off;
    const uint8_t *lim = in + in_total;
    const uint32_t in_bits = (uint32_t)m.in_len * 8u;
    uint8_t *dst = out + m.out_off;
    const uint32_t shift = (uint32_t)((uintptr_t)dst & 15u);
    uint8_t *ob = S.out + shift;
    // ---- block header (uniform): BFINAL, BTYPE, and for a dynamic block the flat header form ----
    const uint64_t h0 = za_peek(src, 0);
    const int last = (int)(h0 & 1u), type = (int)((h0 >> 1) & 3u);
    if (!last || type == 0 || type == 3) { if (tid == 0) status_out[blockIdx.x] = ZA_I_INDEX; return; }
    uint32_t nlen = 288, ndist = 30, hdr_end = 3;
    if (type == 2) {
        nlen = (uint32_t)((h0 >> 3) & 31u) + 257u; ndist = (uint32_t)((h0 >> 8) & 31u) + 1u;
        const uint32_t hclen = (uint32_t)((h0 >> 13) & 15u);
        // code-length code lengths in the order 16 17 18 0 8 7 ...: the flat form has 0 0 0 and then sixteen 4s
        uint64_t want = 0;
        for (int i = 3; i < 19; i++) want |= 4ull << (3 * i);
        const uint64_t got = za_peek(src, 17) & ((1ull << 57) - 1ull);
        hdr_end = 74u + 4u * (nlen + ndist);
        if (nlen > 286 || ndist > 30 || hclen != 15 || got != want || hdr_end > in_bits) {
            if (tid == 0) status_out[blockIdx.x] = ZA_I_INDEX;
            return;
        }
    }
    // ---- index, code lengths ----
    const uint8_t *ixp = src - m.index_off;
    for (uint32_t c = (uint32_t)tid; c <= nchunk; c += ZA_IDX_THREADS) {
        const uint32_t e = za_ld32(ixp + 4u * c);
        S.idx[c] = e;
    }
    for (uint32_t i = (uint32_t)tid; i < ZA_MAX_UNIT / 32 + 2; i += ZA_IDX_THREADS) S.fin[i] = 0;
    if (tid == 0) S.err = 0;
    if (tid < 320) {
        uint32_t v = 0;
        if (type == 1) v = tid < 144 ? 8u : tid < 256 ? 9u : tid < 280 ? 7u : tid < 288 ? 8u : tid < 318 ? 5u : 0u;
        else {
            const bool isl = (uint32_t)tid < nlen, isd = tid >= 288 && (uint32_t)(tid - 288) < ndist;
            if (isl || isd) {
                const uint32_t k = isl ? (uint32_t)tid : nlen + (uint32_t)(tid - 288);
                v = za_rev4((uint32_t)(za_peek(src, 74u + 4u * k) & 15u));
            }
        }
        S.lens[tid] = (uint8_t)v;
    }
    if (tid < 96) ((uint32_t *)S.wcnt)[tid] = 0;
    __syncthreads();
    // ---- tables.  Waves 0..4 hold the 320 literal/length slots (symbol = tid), wave 5 the distance symbols. ----
    {
        const int tab = wave < 5 ? 0 : 1;
        uint32_t mylen = 0, mysym = 0;
        if (wave < 5) { mysym = (uint32_t)tid; mylen = tid < 288 ? S.lens[tid] : 0u; }
        else if (wave == 5) { mysym = (uint32_t)lane; mylen = lane < 32 ? S.lens[288 + lane] : 0u; }
        const uint32_t maxl = tab ? ZA_LIMIT_D : ZA_LIMIT_L;
        uint32_t rank = 0, wc = 0;
        bool toolong = false;
        if (wave < 6) {
            toolong = mylen > maxl;
            for (uint32_t l = 1; l <= ZA_LIMIT_L; l++) {
                const unsigned long long mk = __ballot(mylen == l);
                if (mylen == l) rank = (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                if ((uint32_t)lane == l) wc = (uint32_t)__popcll(mk);
            }
            if (lane >= 1 && lane <= ZA_LIMIT_L) S.wcnt[wave][lane] = wc;
        }
        if (__ballot(toolong) != 0ull && lane == 0) S.err = ZA_I_INDEX;      // a code this table cannot hold: sequential decoder
        __syncthreads();
        if (tid < 32) {
            const int t = tid >> 4, l = tid & 15;
            uint32_t s = 0;
            if (l >= 1 && l <= ZA_LIMIT_L) { if (t == 0) for (int w = 0; w < 5; w++) s += S.wcnt[w][l]; else s = S.wcnt[5][l]; }
            S.cnt[t][l] = s;
        }
        __syncthreads();
        if (tid < 2) {
            uint32_t code = 0, off = 0;
            int left = 1;
            S.first[tid][0] = 0; S.offs[tid][0] = 0;
            uint32_t total = 0;
            for (int l = 1; l <= 15; l++) {
                const uint32_t cl = (l <= ZA_LIMIT_L) ? S.cnt[tid][l] : 0u, cprev = (l >= 2) ? S.cnt[tid][l - 1] : 0u;
                code = (code + cprev) << 1;
                S.first[tid][l] = code; S.offs[tid][l] = off; off += cl; total += cl;
                left = (left << 1) - (int)cl;
                if (left < 0) break;
            }
            // a complete code (the fixed block's 30 distance codes of 5 bits are the one accepted exception);
            // anything else is for the sequential decoder to judge
            if (left < 0 || total == 0 || (left != 0 && !(type == 1 && tid == 1))) S.err = ZA_I_INDEX;
        }
        __syncthreads();
        if (wave < 6 && mylen != 0 && mylen <= maxl) {
            uint32_t before = 0;
            if (tab == 0) for (int w = 0; w < wave; w++) before += S.wcnt[w][mylen];
            S.sym[tab][S.offs[tab][mylen] + before + rank] = (uint16_t)mysym;
        }
        __syncthreads();
        if (S.err != 0) { if (tid == 0) status_out[blockIdx.x] = S.err; return; }
        // one table entry per thread and step: canonical decode of the entry's low bits
        for (uint32_t e = (uint32_t)tid; e < (1u << ZA_LIMIT_L); e += ZA_IDX_THREADS) {
            uint32_t code = 0, r = 0;
            for (uint32_t l = 1; l <= ZA_LIMIT_L; l++) {
                code = (code << 1) | ((e >> (l - 1)) & 1u);
                const uint32_t k = code - S.first[0][l];
                if (r == 0 && k < S.cnt[0][l]) {
                    const uint32_t s = S.sym[0][S.offs[0][l] + k];
                    if (s < 256) r = (s << 4) | l;
                    else if (s == 256) r = 0xF000u | l;
                    else if (s < 286) { int nx; const int base = za_len_base((int)s - 257, nx); r = 0x8000u | ((uint32_t)nx << 12) | ((uint32_t)(base - 3) << 4) | l; }
                    else r = 0x10000u;        // 286 / 287: never valid
                }
            }
            S.lut_l[e] = (uint16_t)r;         // (0x10000 stores 0 = invalid)
        }
        {
            const uint32_t e = (uint32_t)tid;       // 512 entries, 512 threads
            uint32_t code = 0, r = 0;
            for (uint32_t l = 1; l <= ZA_LIMIT_D; l++) {
                code = (code << 1) | ((e >> (l - 1)) & 1u);
                const uint32_t k = code - S.first[1][l];
                if (r == 0 && k < S.cnt[1][l]) {
                    const uint32_t s = S.sym[1][S.offs[1][l] + k];
                    if (s < 30) { int nx; const int base = za_dist_base((int)s, nx); r = ((uint32_t)base << 8) | ((uint32_t)nx << 4) | l; }
                    else r = 0x80000000u;
                }
            }
            S.lut_d[e] = r == 0x80000000u ? 0u : r;
        }
    }
    __syncthreads();
    ZA_IDX_T(t_a);
    ZA_IDX_ADD(0, t_start, t_a);

    // ---- phase A ----
    const uint32_t c = (uint32_t)tid;
    uint32_t first_m = ZA_IDX_NONE;
    uint32_t end = 0;
    int lerr = 0;
    {
        bool act = c < nchunk;
        uint32_t pos = 0, sb = 0, eb = 0;
        if (act) {
            const uint32_t e0 = S.idx[c], e1 = S.idx[c + 1];
            pos = (c << ZA_CHUNK_SHIFT) + (e0 >> 23);
            end = c + 1 < nchunk ? ((c + 1) << ZA_CHUNK_SHIFT) + (e1 >> 23) : n;
            sb = e0 & 0x7FFFFFu; eb = e1 & 0x7FFFFFu;
            // the index itself: offsets ascend, a chunk starts at most 257 bytes behind its nominal start, the first at 0 / header end
            if (pos > end || end > n || (e0 >> 23) > 257u || sb > eb || eb > in_bits || (c == 0 && (pos != 0 || sb != hdr_end)) ||
                (pos == end && sb != eb)) { lerr = ZA_I_INDEX; act = false; }
            else if (pos == end) act = false;                      // a chunk no token starts in
        }
        // input queue: bit buffer, up to five dwords behind it, two 16-byte loads in flight behind those
        uint64_t bb = 0; int nb = 0;
        uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0; int qn = 0;
        ZaU4 nx = {0, 0, 0, 0}, ld = {0, 0, 0, 0};
        const uint8_t *lp = src;
        uint32_t bp = sb;                       // bit offset of the next token
        if (act) {
            const uint8_t *a = src + (sb >> 3);
            const uint8_t *a4 = (const uint8_t *)((uintptr_t)a & ~(uintptr_t)3);
            const uint32_t skip = (uint32_t)(a - a4) * 8u + (sb & 7u);        // < 32
            const ZaU4 f = za_idx_load16(a4, lim);
            nx = za_idx_load16(a4 + 16, lim);
            ld = za_idx_load16(a4 + 32, lim);
            lp = a4 + 48;
            bb = (uint64_t)(f.x >> skip); nb = 32 - (int)skip;
            q0 = f.y; q1 = f.z; q2 = f.w; qn = 3;
        }
        uint32_t prev_rec = ZA_IDX_NONE, prev_end = 0;
        const uint32_t pos0 = pos;
        uint32_t fw = pos >> 5, fbits = 0;          // bitmap word being filled with the bits of my literals
        for (;;) {
            if (__ballot(act) == 0ull) break;
            // top-up: a queue that is down to one dword takes the next 16 bytes; the load issued here is used two top-ups later
            if (act && qn <= 1) {
                if (qn == 0) { q0 = nx.x; q1 = nx.y; q2 = nx.z; q3 = nx.w; }
                else { q1 = nx.x; q2 = nx.y; q3 = nx.z; q4 = nx.w; }
                qn += 4;
                nx = ld;
                ld = za_idx_load16(lp, lim);
                lp += 16;
            }
#pragma unroll
            for (int r = 0; r < ZA_IDX_ROUNDS; r++) {
                if (nb <= 32 && qn > 0) {
                    bb |= (uint64_t)q0 << nb; nb += 32;
                    q0 = q1; q1 = q2; q2 = q3; q3 = q4; qn--;
                }
                // a token takes at most 11 + 5 bits (literal / length) and 9 + 13 (distance): with 38 bits in the buffer and
                // the queue it can be decoded (the buffer is refilled once more in front of the distance); a lane with less
                // waits for its next top-up
                if (act && nb + 32 * qn >= 38) {
                    const uint32_t e = S.lut_l[(uint32_t)bb & ((1u << ZA_LIMIT_L) - 1u)];
                    const uint32_t l = e & 15u;
                    if (!(e & 0x8000u)) {
                        if (e == 0u) { lerr = ZA_I_DATA; act = false; }
                        else {
                            ob[pos] = (uint8_t)(e >> 4);
                            if ((pos >> 5) != fw) { if (fbits) atomicOr(&S.fin[fw], fbits); fw = pos >> 5; fbits = 0; }
                            fbits |= 1u << (pos & 31u);
                            pos++; bb >>= l; nb -= (int)l; bp += l;
                        }
                    } else {
                        const uint32_t nxb = (e >> 12) & 7u;
                        const uint32_t len = ((e >> 4) & 0xFFu) + 3u + ((uint32_t)(bb >> l) & ((1u << nxb) - 1u));
                        uint32_t used = l + nxb;
                        bb >>= used; nb -= (int)used; bp += used;
                        if (nb <= 32 && qn > 0) {
                            bb |= (uint64_t)q0 << nb; nb += 32;
                            q0 = q1; q1 = q2; q2 = q3; q3 = q4; qn--;
                        }
                        used = 0;
                        const uint32_t d = S.lut_d[(uint32_t)(bb >> used) & ((1u << ZA_LIMIT_D) - 1u)];
                        const uint32_t dl = d & 15u, dnx = (d >> 4) & 15u;
                        const uint32_t dist = (d >> 8) + ((uint32_t)(bb >> (used + dl)) & ((1u << dnx) - 1u));
                        used += dl + dnx;
                        // end-of-block inside a chunk, a match this scheme cannot hold (length 3: no room for its record) or one
                        // that leaves the chunk: the index does not describe this stream; invalid codes / distances: data error
                        if (nxb == 7u || len < 4u || pos + len > end) { lerr = ZA_I_INDEX; act = false; }
                        else if (d == 0u || dist > pos) { lerr = ZA_I_DATA; act = false; }
                        else {
                            *(za_u32u *)(ob + pos) = (dist - 1u) | ((len - 3u) << 15) | 0xFF000000u;
                            if (prev_rec != ZA_IDX_NONE) ob[prev_rec + 3] = (uint8_t)(pos - prev_end);
                            else first_m = pos;
                            prev_rec = pos; pos += len; prev_end = pos;
                            bb >>= used; nb -= (int)used; bp += used;
                        }
                    }
                    if (pos >= end) act = false;
                }
            }
        }
        if (fbits) atomicOr(&S.fin[fw], fbits);
        if (c < nchunk && lerr == 0 && pos0 != end) {
            if (bp != eb) lerr = ZA_I_INDEX;             // must stop exactly on the next chunk's first token
        }
        if (c + 1 == nchunk && lerr == 0) {
            // behind the last chunk: the end-of-block code, then nothing but padding up to the member's last byte
            const uint32_t e = S.lut_l[(uint32_t)za_peek(src, bp) & ((1u << ZA_LIMIT_L) - 1u)];      // bp <= in_bits: inside the padded buffer
            if ((e & 0xF000u) != 0xF000u) lerr = ZA_I_INDEX;
            else if (((bp + (e & 15u) + 7u) >> 3) != (uint32_t)m.in_len) lerr = ZA_I_INDEX;
        }
        if (lerr) atomicMin(&S.err, lerr);               // data error (-3) outranks index (-7)? no: any failure sends the member to the sequential decoder
    }
    __syncthreads();
    if (S.err != 0) { if (tid == 0) status_out[blockIdx.x] = S.err; return; }
    ZA_IDX_T(t_b);
    ZA_IDX_ADD(1, t_a, t_b);
    // the CRC slice tables take the place of the literal/length table
    uint32_t *crct = (uint32_t *)S.lut_l;
    for (int i = tid; i < 1024; i += ZA_IDX_THREADS) crct[i] = crc_slice4[i];

    // ---- phase B ----
    {
        uint32_t mp = first_m;
        bool pend = mp != ZA_IDX_NONE;
        bool have = false;
        uint32_t mlen = 0, mdist = 1, gap = 0, done_b = 0, dd = 1;
        for (uint32_t guard = 0; guard < (1u << 22); guard++) {
            if (__ballot(pend) == 0ull) break;
            if (pend && !have) {
                const uint32_t rec = *(const za_u32u *)(ob + mp);
                mdist = (rec & 0x7FFFu) + 1u; mlen = ((rec >> 15) & 0xFFu) + 3u; gap = rec >> 24;
                have = true; done_b = 0; dd = mdist;
            }
            if (pend) {
                // the next piece: at most 16 bytes, and no more than the distance it is copied over (a self-overlapping match
                // copies from a multiple of its distance that is already there, so the pieces double)
                uint32_t piece = mlen - done_b;
                piece = piece < 16u ? piece : 16u;
                piece = piece < dd ? piece : dd;
                const uint32_t d0 = mp + done_b, s0 = d0 - dd;
                // are the source bytes final?  bits s0 .. s0 + piece - 1 of the bitmap (two dwords cover them)
                const uint32_t w0 = s0 >> 5, sh = s0 & 31u;
                const uint64_t two = ((uint64_t)*(volatile uint32_t *)&S.fin[w0 + 1u] << 32) | *(volatile uint32_t *)&S.fin[w0];
                const uint32_t mask = piece >= 32u ? 0xFFFFFFFFu : ((1u << piece) - 1u);
                if (((uint32_t)(two >> sh) & mask) == mask) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    uint8_t *d8 = ob + d0;
                    const uint8_t *s8 = ob + s0;
                    const uint32_t v0 = *(const za_u32u *)(s8), v1 = *(const za_u32u *)(s8 + 4), v2 = *(const za_u32u *)(s8 + 8), v3 = *(const za_u32u *)(s8 + 12);
                    if (piece >= 4u) *(za_u32u *)(d8) = v0;
                    if (piece >= 8u) *(za_u32u *)(d8 + 4) = v1;
                    if (piece >= 12u) *(za_u32u *)(d8 + 8) = v2;
                    if (piece >= 16u) *(za_u32u *)(d8 + 12) = v3;
                    const uint32_t k4 = piece & ~3u, t = piece & 3u;
                    const uint32_t tv = k4 == 0u ? v0 : k4 == 4u ? v1 : k4 == 8u ? v2 : v3;
                    if (t >= 1u) d8[k4] = (uint8_t)tv;
                    if (t >= 2u) d8[k4 + 1u] = (uint8_t)(tv >> 8);
                    if (t >= 3u) d8[k4 + 2u] = (uint8_t)(tv >> 16);
                    // the piece is final: its bits (they may straddle a dword)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    const uint32_t dw = d0 >> 5, dsh = d0 & 31u;
                    const uint64_t bits = (uint64_t)mask << dsh;
                    atomicOr(&S.fin[dw], (uint32_t)bits);
                    if ((uint32_t)(bits >> 32)) atomicOr(&S.fin[dw + 1u], (uint32_t)(bits >> 32));
                    done_b += piece;
                    if (dd + dd <= done_b + mdist) dd += dd;
                    if (done_b == mlen) {
                        mp = gap == 0xFFu ? ZA_IDX_NONE : mp + mlen + gap;
                        pend = mp != ZA_IDX_NONE; have = false;
                    }
                }
            }
        }
        if (pend) S.err = ZA_I_INDEX;       // (never: the lowest pending piece is always ready)
    }
    __syncthreads();
    if (S.err != 0) { if (tid == 0) status_out[blockIdx.x] = S.err; return; }

    ZA_IDX_T(t_c);
    ZA_IDX_ADD(2, t_b, t_c);
    // ---- phase C: CRC-32 / ISIZE against the trailer (zlib_ngmodule.c:2577-2599), output to HBM ----
    {
        const uint32_t nblk = nchunk;                       // 256-byte blocks
        const uint32_t b0 = (uint32_t)tid << 8;
        uint32_t cpart = 0;
        if ((uint32_t)tid < nblk) {
            uint32_t b1 = b0 + 256u; if (b1 > n) b1 = n;
            uint32_t r = 0xFFFFFFFFu, p = b0;
            for (; p + 4u <= b1; p += 4u) {
                r ^= *(const za_u32u *)(ob + p);
                r = crct[768 + (r & 0xFFu)] ^ crct[512 + ((r >> 8) & 0xFFu)] ^ crct[256 + ((r >> 16) & 0xFFu)] ^ crct[r >> 24];
            }
            for (; p < b1; p++) r = crct[(r ^ ob[p]) & 0xFFu] ^ (r >> 8);
            cpart = r ^ 0xFFFFFFFFu;
            // crc(A||B) = crc(A) x^(8|B|) ^ crc(B); |B| = (nblk-2-tid) whole blocks + the last block (folded in below)
            if ((uint32_t)tid + 1u < nblk) cpart = za_multmodp(x256_table[nblk - 2u - (uint32_t)tid], cpart);
        }
        const bool is_last = (uint32_t)tid + 1u == nblk;
        uint32_t lastc = is_last ? cpart : 0u;
        uint32_t rest = is_last ? 0u : cpart;
        rest = za_wave_xor_reduce(rest); lastc = za_wave_xor_reduce(lastc);
        if (lane == 0) { S.crcpart[wave] = rest; S.wcnt[0][wave] = lastc; }
        // output: 16-byte granules of the LDS image at their (aligned) place in memory
        uint8_t *g0 = dst - shift;
        const uint32_t lo = shift, hi = shift + n;            // valid bytes of the image
        for (uint32_t o = (uint32_t)tid * 16u; o < hi; o += ZA_IDX_THREADS * 16u) {
            if (o >= lo && o + 16u <= hi) *(uint4 *)(g0 + o) = *(const uint4 *)(S.out + o);
            else for (uint32_t k = 0; k < 16u; k++) if (o + k >= lo && o + k < hi) g0[o + k] = S.out[o + k];
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rs = 0, lc = 0;
            for (int w = 0; w < 8; w++) { rs ^= S.crcpart[w]; lc ^= S.wcnt[0][w]; }
            const uint32_t lastlen = n - ((nblk - 1u) << 8);
            const uint32_t crc = za_multmodp(x8_table[lastlen], rs) ^ lc;
            const uint32_t want_crc = za_ld32(src + m.in_len), want_len = za_ld32(src + m.in_len + 4);
            status_out[blockIdx.x] = (crc != want_crc) ? ZA_I_CRC : (want_len != n) ? ZA_I_LENGTH : ZA_I_OK;
        }
        ZA_IDX_T(t_end);
        ZA_IDX_ADD(3, t_c, t_end);
        ZA_IDX_ADD(4, t_start, t_end);
    }
}
