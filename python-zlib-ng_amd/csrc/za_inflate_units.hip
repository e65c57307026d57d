// The default writer's stream -- ONE deflate stream of dict-chained, sync-flushed units (gzip_ng_threaded.py:299-338) -- decoded with
// the writer's own segment index: the member decoder's lane-per-segment phase A and its in-order phase B, on 16-bit symbols, so
// that the units need not wait for each other (a source in front of a unit is a marker; the chunk pipeline's window kernels
// -- za_k_chunk_compose / _chain / _resolve -- turn markers into bytes).  Product code; included by zng_amd.hip behind za_inflate.hip.
//
// The unit's stream must be what this engine writes for full-size units (or with ZA_FLAG_FLATHDR for smaller ones): one block per
// unit -- stored blocks, fixed, or dynamic with the header in either form --, token boundaries at every 2 KiB of output, codes of at
// most 10 / 9 bits; the index = cidx of za_k_pack.  Anything else is reported (ZA_I_INDEX) and the caller decodes the stream
// without the index.
#pragma once

__global__ __launch_bounds__(64, 5) void za_k_inflate_units_marked(const uint8_t *__restrict__ in, uint64_t in_total,
                                                                   const ZaMember *__restrict__ members,     // per unit: in_off / in_len = its bytes in the stream (sync marker included), out_off = first symbol of its AREA in out16, out_len, crc = bytes of history in front of it (<= 32 768), nseg
                                                                   uint16_t *__restrict__ out16, uint64_t out_cap,     // symbols
                                                                   uint32_t *__restrict__ matchq,            // [grid][64][ZA_MATCHQ_PER_SEG]
                                                                   const uint32_t *__restrict__ ext_index,   // [grid][ZA_CIDX_STRIDE]: the units' segment indices (what za_k_pack left in cidx)
                                                                   ZaChunkRes *__restrict__ res_out)
{
    int32_t *status_out = nullptr; (void)status_out;
#define ZA_UM_FAIL(code) do { if (lane == 0) { ZaChunkRes r_; r_.status = (code); r_.max_back = 0; r_.bits = 0; r_.out_len = 0; res_out[blockIdx.x] = r_; } return; } while (0)
    __shared__ ZaMemTabs T;
    __shared__ int scratch[2];
    __shared__ __attribute__((aligned(16))) uint32_t rows[64 * ZA_IROW];      // table build: ZaMemBuild; phase A: staged input; then the CRC table
    static_assert(sizeof(ZaMemBuild) <= sizeof(uint32_t) * 64 * ZA_IROW, "build area");
    ZaMemBuild &B = *(ZaMemBuild *)rows;
    const int lane = za_lane();
    const ZaMember m = members[blockIdx.x];
    const uint8_t *src = in + m.in_off;
    const uint64_t in_bits = m.in_len * 8ull;
    // area coordinates: symbol 0 of the area is the first of the 32 768 marker symbols in front of the unit, so that no source
    // position is ever negative and a source in front of the unit is read like any other far source
    uint16_t *dst16 = out16 + m.out_off;
    const uint32_t hist = m.crc;                                      // (the field's role here)
    const int n = (int)m.out_len;
    const int nseg = (int)m.nseg;
    if (m.in_off + m.in_len > in_total || m.out_off + ZA_WIN + (uint64_t)m.out_len > out_cap || n > ZA_MAX_UNIT || m.in_len > (1u << 20) ||
        nseg != ((n + ZA_SEG - 1) >> ZA_SEG_SHIFT) || n == 0 || hist > (uint32_t)ZA_WIN) ZA_UM_FAIL(ZA_I_INDEX);
    const uint32_t *index = ext_index + (size_t)blockIdx.x * ZA_CIDX_STRIDE;
    if (index[nseg] == 0u) {
        // a unit of STORED blocks (what the packer writes for input that does not compress; its index is all zeros): blocks of at
        // most 65 535 bytes, each `BFINAL | 00`, LEN, ~LEN, bytes, on byte boundaries -- and the sync marker behind the last
        const uint32_t nblk = ((uint32_t)n + 65534u) / 65535u;
        uint32_t at = 0;
        bool ok = true, fin = false;
        for (uint32_t c = 0; c < nblk && ok; c++) {
            const uint32_t len = (uint32_t)n - 65535u * c > 65535u ? 65535u : (uint32_t)n - 65535u * c;
            if ((uint64_t)at + 5u + len > m.in_len) { ok = false; break; }
            const uint32_t h = src[at], l = za_ld16(src + at + 1), nl = za_ld16(src + at + 3);
            fin = (h & 1u) != 0u;
            ok = (h & 0xFEu) == 0u && l == len && nl == (~len & 0xFFFFu) && (!fin || c + 1 == nblk);
            const uint8_t *pb = src + at + 5;
            uint16_t *ps = dst16 + ZA_WIN + 65535u * c;
            const uint32_t full = len & ~7u;
            for (uint32_t i = 8u * (uint32_t)lane; i < full; i += 8u * 64u) {
                const ZaU2u w = *(const ZaU2u *)(pb + i);
                ZaU4u v;
                v.x = __builtin_amdgcn_perm(0u, w.x, 0x0C010C00u); v.y = __builtin_amdgcn_perm(0u, w.x, 0x0C030C02u);
                v.z = __builtin_amdgcn_perm(0u, w.y, 0x0C010C00u); v.w = __builtin_amdgcn_perm(0u, w.y, 0x0C030C02u);
                *(ZaU4u *)(ps + i) = v;
            }
            for (uint32_t i = full + (uint32_t)lane; i < len; i += 64) ps[i] = pb[i];
            at += 5u + len;
        }
        if (ok) ok = fin ? (uint64_t)at == m.in_len : ((uint64_t)at + 5u == m.in_len && src[at] == 0u && za_ld32(src + at + 1) == 0xFFFF0000u);
        if (!ok) ZA_UM_FAIL(ZA_I_INDEX);
        if (lane == 0) { ZaChunkRes r; r.status = fin ? ZA_I_END : ZA_I_SYNC; r.max_back = 0; r.bits = (m.in_off + m.in_len) * 8ull; r.out_len = (uint64_t)n; res_out[blockIdx.x] = r; }
        return;
    }
    // index entries: bit offset | overshoot << 23; at this granularity (one entry per 2 KiB segment, where the codec forces a
    // token boundary) the overshoot is zero
    const uint32_t my_start = za_ld32((const uint8_t *)(index + (lane < nseg ? lane : nseg)));
    const uint32_t my_stop = za_ld32((const uint8_t *)(index + (lane < nseg ? lane + 1 : nseg)));
    if (__ballot((my_start >> 23) != 0u || (my_stop >> 23) != 0u) != 0ull || in_bits < 3) ZA_UM_FAIL(ZA_I_INDEX);
    // ---- block header (uniform).  Members written by this engine are one final block, fixed or dynamic with the header in its
    // flat form: HCLEN = 19, the code-length code is the fixed 4-bit code of the symbols 0..15, so code length k sits in the
    // 4 bits at 74 + 4 k (bit-reversed) and all lanes read the header at once.  Anything else: sequential decoder.
    const uint64_t bits = za_peek(src, 0);
    const int last = (int)(bits & 1u), type = (int)((bits >> 1) & 3u);
    uint32_t nlen = 288, ndist = 30, hdr_end = 3;
    bool hdr_ok = type == 1 || type == 2;                            // (a unit in mid-stream is not the last block; the stream's last one may be)
    bool flat = false;
    if (hdr_ok && type == 2) {
        nlen = (uint32_t)((bits >> 3) & 31u) + 257u; ndist = (uint32_t)((bits >> 8) & 31u) + 1u;
        uint64_t want = 0;
        for (int i = 3; i < 19; i++) want |= 4ull << (3 * i);
        hdr_ok = in_bits >= 74 && nlen <= 286 && ndist <= 30;
        flat = hdr_ok && ((bits >> 13) & 15u) == 15u && (za_peek(src, 17) & ((1ull << 57) - 1ull)) == want;
        if (flat) { hdr_end = 74u + 4u * (nlen + ndist); hdr_ok = hdr_end <= in_bits; }
    }
    if (!hdr_ok) ZA_UM_FAIL(ZA_I_INDEX);
    {
        bool toolong = false;
        if (type == 2 && !flat) {
            // The ordinary dynamic header (what the writer leaves unless asked for flat ones): HCLEN code-length code lengths, then
            // the run-length coded lengths, decoded by ONE lane -- about 300 dependent steps through the header's bytes, a twentieth
            // of what the unit's tokens cost its lanes -- into a byte per symbol; everything behind is the flat form's path.
            const uint32_t ncode = (uint32_t)((bits >> 13) & 15u) + 4u;
            if (17u + 3u * ncode > in_bits) ZA_UM_FAIL(ZA_I_INDEX);
            __syncthreads();
            if (lane < 19) B.lens[lane] = 0;
            __syncthreads();
            if (lane == 0) for (uint32_t i = 0; i < ncode; i++) B.lens[za_i_cl_order[i]] = (uint8_t)(za_peek(src, 17u + 3u * i) & 7u);
            if (za_build_table(B.lens, 19, B.cnt_d, B.sym_d, B.tmp_d, 7, &scratch[0], &scratch[1]) != 0) ZA_UM_FAIL(ZA_I_INDEX);       // (must be complete)
            __syncthreads();
            uint8_t *seq = (uint8_t *)B.sym_l;                         // 576 bytes: the nlen + ndist <= 316 lengths in stream order
            if (lane == 0) {
                uint64_t bp = 17u + 3u * ncode;
                uint32_t idx = 0, prev = 0;
                int err = 0;
                while (idx < nlen + ndist) {
                    if (bp + 15u > in_bits) { err = 1; break; }
                    const uint64_t b = za_peek(src, bp);
                    const uint32_t e = za_decode_sym(b, B.tmp_d, 7, B.cnt_d, B.sym_d);
                    if (!e) { err = 1; break; }
                    const uint32_t sy = e >> 4, l = e & 15u;
                    bp += l;
                    if (sy < 16u) { seq[idx++] = (uint8_t)sy; prev = sy; continue; }
                    uint32_t rep, val = 0;
                    const uint64_t x = b >> l;
                    if (sy == 16u) { if (idx == 0) { err = 1; break; } val = prev; rep = 3u + (uint32_t)(x & 3u); bp += 2; }
                    else if (sy == 17u) { rep = 3u + (uint32_t)(x & 7u); bp += 3; }
                    else { rep = 11u + (uint32_t)(x & 127u); bp += 7; }
                    if (idx + rep > nlen + ndist) { err = 1; break; }
                    while (rep--) seq[idx++] = (uint8_t)val;
                    prev = val;
                }
                scratch[0] = err; scratch[1] = (int)bp;
            }
            __syncthreads();
            if (scratch[0] != 0) ZA_UM_FAIL(ZA_I_INDEX);
            hdr_end = (uint32_t)scratch[1];
            __syncthreads();
            uint32_t vv[5];
#pragma unroll
            for (int b5 = 0; b5 < 5; b5++) {
                const int i = lane + 64 * b5;
                const bool isl = (uint32_t)i < nlen, isd = i >= 288 && (uint32_t)(i - 288) < ndist;
                vv[b5] = isl ? seq[i] : isd ? seq[nlen + (uint32_t)(i - 288)] : 0u;
            }
            __syncthreads();                                           // (the sequence lies where the tables' symbol lists go)
#pragma unroll
            for (int b5 = 0; b5 < 5; b5++) {
                const int i = lane + 64 * b5;
                toolong = toolong || vv[b5] > (i < 288 ? (uint32_t)ZA_ML_BITS : (uint32_t)ZA_MD_BITS);
                B.lens[i] = (uint8_t)vv[b5];
            }
        } else
        for (int i = lane; i < 320; i += 64) {
            uint32_t v = 0;
            if (type == 1) v = i < 144 ? 8u : i < 256 ? 9u : i < 280 ? 7u : i < 288 ? 8u : i < 318 ? 5u : 0u;
            else {
                const bool isl = (uint32_t)i < nlen, isd = i >= 288 && (uint32_t)(i - 288) < ndist;
                if (isl || isd) {
                    const uint32_t k = isl ? (uint32_t)i : nlen + (uint32_t)(i - 288);
                    const uint32_t f = (uint32_t)(za_peek(src, 74u + 4u * k) & 15u);
                    v = ((f & 1u) << 3) | ((f & 2u) << 1) | ((f & 4u) >> 1) | ((f & 8u) >> 3);
                }
            }
            toolong = toolong || v > (i < 288 ? (uint32_t)ZA_ML_BITS : (uint32_t)ZA_MD_BITS);
            B.lens[i] = (uint8_t)v;
        }
        if (__shfl(my_start, 0, 64) != hdr_end) ZA_UM_FAIL(ZA_I_INDEX);
        if (__ballot(toolong) != 0ull) ZA_UM_FAIL(ZA_I_INDEX);
        __syncthreads();
        // plain tables ((symbol << 4) | length) first -- the distance one in the row area -- then the entries are rewritten
        uint16_t *tmp_d = B.tmp_d;
        int ok = B.lens[256] != 0;
        int st = za_build_table(B.lens, (int)nlen, B.cnt_l, B.sym_l, T.lut_l, ZA_ML_BITS, &scratch[0], &scratch[1]);
        if (st < 0 || (st > 0 && scratch[1] != 1)) ok = 0;
        st = za_build_table(B.lens + 288, (int)(type == 1 ? 32u : ndist), B.cnt_d, B.sym_d, tmp_d, ZA_MD_BITS, &scratch[0], &scratch[1]);
        if (st < 0 || (st > 0 && scratch[1] != 1 && type != 1)) ok = 0;       // (the fixed block's 30 five-bit distance codes are incomplete by design)
        if (!ok) ZA_UM_FAIL(ZA_I_INDEX);
        for (int e = lane; e < (1 << ZA_ML_BITS); e += 64) {
            const uint32_t v = T.lut_l[e], s = v >> 4, l = v & 15u;
            uint32_t r = 0;
            if (l) {
                if (s < 256) r = (s << 4) | l;
                else if (s == 256) r = 0xF000u | l;
                else if (s < 286) { int nx; const int base = za_len_base((int)s - 257, nx); r = 0x8000u | ((uint32_t)nx << 12) | ((uint32_t)(base - 3) << 4) | l; }
            }
            T.lut_l[e] = (uint16_t)r;
        }
        for (int e = lane; e < (1 << ZA_MD_BITS); e += 64) {
            const uint32_t v = tmp_d[e], s = v >> 4, l = v & 15u;
            uint32_t r = 0;
            if (l && s < 30) { int nx; const int base = za_dist_base((int)s, nx); r = ((uint32_t)base << 8) | ((uint32_t)nx << 4) | l; }
            T.lut_d[e] = r;
        }
        __syncthreads();
    }

    // ---- phase A: every lane decodes its own segment into two compact streams: the segment's LITERAL BYTES, written to the
    // front of the segment's own 2 KiB of the output buffer (phase B expands them in place; nothing else lives there yet), and
    // one 4-byte QUEUE ENTRY per match.
    // A dependent 8-byte global load per token would cost microseconds, so each lane's compressed bytes are staged through an
    // LDS row: row r holds the 64 bytes at the lane's (16-byte aligned) origin + 48 r; a lane decodes while its read position is
    // inside the first 48 bytes of the row, and the next row (four aligned 16-byte loads) is already in flight in registers.
    uint32_t *myq = matchq + ((size_t)blockIdx.x * 64 + (size_t)lane) * ZA_MATCHQ_PER_SEG;
    uint32_t nmatch = 0;       // queue entries of my segment
    uint32_t nlit = 0;         // literal bytes of my segment
    int lane_err = 0;          // 0 ok, 1 index mismatch, 2 data error
    {
        uint32_t *myrow = rows + lane * ZA_IROW;
        const bool act = lane < nseg;
        int pos = lane << ZA_SEG_SHIFT;
        const int seg0 = pos;
        int end = pos + ZA_SEG; if (end > n) end = n;
        uint32_t bp = my_start;
        if (act && (my_stop > in_bits || my_stop < my_start)) lane_err = 1;
        // my stream starts in the byte at src + (my_start >> 3); rows start at the 16-byte aligned address below it
        const uint8_t *a0 = src + (my_start >> 3);
        const uint8_t *org = (const uint8_t *)((uintptr_t)a0 & ~(uintptr_t)15);
        const uint32_t org_bit = my_start - ((uint32_t)(a0 - org) * 8u + (my_start & 7u));      // bit offset (from src bit 0) of the origin; may be "negative" (wraps): only differences are used
        const uint8_t *lim = in + in_total;
#ifdef ZA_ABL_NO_A
        bool done = true;                                       // (instruction split only: nothing is decoded, the checks below fail)
#else
        bool done = !act || lane_err != 0 || pos >= end;
#endif
        // The literal bytes are collected in a 16-byte block (nlit & 15 bytes of it are taken) and leave as one 16-byte store
        // per block: 36 stores for the 570 literal bytes of an average 2 KiB of text.
        uint8_t *litp = (uint8_t *)(dst16 + ZA_WIN + seg0);          // the front of the segment's own 4 KiB of symbols
        uint64_t blk_lo = 0, blk_hi = 0;
        // Queue entries: distance - 1 | (length - 2) << 15 | (literals since the previous entry) << 24.  A run of literals is cut
        // into entries of its own (length field 0: the word is the count) before it exceeds 32, so that phase B moves every run
        // with two 16-byte copies.  Entries leave four at a time as one 16-byte store.
        uint32_t qb0 = 0, qb1 = 0, qb2 = 0;
        uint32_t gap = 0;                                       // literals since the previous entry: < 27 between two rounds
        auto push = [&](uint32_t ent) {
            const uint32_t k = nmatch & 3u;
            if (k == 3u) *(uint4 *)(myq + (nmatch & ~3u)) = make_uint4(qb0, qb1, qb2, ent);
            qb0 = k == 0u ? ent : qb0; qb1 = k == 1u ? ent : qb1; qb2 = k == 2u ? ent : qb2;
            nmatch++;
        };
        uint4 pre[ZA_IROW_LOADS];
        auto prefetch = [&](uint32_t r) {
#pragma unroll
            for (int j = 0; j < ZA_IROW_LOADS; j++) {
                const uint8_t *p = org + (size_t)ZA_IROW_BYTES * r + 16u * (unsigned)j;
                pre[j] = make_uint4(0, 0, 0, 0);
                if (!done && p >= in && p + 16 <= lim) pre[j] = *(const uint4 *)p;
            }
            if (!done && (org + (size_t)ZA_IROW_BYTES * r < in || org + (size_t)ZA_IROW_BYTES * r + 16 * ZA_IROW_LOADS > lim)) {
                // a row that reaches over an end of the caller's buffer (first / last member only): byte by byte
                uint32_t t[4 * ZA_IROW_LOADS];
#pragma unroll 1
                for (int k = 0; k < 4 * ZA_IROW_LOADS; k++) {
                    uint32_t v = 0;
                    for (int q = 0; q < 4; q++) { const uint8_t *p = org + (size_t)ZA_IROW_BYTES * r + 4u * (unsigned)k + (unsigned)q; if (p >= in && p < lim) v |= (uint32_t)*p << (8 * q); }
                    t[k] = v;
                }
#pragma unroll
                for (int j = 0; j < ZA_IROW_LOADS; j++) pre[j] = make_uint4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
            }
        };
        prefetch(0);
#pragma unroll 1
        for (uint32_t r = 0;; r++) {
            if (__ballot(!done) == 0ull) break;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < ZA_IROW_LOADS; j++) { myrow[4 * j] = pre[j].x; myrow[4 * j + 1] = pre[j].y; myrow[4 * j + 2] = pre[j].z; myrow[4 * j + 3] = pre[j].w; }
            __builtin_amdgcn_wave_barrier();
            prefetch(r + 1);
            const uint32_t row_bit0 = org_bit + (uint32_t)ZA_IROW_BYTES * 8u * r;
            // One round per lane = up to six literals AND the match behind them, on one straight path: text at level 6 is runs
            // of 3.8 literals between matches, so most rounds take a whole run with its match and no lane waits in a branch the
            // others do not take.  No break / continue inside (the compiler otherwise copies the whole lane state at every edge).
#pragma unroll 1
            for (;;) {
                const uint32_t rel = bp - row_bit0;                 // < 384 while inside the row's first 48 bytes
                const bool go = !done && rel < (uint32_t)ZA_IROW_BYTES * 8u;
                if (__ballot(go) == 0ull) break;
                if (go) {
                    const uint32_t w = rel >> 5, sh = rel & 31u;
                    // 128 bits starting at bit `rel` of the row (5 dwords): six literals take at most 60, a match 10 + 5 + 9 + 13
                    const uint32_t d0 = myrow[w], d1 = myrow[w + 1], d2 = myrow[w + 2], d3 = myrow[w + 3], d4 = myrow[w + 4];
                    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
                    const uint32_t h2 = __builtin_amdgcn_alignbit(d3, d2, sh), h3 = __builtin_amdgcn_alignbit(d4, d3, sh);
                    const int room = end - pos;                       // >= 1
                    // -- up to six literals, three at a time: each code is looked up in the 32 bits at hand (bit offsets <= 20),
                    // then the window moves on by what the three took.  (Runs of literals are long-tailed -- 15 % of them are longer
                    // than nine -- and the lanes with the long runs are the ones the whole wave waits for.)
                    auto lit3 = [&](uint32_t win, int left, bool on, uint32_t &bits, uint32_t &bytes) -> uint32_t {
                        const uint32_t e0 = T.lut_l[win & ((1u << ZA_ML_BITS) - 1u)];
                        const bool l0 = on && e0 != 0u && e0 < 0x8000u && left > 0;
                        uint32_t u = l0 ? (e0 & 15u) : 0u;
                        uint32_t g = l0 ? (e0 >> 4) : 0u;
                        const uint32_t e1 = T.lut_l[__builtin_amdgcn_ubfe(win, u, ZA_ML_BITS)];
                        const bool l1 = l0 && e1 != 0u && e1 < 0x8000u && left > 1;
                        u += l1 ? (e1 & 15u) : 0u;
                        g |= l1 ? (e1 >> 4) << 8 : 0u;
                        const uint32_t e2 = T.lut_l[__builtin_amdgcn_ubfe(win, u, ZA_ML_BITS)];
                        const bool l2 = l1 && e2 != 0u && e2 < 0x8000u && left > 2;
                        u += l2 ? (e2 & 15u) : 0u;
                        g |= l2 ? (e2 >> 4) << 16 : 0u;
                        bits = u; bytes = g;
                        return (l0 ? 1u : 0u) + (l1 ? 1u : 0u) + (l2 ? 1u : 0u);
                    };
                    uint32_t u1, g1, u2, g2b;
                    const uint32_t n1 = lit3(lo, room, true, u1, g1);
                    const uint32_t lo1 = __builtin_amdgcn_alignbit(hi, lo, u1), hi1 = __builtin_amdgcn_alignbit(h2, hi, u1), h21 = __builtin_amdgcn_alignbit(h3, h2, u1);
#ifdef ZA_ABL_ONE_LIT3
                    const uint32_t n2 = 0; u2 = 0; g2b = 0;                  // (experiment: three literals per round at most)
#else
                    const uint32_t n2 = lit3(lo1, room - 3, n1 == 3u, u2, g2b);
#endif
                    const uint32_t nl = n1 + n2, u = u1 + u2;
                    const uint64_t grp = (uint64_t)g1 | ((uint64_t)g2b << 24);
                    // -- the token behind them (64 bits from there on)
                    const uint32_t m_lo = __builtin_amdgcn_alignbit(hi1, lo1, u2), m_hi = __builtin_amdgcn_alignbit(h21, hi1, u2);
                    const uint32_t em = T.lut_l[m_lo & ((1u << ZA_ML_BITS) - 1u)];
                    const uint32_t l = em & 15u, nxb = (em >> 12) & 7u;
                    const uint32_t len = ((em >> 4) & 0xFFu) + 3u + __builtin_amdgcn_ubfe(m_lo, l, nxb);
                    const uint32_t used = l + nxb;                                    // <= 15
                    const uint32_t d = T.lut_d[__builtin_amdgcn_ubfe(m_lo, used, ZA_MD_BITS)];
                    const uint32_t dl = d & 15u, dnx = (d >> 4) & 15u;
                    const uint32_t off2 = used + dl;                                  // <= 24
                    const uint32_t dist = (d >> 8) + __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(m_hi, m_lo, off2), 0u, dnx);
                    const int pos1 = pos + (int)nl;
                    // a match is due unless the segment ends behind the literals or a fourth literal follows
                    const bool want = pos1 < end && !(em != 0u && em < 0x8000u);
                    const bool bad_data = em == 0u || ((em & 0x7000u) != 0x7000u && (d == 0u || dist > (uint32_t)pos1 + hist));
                    // end of block inside a segment, a match across the segment end, queue full
                    const bool bad_index = (em & 0x7000u) == 0x7000u || pos1 + (int)len > end || nmatch + 2u > ZA_MATCHQ_PER_SEG;
                    const bool take = want && !bad_data && !bad_index;
                    const int err = want && !take ? (bad_data && (em & 0x7000u) != 0x7000u ? 2 : 1) : 0;
                    // -- the literals go into the open block
                    if (nl) {
                        const uint32_t o = nlit & 15u, s8 = (o & 7u) * 8u;
                        const uint64_t t = grp << s8;
                        const uint64_t sp = s8 ? grp >> (64u - s8) : 0ull;           // bytes that cross into the next half
                        if (o < 8u) { blk_lo |= t; blk_hi |= sp; } else blk_hi |= t;
                        if (o + nl >= 16u) {
                            ZaU4u v; v.x = (uint32_t)blk_lo; v.y = (uint32_t)(blk_lo >> 32); v.z = (uint32_t)blk_hi; v.w = (uint32_t)(blk_hi >> 32);
                            *(ZaU4u *)(litp + (nlit & ~15u)) = v;
                            blk_lo = o >= 8u ? sp : 0ull;                             // what did not fit opens the next block (o + nl > 16 needs o >= 11)
                            blk_hi = 0;
                        }
                        nlit += nl;
                    }
                    // -- one queue entry per round at most: the match with the literals in front of it, or a long run's count
                    const uint32_t g2 = gap + nl;                     // <= 26 + 6
                    if (take) push((dist - 1u) | ((len - 2u) << 15) | (g2 << 24));
                    else if (g2 >= 27u) push(g2);
                    gap = (take || g2 >= 27u) ? 0u : g2;
                    pos = pos1 + (take ? (int)len : 0);
                    bp += u + (take ? off2 + dnx : 0u);
                    if (err) { lane_err = err; done = true; }
                    if (pos >= end) done = true;
                }
            }
        }
        if (act && !lane_err && (nlit & 15u)) {                 // the block that was open when the segment ended
            const uint32_t b = nlit & ~15u;
            // (a segment's symbol area is twice its bytes: the block always fits)
            ZaU4u v; v.x = (uint32_t)blk_lo; v.y = (uint32_t)(blk_lo >> 32); v.z = (uint32_t)blk_hi; v.w = (uint32_t)(blk_hi >> 32);
            *(ZaU4u *)(litp + b) = v;
        }
        {   // the last, partial group of queue entries
            const uint32_t k = nmatch & 3u, b4 = nmatch & ~3u;
            if (k > 0) myq[b4] = qb0;
            if (k > 1) myq[b4 + 1] = qb1;
            if (k > 2) myq[b4 + 2] = qb2;
        }
        if (act && !lane_err && bp != my_stop) lane_err = 1;
        if (act && !lane_err && lane == nseg - 1) {     // the last segment must be followed by end-of-block
            const uint32_t e = T.lut_l[(uint32_t)za_peek(src, bp) & ((1u << ZA_ML_BITS) - 1u)];         // bp == my_stop <= in_bits
            // behind the end-of-block code: the stream's last unit ends on its byte; any other with the sync marker (000, padding,
            // 00 00 FF FF)
            const uint32_t eob_end = bp + (e & 15u);
            if ((e & 0xF000u) != 0xF000u) lane_err = 1;
            else if (last ? ((eob_end + 7u) >> 3) != (uint32_t)m.in_len
                          : (((eob_end + 3u + 7u) >> 3) + 4u != (uint32_t)m.in_len || ((uint32_t)za_peek(src, eob_end) & 7u) != 0u ||
                             za_ld32(src + ((eob_end + 3u + 7u) >> 3)) != 0xFFFF0000u)) lane_err = 1;
        }
    }
    const unsigned long long e1 = __ballot(lane_err == 1), e2 = __ballot(lane_err == 2);
    if (e1 || e2) ZA_UM_FAIL(e2 ? ZA_I_DATA : ZA_I_INDEX);
    __threadfence_block();       // the literal bytes and the match queues are visible to the whole wave


    // ---- phase B: expand, segment by segment, as za_k_inflate_members does -- but every symbol is 16 bits wide (a byte, or
    // 256 + j = "byte j of the 32 KiB in front of this unit", which nobody knows yet), kept as TWO BYTE PLANES: the low bytes in
    // the row area, the high bytes where the decode tables stood.  A copy moves the same offsets of both planes, so the index
    // arithmetic, the dependency masks and the tail handling are the member decoder's; literals have a zero high byte (the high
    // plane of a segment starts as zeros and only matches write it).  Positions are AREA coordinates (unit position + 32 768): a
    // source in front of the unit is a far source like any other and reads marker symbols, which an initialisation kernel put in
    // front of every unit's symbols once.  No CRC here: the bytes are not known until the windows are.
    {
        uint8_t *img = (uint8_t *)rows;                            // [0, 272): symbols in front of the segment, [272, 272 + 2048): the segment -- low bytes
        uint8_t *imh = (uint8_t *)&T;                              // ... high bytes
        const uint32_t TAIL = 272u, AB = (uint32_t)ZA_WIN;
        static_assert(sizeof(uint32_t) * 64 * ZA_IROW >= 272 + ZA_SEG + 32, "low plane");
        static_assert(sizeof(ZaMemTabs) >= 272 + ZA_SEG + 32, "high plane");
        __builtin_amdgcn_wave_barrier();
        // the 272 symbols in front of the unit: markers
        for (uint32_t i = (uint32_t)lane; i < TAIL; i += 64) { const uint32_t sy = 256u + (AB - TAIL + i); img[i] = (uint8_t)sy; imh[i] = (uint8_t)(sy >> 8); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 16 symbols at dst16 + a -> their low and high bytes
        auto ld_syms = [&](uint32_t a, ZaU4u &lo, ZaU4u &hi) {
            const ZaU4u d0 = *(const ZaU4u *)(dst16 + a), d1 = *(const ZaU4u *)(dst16 + a + 8);
            lo.x = __builtin_amdgcn_perm(d0.y, d0.x, 0x06040200u); lo.y = __builtin_amdgcn_perm(d0.w, d0.z, 0x06040200u);
            lo.z = __builtin_amdgcn_perm(d1.y, d1.x, 0x06040200u); lo.w = __builtin_amdgcn_perm(d1.w, d1.z, 0x06040200u);
            hi.x = __builtin_amdgcn_perm(d0.y, d0.x, 0x07050301u); hi.y = __builtin_amdgcn_perm(d0.w, d0.z, 0x07050301u);
            hi.z = __builtin_amdgcn_perm(d1.y, d1.x, 0x07050301u); hi.w = __builtin_amdgcn_perm(d1.w, d1.z, 0x07050301u);
        };
        // up to 32 bytes (v, then v2) to a plane at o
        auto put = [&](uint8_t *o, uint32_t len, ZaU4u v, ZaU4u v2) {
            uint32_t rem = len;
            if (len > 16u) {
                *(za_u32u *)o = v.x; *(za_u32u *)(o + 4) = v.y; *(za_u32u *)(o + 8) = v.z; *(za_u32u *)(o + 12) = v.w;
                o += 16; v = v2; rem = len - 16u;
            }
            if (rem & 16u) { *(za_u32u *)o = v.x; *(za_u32u *)(o + 4) = v.y; *(za_u32u *)(o + 8) = v.z; *(za_u32u *)(o + 12) = v.w; }
            else {
                if (rem & 8u) { *(za_u32u *)o = v.x; *(za_u32u *)(o + 4) = v.y; o += 8; v.x = v.z; v.y = v.w; }
                if (rem & 4u) { *(za_u32u *)o = v.x; o += 4; v.x = v.y; }
                if (rem & 2u) { *(za_u16u *)o = (uint16_t)v.x; o += 2; v.x >>= 16; }
                if (rem & 1u) *o = (uint8_t)v.x;
            }
        };
        for (int s = 0; s < nseg; s++) {
            const uint32_t cnt = __shfl(nmatch, s, 64), lits = __shfl(nlit, s, 64);
            const uint32_t *q = matchq + ((size_t)blockIdx.x * 64 + (size_t)s) * ZA_MATCHQ_PER_SEG;
            const uint32_t seg_start = AB + ((uint32_t)s << ZA_SEG_SHIFT);
            const uint32_t seg_len = (uint32_t)n - ((uint32_t)s << ZA_SEG_SHIFT) < (uint32_t)ZA_SEG ? (uint32_t)n - ((uint32_t)s << ZA_SEG_SHIFT) : (uint32_t)ZA_SEG;
            uint32_t ent_next = (uint32_t)lane < cnt ? q[lane] : 0u;
            // images: the tail of the previous segment moves to the front (final symbols); the literal BYTES come from the front of
            // the segment's own symbol area, piece by piece of 16 where a piece holds any (image byte x is literal byte x - shift)
            __builtin_amdgcn_wave_barrier();
            uint32_t t0 = 0, t1 = 0, u0 = 0, u1 = 0;
            if (s > 0) {
                t0 = ((const uint32_t *)(img + ZA_SEG))[lane]; u0 = ((const uint32_t *)(imh + ZA_SEG))[lane];
                if (lane < 4) { t1 = ((const uint32_t *)(img + ZA_SEG))[64 + lane]; u1 = ((const uint32_t *)(imh + ZA_SEG))[64 + lane]; }
            }
            {
                const int shift = (int)(seg_len - lits);             // match symbols of the segment
                const uint8_t *litb = (const uint8_t *)(dst16 + seg_start);
                ZaU4u pc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int x = lane * 32 + 16 * j, o = x - shift;
                    // (bytes in front of the first literal or behind the last are of no meaning; they are readable: the area in front is
                    // the segment before or the marker symbols, the area behind is this segment's own 4 KiB)
                    if (x < (int)seg_len && o + 16 > 0) pc[j] = *(const ZaU4u *)(litb + o);
                }
                __builtin_amdgcn_wave_barrier();
                if (s > 0) {
                    ((uint32_t *)img)[lane] = t0; ((uint32_t *)imh)[lane] = u0;
                    if (lane < 4) { ((uint32_t *)img)[64 + lane] = t1; ((uint32_t *)imh)[64 + lane] = u1; }
                }
                *(uint4 *)(img + TAIL + lane * 32) = make_uint4(pc[0].x, pc[0].y, pc[0].z, pc[0].w);
                *(uint4 *)(img + TAIL + lane * 32 + 16) = make_uint4(pc[1].x, pc[1].y, pc[1].z, pc[1].w);
                *(uint4 *)(imh + TAIL + lane * 32) = make_uint4(0u, 0u, 0u, 0u);
                *(uint4 *)(imh + TAIL + lane * 32 + 16) = make_uint4(0u, 0u, 0u, 0u);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t segpos = seg_start;
            uint32_t mrem = seg_len - lits;
            for (uint32_t g = 0; g < cnt; g += 64) {
                const bool hasq = g + (uint32_t)lane < cnt;
                const uint32_t ent = ent_next;
                ent_next = g + 64u + (uint32_t)lane < cnt ? q[g + 64u + lane] : 0u;
                const uint32_t l2 = (ent >> 15) & 0x1FFu;
                const bool has = hasq && l2 != 0u;
                const uint32_t mlen = has ? l2 + 2u : 0u, mdist = (ent & 0x7FFFu) + 1u;
                const uint32_t glit = !hasq ? 0u : has ? (ent >> 24) : ent;
                const uint32_t incl = za_wave_incl_scan(glit + mlen), incm = za_wave_incl_scan(mlen);
                const uint32_t mdst = segpos + incl - mlen;
                const uint32_t up = mrem - (incm - mlen);
                segpos += (uint32_t)__shfl((int)incl, 63, 64);
                mrem -= (uint32_t)__shfl((int)incm, 63, 64);
                const uint32_t ioff = TAIL + (mdst - seg_start);       // my match's destination inside the images; my literals end there
                uint8_t *od = img + ioff, *oh = imh + ioff;
                // the runs of literals move down (low plane only: their high bytes are the zeros the plane started with)
                if (__ballot(glit != 0u && up != 0u) != 0ull) {
                    const uint8_t *sp = od - glit + up;
                    ZaU4u a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
                    const bool mv = glit != 0u && up != 0u;
                    if (mv) {
                        a.x = *(const za_u32u *)sp; a.y = *(const za_u32u *)(sp + 4); a.z = *(const za_u32u *)(sp + 8); a.w = *(const za_u32u *)(sp + 12);
                        if (glit > 16u) { b.x = *(const za_u32u *)(sp + 16); b.y = *(const za_u32u *)(sp + 20); b.z = *(const za_u32u *)(sp + 24); b.w = *(const za_u32u *)(sp + 28); }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (mv) put(od - glit, glit, a, b);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                bool done = !has;
                unsigned long long pending = __ballot(!done);
                const uint32_t sdst = hasq ? mdst : 0xFFFFFFFFu, send = hasq ? mdst + mlen : 0xFFFFFFFFu;
                const uint32_t src_a = mdst - mdist, src_b = src_a + (mlen < mdist ? mlen : mdist);      // (area coordinates: never negative)
                uint32_t jhi = 0, jlo = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                    const uint32_t vd = (uint32_t)__shfl((int)sdst, (int)(jhi + step - 1u), 64);
                    const uint32_t ve = (uint32_t)__shfl((int)send, (int)(jlo + step - 1u), 64);
                    if (vd < src_b) jhi += step;
                    if (ve <= src_a) jlo += step;
                }
                const unsigned long long deps = ((1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull);
                const bool far = src_a + TAIL < seg_start;             // final symbols in memory (or the markers in front of the unit)
                const bool simple = mdist >= mlen && mlen <= 32u;
                ZaU4u fl = {0, 0, 0, 0}, fh = {0, 0, 0, 0}, fl2 = {0, 0, 0, 0}, fh2 = {0, 0, 0, 0};
                if (has && far && simple) {
                    ld_syms(src_a, fl, fh);
                    if (mlen > 16u) ld_syms(src_a + 16u, fl2, fh2);
                }
                while (pending) {
                    const bool ready = !done && (pending & deps) == 0ull;
                    if (ready && simple) {
                        ZaU4u v = fl, v2 = fl2, w = fh, w2 = fh2;
                        if (!far) {
                            const uint8_t *sp = od - mdist, *sh = oh - mdist;       // inside the images: src_a >= seg_start - 272
                            v.x = *(const za_u32u *)sp; v.y = *(const za_u32u *)(sp + 4); v.z = *(const za_u32u *)(sp + 8); v.w = *(const za_u32u *)(sp + 12);
                            w.x = *(const za_u32u *)sh; w.y = *(const za_u32u *)(sh + 4); w.z = *(const za_u32u *)(sh + 8); w.w = *(const za_u32u *)(sh + 12);
                            if (mlen > 16u) {
                                v2.x = *(const za_u32u *)(sp + 16); v2.y = *(const za_u32u *)(sp + 20); v2.z = *(const za_u32u *)(sp + 24); v2.w = *(const za_u32u *)(sp + 28);
                                w2.x = *(const za_u32u *)(sh + 16); w2.y = *(const za_u32u *)(sh + 20); w2.z = *(const za_u32u *)(sh + 24); w2.w = *(const za_u32u *)(sh + 28);
                            }
                        }
                        put(od, mlen, v, v2);
                        put(oh, mlen, w, w2);
                    }
                    // long or self-overlapping matches: the whole wave copies them, one at a time
                    unsigned long long coop = __ballot(ready && !simple);
                    while (coop) {
                        const int j = __builtin_ctzll(coop);
                        coop &= coop - 1ull;
                        const uint32_t cd = (uint32_t)__builtin_amdgcn_readlane((int)mdst, j);
                        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)mlen, j);
                        const uint32_t cdist = (uint32_t)__builtin_amdgcn_readlane((int)mdist, j);
                        const bool cfar = cd - cdist + TAIL < seg_start;      // (then cdist > cl: no overlap)
                        uint8_t *o = img + TAIL + (cd - seg_start), *ohh = imh + TAIL + (cd - seg_start);
                        const float rd = 1.0f / (float)cdist;
                        for (uint32_t base = 0; base < cl; base += 64) {
                            const uint32_t i = base + (uint32_t)lane;
                            if (i < cl) {
                                int k = (int)i;
                                if (cdist < cl) {
                                    k = (int)i - (int)cdist * (int)((float)i * rd);
                                    if (k < 0) k += (int)cdist;
                                    if (k >= (int)cdist) k -= (int)cdist;
                                }
                                if (cfar) { const uint32_t sy = dst16[cd - cdist + (uint32_t)k]; o[i] = (uint8_t)sy; ohh[i] = (uint8_t)(sy >> 8); }
                                else { o[i] = (o - cdist)[k]; ohh[i] = (ohh - cdist)[k]; }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    done = done || ready;
                    pending = __ballot(!done);
                }
            }
            // the finished segment: 32 symbols per lane, the planes interleaved on the way out (its last, partial piece symbol by symbol)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t o = (uint32_t)lane * 32u;
                if (o + 32u <= seg_len) {
                    const uint4 a = *(const uint4 *)(img + TAIL + o), b2 = *(const uint4 *)(img + TAIL + o + 16);
                    const uint4 c4 = *(const uint4 *)(imh + TAIL + o), d4 = *(const uint4 *)(imh + TAIL + o + 16);
                    const uint32_t lw[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w}, hw[8] = {c4.x, c4.y, c4.z, c4.w, d4.x, d4.y, d4.z, d4.w};
                    uint16_t *od16 = dst16 + seg_start + o;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        ZaU4u v;
                        v.x = __builtin_amdgcn_perm(hw[2 * k], lw[2 * k], 0x05010400u); v.y = __builtin_amdgcn_perm(hw[2 * k], lw[2 * k], 0x07030602u);
                        v.z = __builtin_amdgcn_perm(hw[2 * k + 1], lw[2 * k + 1], 0x05010400u); v.w = __builtin_amdgcn_perm(hw[2 * k + 1], lw[2 * k + 1], 0x07030602u);
                        *(ZaU4u *)(od16 + 8 * k) = v;
                    }
                } else for (uint32_t k = 0; k < 32u && o + k < seg_len; k++)
                    dst16[seg_start + o + k] = (uint16_t)((uint32_t)img[TAIL + o + k] | ((uint32_t)imh[TAIL + o + k] << 8));
            }
            __threadfence_block();       // later segments read these symbols from memory
        }
    }
    if (lane == 0) { ZaChunkRes r; r.status = last ? ZA_I_END : ZA_I_SYNC; r.max_back = 0; r.bits = (m.in_off + m.in_len) * 8ull; r.out_len = (uint64_t)n; res_out[blockIdx.x] = r; }
#undef ZA_UM_FAIL
}

// the 32 768 marker symbols in front of every unit's symbols (256 + j = "byte j of the window in front of this unit")
__global__ __launch_bounds__(256) void za_k_fill_marker_prefix(uint16_t *__restrict__ out16, uint64_t area_stride)
{
    uint16_t *p = out16 + (uint64_t)blockIdx.x * area_stride;
    for (uint32_t j = 8u * threadIdx.x; j < (uint32_t)ZA_WIN; j += 8u * 256u) {
        uint4 v;
        v.x = (256u + j) | ((257u + j) << 16); v.y = (258u + j) | ((259u + j) << 16); v.z = (260u + j) | ((261u + j) << 16); v.w = (262u + j) | ((263u + j) << 16);
        *(uint4 *)(p + j) = v;
    }
}
