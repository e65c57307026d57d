// Inflate kernels for MI355X (gfx950).  Product code; hand-written HIP, wave64.
//
// Replaces zng_inflate behind zlib_ng.decompress and _GzipReader (reference
// src/zlib_ng/zlib_ngmodule.c:328 and :2539) and the member loop of GzipReader_read_into_buffer
// (:2426-2637) for streams of many independent members.
//
//   za_k_inflate_serial   any RFC 1951 stream, one wavefront per stream: inside a Huffman block the 64 lanes
//                         decode 64 self-synchronising sub-sequences at once (za_par_sweep); block headers,
//                         stored blocks, the ends of the input / output and errors are handled by wave-uniform
//                         sequential rounds (32 KiB history ring in LDS, copies done by all 64 lanes); r06: a second
//                         wavefront counts the NEXT sweep while the first stores and resolves the current one (ZaSweepHelp)
//   za_k_inflate_serial_small   the same for the small one-shot calls: the output assembled in an LDS image
//   za_k_scan_members     pass 1 of the two-pass scheme: coalesced sweep of the compressed stream
//                         for this engine's indexed gzip members ('Z','A' FEXTRA subfield)
//   za_k_inflate_members  pass 2: one wavefront per member, one lane per 2 KiB segment decodes its
//                         tokens from the indexed bit offset (literals stored directly, matches
//                         queued), then the wave resolves the queued matches in output order,
//                         64 at a time, byte-parallel; CRC-32 / ISIZE verified in the same kernel
//   za_k_assemble_members writes header + index + deflate bytes + trailer of each member
#include "za_common.h"
#include "za_crc.h"

#ifndef ZA_LUT_L_BITS
#define ZA_LUT_L_BITS 10
#endif
#define ZA_IBUF_DW 128             // dwords of bitstream staged in LDS by the sequential decoder
#ifndef ZA_LUT_D_BITS
#define ZA_LUT_D_BITS 9
#endif

// internal status values (host maps them to ZNGAMD_* codes)
#define ZA_I_OK          0
#define ZA_I_END         1
#define ZA_I_DATA      (-3)
#define ZA_I_INPUT     (-5)     // ran out of input
#define ZA_I_OUTFULL   (-6)     // ran out of output space
#define ZA_I_INDEX     (-7)     // segment index inconsistent with the stream: decode sequentially
#define ZA_I_CRC       (-104)
#define ZA_I_LENGTH    (-105)

// LB / DB = index bits of the first-level tables (codes longer than that: za_long_decode).  The chunk kernels take 10 / 9 -- the
// codes this engine writes are at most 10 / 9 bits long --, the members kernel 9 / 8: 1.5 KiB less LDS per wavefront are worth more
// there (15 per CU instead of 13: 9.42 -> 7.89 ms per GiB of zlib-written members) than the long codes that miss the table cost.
template <int LB, int DB>
struct ZaInfTabsT {
    static constexpr int kLBits = LB, kDBits = DB;
    static_assert(LB >= 7 && DB >= 7 && LB <= 15 && DB <= 15, "za_long_decode answers for the lengths 8..15 only: a first-level table must hold every code of up to 7 bits");
    uint16_t lut_l[1 << LB];   // (sym<<4)|len, 0 = longer than the LUT or unassigned
    uint16_t lut_d[1 << DB];
    uint16_t cnt_l[16], cnt_d[16];
    uint16_t fst_l[16], fst_d[16];        // first code of each length (as a number, most significant bit first) and the place of
    uint16_t idx_l[16], idx_d[16];        // its symbol in sym_*: the codes of one length are consecutive (za_long_decode)
    uint16_t sym_l[288], sym_d[32];
    uint8_t lens[320];
    int status;
};
using ZaInfTabs = ZaInfTabsT<ZA_LUT_L_BITS, ZA_LUT_D_BITS>;

// Every kernel that builds decode tables or runs the sequential decoder is ONE wavefront per workgroup (or, r06, lets one of its
// wavefronts do so while another works beside it: za_k_inflate_serial_small): what orders the wave's LDS traffic is a fence and a
// wave barrier -- no s_barrier, which would wait for wavefronts that never come.
__device__ __forceinline__ void za_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct ZaInfResult {
    int32_t status; uint32_t pad;
    uint64_t out_len;
    uint64_t in_bits;      // bits consumed (counted from bit 0 of the buffer)
    uint64_t block_bits;   // bit offset of the header of the last block that was entered (resume point)
    uint64_t block_out;    // bytes produced before that block
};

// bit-serial canonical decode of the low bits of v; returns (sym<<4)|len or 0
__device__ __forceinline__ uint32_t za_slow_decode(uint32_t v, const uint16_t *cnt, const uint16_t *sym, int maxbits)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= maxbits; l++) {
        code |= (int)(v & 1u); v >>= 1;
        const int c = cnt[l];
        if (code - c < first) return ((uint32_t)sym[index + (code - first)] << 4) | (uint32_t)l;
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return 0;
}

// A code longer than the first-level table in one step (the bit-serial walk above costs fifteen dependent LDS reads, and a wave
// takes it whenever ONE of its lanes meets such a code -- rare per symbol, but with 64 lanes nearly every round): the stream's
// bits reversed are the code with its first bit on top, the codes of length l are the integers fst[l] .. fst[l] + cnt[l] - 1,
// and no extension of a shorter code reaches fst[l] (canonical order), so at most one length answers.  `from` = first length
// the table does not hold.  Returns (sym << 4) | len, 0 = no such code.
__device__ __forceinline__ uint32_t za_long_decode(uint32_t v, const uint16_t *cnt, const uint16_t *fst, const uint16_t *idx, const uint16_t *sym, int from)
{
    const uint32_t r = __builtin_bitreverse32(v);
    uint32_t hit = 0;
#pragma unroll
    for (int l = 15; l >= 8; l--) {
        if (l >= from) {
            const uint32_t rel = (r >> (32 - l)) - (uint32_t)fst[l];
            if (rel < (uint32_t)cnt[l]) hit = (((uint32_t)idx[l] + rel) << 4) | (uint32_t)l;
        }
    }
    if (hit) hit = ((uint32_t)sym[hit >> 4] << 4) | (hit & 15u);
    return hit;
}

// Build count/symbol arrays and the LUT from lens[0..n), n <= 320, by the whole wave.  Returns <0 for an
// over-subscribed set, >0 for an incomplete one, 0 otherwise; *shared_maxlen = longest code.
// Every lane holds the lengths of symbols lane, lane + 64, ...; the number of codes of each length is a sum of
// ballot population counts (wave-uniform, so it lives in scalar registers), and a symbol's slot in the canonical
// order is the offset of its length plus the number of lower symbols of that length, again from the ballots.
__device__ int za_build_table(const uint8_t *lens, int n, uint16_t *cnt, uint16_t *sym, uint16_t *lut, int lut_bits,
                              int *shared_status, int *shared_maxlen, uint16_t *fst = nullptr, uint16_t *idx = nullptr)
{
    const int lane = za_lane();
    za_wave_sync();
    int L[5];
#pragma unroll
    for (int b = 0; b < 5; b++) { const int i = b * 64 + lane; L[b] = i < n ? (int)lens[i] : 0; }
    int c[16];
    c[0] = 0;
    int maxlen = 0, coded = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
        int t = 0;
#pragma unroll
        for (int b = 0; b < 5; b++) t += __popcll(__ballot(L[b] == l));
        c[l] = t; coded += t;
        if (t) maxlen = l;
    }
    int st = 0;
    if (coded) {
        int left = 1;
#pragma unroll
        for (int l = 1; l <= 15; l++) { if (left >= 0) { left <<= 1; left -= c[l]; } }
        st = left;
    }
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l <= 15; l++) cnt[l] = (uint16_t)c[l];
        *shared_status = st; *shared_maxlen = maxlen;
        if (fst) {
            int first = 0, index = 0;
            fst[0] = 0; idx[0] = 0;
#pragma unroll
            for (int l = 1; l <= 15; l++) { fst[l] = (uint16_t)first; idx[l] = (uint16_t)index; index += c[l]; first = (first + c[l]) << 1; }
        }
    }
    // (r06) The first-level table is filled BY SYMBOL: a symbol of length l <= lut_bits owns the 2^(lut_bits - l) entries whose low l
    // bits are its code in stream order (the code's bits reversed), and a lane learns its symbols' codes where it places them --
    // first code of the length + the symbol's rank among its length's symbols.  Until r06 every lane decoded its share of the
    // ENTRIES bit by bit (ten dependent steps and an LDS read per entry, sixteen entries a lane for the 10-bit table): 8 of the
    // 16 us the literal/length table of a dynamic header took.
    uint32_t ecode[5] = {0u, 0u, 0u, 0u, 0u};
    if (st >= 0) {
        const unsigned long long below = (1ull << lane) - 1ull;
        int base = 0, first = 0;
#pragma unroll
        for (int l = 1; l <= 15; l++) {
            first = (first + c[l - 1]) << 1;                     // first code of length l (c[0] = 0)
            if (c[l]) {
                const int idx_l = base;
#pragma unroll
                for (int b = 0; b < 5; b++) {
                    const unsigned long long m = __ballot(L[b] == l);
                    if (L[b] == l) {
                        const int pos = base + __popcll(m & below);
                        sym[pos] = (uint16_t)(b * 64 + lane);
                        ecode[b] = __brev((uint32_t)(first + (pos - idx_l))) >> (32 - l);
                    }
                    base += __popcll(m);
                }
            }
        }
        for (int e = lane; e < (1 << lut_bits); e += 64) lut[e] = 0;      // (entries of longer codes, and of an incomplete code's gaps, stay 0)
    }
    za_wave_sync();
    if (st >= 0) {
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const int l = L[b];
            if (l != 0 && l <= lut_bits) {
                const uint16_t entry = (uint16_t)(((uint32_t)(b * 64 + lane) << 4) | (uint32_t)l);
                for (uint32_t e = ecode[b]; e < (1u << lut_bits); e += 1u << l) lut[e] = entry;
            }
        }
    }
    za_wave_sync();
    return st;
}

__device__ __forceinline__ uint64_t za_peek(const uint8_t *in, uint64_t bitpos)
{
    return za_ld64(in + (bitpos >> 3)) >> (bitpos & 7u);     // >= 57 valid bits; buffer is padded by 8 bytes
}

__device__ __forceinline__ uint32_t za_decode_sym(uint64_t bits, const uint16_t *lut, int lut_bits,
                                                  const uint16_t *cnt, const uint16_t *sym)
{
    uint32_t e = lut[bits & ((1u << lut_bits) - 1u)];
    if (!e) e = za_slow_decode((uint32_t)bits, cnt, sym, 15);
    return e;     // (sym<<4)|len, 0 = invalid code
}

__device__ __forceinline__ int za_len_base(int code, int &nextra)    // code 0..28
{
    if (code < 8) { nextra = 0; return 3 + code; }
    if (code == 28) { nextra = 0; return 258; }
    const int nb = (code >> 2) + 1;
    nextra = nb - 2;
    return ((4 + (code & 3)) << (nb - 2)) + 3;
}
__device__ __forceinline__ int za_dist_base(int code, int &nextra)   // code 0..29
{
    if (code < 4) { nextra = 0; return code + 1; }
    const int nb = code >> 1;
    nextra = nb - 1;
    return ((2 + (code & 1)) << (nb - 1)) + 1;
}

__constant__ uint8_t za_i_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Parse a fixed or dynamic block header at `bitpos` (uniform across the wave) and build the decode
// tables.  Returns ZA_I_OK / ZA_I_DATA / ZA_I_INPUT; advances bitpos past the header.
#ifdef ZA_PS_STATS
__device__ unsigned long long za_ps_stat[32];     // profiling build (profiles/ps_stats.sh): 0..15 the sweeps' counters, 16: rounds of the counting passes (as the wave runs them), 17: their core clock cycles, 18: tokens counted, 20..24: phases of a dynamic header
#define ZA_STAT_ADD(i, v) do { if (za_lane() == 0) atomicAdd(&za_ps_stat[i], (unsigned long long)(v)); } while (0)
#define ZA_STAT_T() wall_clock64()
#else
#define ZA_STAT_ADD(i, v) do { } while (0)
#define ZA_STAT_T() 0ull
#endif
#define ZA_HDR_DW 148     // dwords of LDS that hold a whole dynamic header behind its three counts: 57 + 316 * 14 bits, + alignment and look-ahead
template <typename TT>
__device__ int za_read_tables(const uint8_t *in, uint64_t in_bits, uint64_t &bitpos, int type, TT &T, int *scratch /*2 ints LDS*/,
                              uint32_t *hb = nullptr /* ZA_HDR_DW dwords of LDS, or none: the header is then read from memory bit by bit */)
{
    const int lane = za_lane();
    if (type == 1) {
        za_wave_sync();
        for (int i = lane; i < 320; i += 64)
            T.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : i < 318 ? 5 : 0);
        za_build_table(T.lens, 288, T.cnt_l, T.sym_l, T.lut_l, TT::kLBits, &scratch[0], &scratch[1], T.fst_l, T.idx_l);
        za_build_table(T.lens + 288, 30, T.cnt_d, T.sym_d, T.lut_d, TT::kDBits, &scratch[0], &scratch[1], T.fst_d, T.idx_d);
        return ZA_I_OK;
    }
    if (bitpos + 14 > in_bits) return ZA_I_INPUT;
    unsigned long long tph = ZA_STAT_T();
#define ZA_HDR_T(i) do { const unsigned long long t_ = ZA_STAT_T(); ZA_STAT_ADD(i, t_ - tph); tph = t_; } while (0)
    uint64_t bits = za_peek(in, bitpos);
    const int nlen = (int)(bits & 31u) + 257, ndist = (int)((bits >> 5) & 31u) + 1, ncode = (int)((bits >> 10) & 15u) + 4;
    bitpos += 14;
    if (nlen > 286 || ndist > 30) return ZA_I_DATA;
    if (bitpos + 3ull * (unsigned)ncode > in_bits) return ZA_I_INPUT;
    za_wave_sync();
    if (lane < 19) T.lens[lane] = 0;
    // The code lengths are decoded by one lane, one dependent read per symbol: from LDS that is a fraction of a microsecond
    // for the whole header, from memory about a microsecond per symbol.
    const uint64_t hbyte = (bitpos >> 3) & ~3ull;
    if (hb) {
        const uint64_t in_len8 = (in_bits >> 3) + 8ull;                  // the buffer is padded by >= 8 bytes
        for (int i = lane; i < ZA_HDR_DW; i += 64) {
            const uint64_t o = hbyte + 4ull * (unsigned)i;
            uint32_t v = 0;
            if (o + 4 <= in_len8) v = za_ld32(in + o);
            else for (int k = 0; k < 4; k++) if (o + (unsigned)k < in_len8) v |= (uint32_t)in[o + (unsigned)k] << (8 * k);
            hb[i] = v;
        }
    }
    za_wave_sync();
    auto peek = [&](uint64_t bp) -> uint64_t {
        if (!hb) return za_peek(in, bp);
        const uint32_t rel = (uint32_t)(bp - hbyte * 8ull), w = rel >> 5, sh = rel & 31u;
        const uint64_t lo = ((uint64_t)hb[w + 1] << 32) | hb[w];
        return sh ? ((lo >> sh) | ((uint64_t)hb[w + 2] << (64 - sh))) : lo;
    };
    if (lane < ncode) T.lens[za_i_cl_order[lane]] = (uint8_t)(peek(bitpos + 3ull * (unsigned)lane) & 7u);      // (a lane per 3-bit field)
    bitpos += 3ull * (unsigned)ncode;
    ZA_HDR_T(20);
    // code-length code: tables go to the distance slots for now (7-bit LUT)
    int st = za_build_table(T.lens, 19, T.cnt_d, T.sym_d, T.lut_d, 7, &scratch[0], &scratch[1]);
    ZA_HDR_T(21);
    if (st != 0) return ZA_I_DATA;                       // must be complete
    // decode nlen + ndist code lengths, result in T.lens[0..] then moved
    za_wave_sync();
    int err = ZA_I_OK, adv = 0;
    if (hb) {
        // A chain of dependent decodes -- but nothing in it needs LDS (r06): the staged header lies in three registers across the
        // lanes and the 7-bit table of the code-length code in two, every position is wave-uniform, so a window of the stream is
        // two v_readlane and a scalar shift, a look-up one v_readlane, and the loop runs on the scalar unit; the lengths are written
        // as they come (a run by as many lanes as it is long).  Lane 0 alone, three LDS round trips per symbol: 35 of the 60 us a
        // dynamic header took -- the whole latency of a small stream's first block.
        uint32_t hv[3], lutv[2];
#pragma unroll
        for (int k = 0; k < 3; k++) { const int i = lane + 64 * k; hv[k] = i < ZA_HDR_DW ? hb[i] : 0u; }
#pragma unroll
        for (int k = 0; k < 2; k++) lutv[k] = T.lut_d[lane + 64 * k];
        auto rdh = [&](uint32_t w) -> uint32_t {
            const uint32_t r = w >> 6;
            const uint32_t v = r == 0u ? hv[0] : r == 1u ? hv[1] : hv[2];
            return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(w & 63u));
        };
        uint64_t bp = bitpos;
        int idx = 0;
        uint32_t prev = 0;
        uint8_t *L = T.lens;           // from index 0: the 19 code-length lengths are no longer needed (their table is in registers now)
        const int ntot = nlen + ndist;
        while (idx < ntot) {
            if (bp > in_bits) { err = ZA_I_INPUT; break; }
            const uint32_t rel = (uint32_t)(bp - hbyte * 8ull), w = rel >> 5, sh = rel & 31u;
            const uint64_t lo = ((uint64_t)rdh(w + 1u) << 32) | rdh(w);
            const uint32_t b = (uint32_t)(lo >> sh);                 // 32 bits of the stream: a code and its extra bits take 14 at most
            const uint32_t li = b & 127u;
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)(li < 64u ? lutv[0] : lutv[1]), (int)(li & 63u));
            if (!e) { err = ZA_I_DATA; break; }                      // (the code-length code has no code longer than its table)
            const int sy = (int)(e >> 4), l = (int)(e & 15u);
            bp += (unsigned)l;
            if (bp > in_bits) { err = ZA_I_INPUT; break; }
            if (sy < 16) { if (lane == 0) L[idx] = (uint8_t)sy; idx++; prev = (uint32_t)sy; }
            else {
                int rep; uint32_t val = 0;
                const uint32_t x = b >> l;
                if (sy == 16) { if (idx == 0) { err = ZA_I_DATA; break; } val = prev; rep = 3 + (int)(x & 3u); bp += 2; }
                else if (sy == 17) { rep = 3 + (int)(x & 7u); bp += 3; }
                else { rep = 11 + (int)(x & 127u); bp += 7; }
                if (bp > in_bits) { err = ZA_I_INPUT; break; }
                if (idx + rep > ntot) { err = ZA_I_DATA; break; }
                for (int j = lane; j < rep; j += 64) L[idx + j] = (uint8_t)val;
                idx += rep;
                prev = val;
            }
        }
        adv = (int)(bp - bitpos);
        za_wave_sync();
        if (err == ZA_I_OK && L[256] == 0) err = ZA_I_DATA;          // missing end-of-block
        za_wave_sync();
    } else {
    if (lane == 0) {
        uint64_t bp = bitpos;
        int idx = 0, err = ZA_I_OK;
        uint8_t *L = T.lens + 0;      // reuse from index 0: the 19 code-length lengths are no longer needed
        uint8_t tmp_prev = 0;
        while (idx < nlen + ndist) {
            if (bp > in_bits) { err = ZA_I_INPUT; break; }
            const uint64_t b = peek(bp);
            const uint32_t e = za_decode_sym(b, T.lut_d, 7, T.cnt_d, T.sym_d);
            if (!e) { err = ZA_I_DATA; break; }
            const int s = (int)(e >> 4), l = (int)(e & 15u);
            bp += (unsigned)l;
            if (bp > in_bits) { err = ZA_I_INPUT; break; }
            if (s < 16) { L[idx++] = (uint8_t)s; tmp_prev = (uint8_t)s; }
            else {
                int rep, val = 0;
                const uint64_t x = b >> l;
                if (s == 16) { if (idx == 0) { err = ZA_I_DATA; break; } val = tmp_prev; rep = 3 + (int)(x & 3u); bp += 2; }
                else if (s == 17) { rep = 3 + (int)(x & 7u); bp += 3; }
                else { rep = 11 + (int)(x & 127u); bp += 7; }
                if (bp > in_bits) { err = ZA_I_INPUT; break; }
                if (idx + rep > nlen + ndist) { err = ZA_I_DATA; break; }
                while (rep--) L[idx++] = (uint8_t)val;
                tmp_prev = (uint8_t)val;
            }
        }
        if (err == ZA_I_OK && L[256] == 0) err = ZA_I_DATA;      // missing end-of-block
        scratch[0] = err;
        scratch[1] = (int)(bp - bitpos);
    }
    za_wave_sync();
    err = scratch[0];
    adv = scratch[1];
    za_wave_sync();
    }
    if (err != ZA_I_OK) return err;
    bitpos += (unsigned)adv;
    ZA_HDR_T(22);
    // distance lengths first (they sit after the literal/length ones), into lens[288..]
    if (lane < 32) {
        const uint8_t v = lane < ndist ? T.lens[nlen + lane] : (uint8_t)0;
        T.lens[288 + lane] = v;
    }
    za_wave_sync();
    for (int i = nlen + lane; i < 288; i += 64) T.lens[i] = 0;
    za_wave_sync();
    int maxl;
    st = za_build_table(T.lens, nlen, T.cnt_l, T.sym_l, T.lut_l, TT::kLBits, &scratch[0], &scratch[1], T.fst_l, T.idx_l);
    maxl = scratch[1];
    ZA_HDR_T(23);
    if (st < 0 || (st > 0 && maxl != 1)) return ZA_I_DATA;
    st = za_build_table(T.lens + 288, ndist, T.cnt_d, T.sym_d, T.lut_d, TT::kDBits, &scratch[0], &scratch[1], T.fst_d, T.idx_d);
    maxl = scratch[1];
    ZA_HDR_T(24);
    if (st < 0 || (st > 0 && maxl != 1)) return ZA_I_DATA;
    return ZA_I_OK;
#undef ZA_HDR_T
}

// ------------------------------------------------------------------------------------------------
// Parallel decode inside ONE Huffman block of a foreign stream (no index): self-synchronising sub-sequences.
// The next 64 x S bits of the block are cut into 64 sub-sequences, one per lane.  Only lane 0 knows a real token
// boundary; the others start at a guess.  Every lane decodes from its start up to the first token boundary at or
// behind the next lane's nominal start and hands that boundary on as the next lane's start; lanes whose start
// changed decode again.  Deflate's prefix codes re-synchronise after a few tokens, so after about three passes
// nothing changes any more and all starts are exact (lane 0 is exact, and a lane whose start and predecessors did
// not change is exact by induction).  After ZA_PS_MAXIT passes the lanes below the first one that still changed are
// exact and the sweep covers only those.  The counting passes give each lane's output length and number of
// matches; a last pass stores literals at their final place and queues the matches in LDS, and the queue is
// resolved in output order, 64 matches at a time, exactly like phase B of za_k_inflate_members.
// Anything unusual (end of input near, output nearly full, invalid code on the true chain, a distance before
// the history, more output / matches than the LDS queue addresses) shortens the sweep or leaves the position
// untouched; the sequential rounds of the caller then deal with it and produce the status.
#define ZA_PS_MINBITS 64
#ifndef ZA_PS_MAXIT
#define ZA_PS_MAXIT  6
#endif
#define ZA_PS_NONE   0xFFFFFFFFu
#ifndef ZA_PS_LITS
#define ZA_PS_LITS 3                 // a counting round: up to this many literals ...
#endif
#ifndef ZA_PS_LIT_GROUPS
#define ZA_PS_LIT_GROUPS 1           // ... this many times, then the token behind them
#endif
// LDS of the sweeps, sized per kernel: BITS = longest sub-sequence (bits per lane), Q = matches of one sweep.  The single-
// stream kernel (one wavefront on the whole GPU) takes long sub-sequences, 1024 bits, which re-synchronise in fewer passes;
// the kernels that run thousands of wavefronts (members, chunks) take 256 bits and a queue of 1024, which leaves room for
// 9-11 workgroups per CU instead of 4 (chunk decode of 128 MiB: 5-6.6 -> 3.5 ms, 256 MiB of BGZF members: 8.3 -> 5.1 ms).
#ifndef ZA_CHUNK_MAXIT
#define ZA_CHUNK_MAXIT 3            // pass limit of the chunk kernels (sub-sequences of 512 / 1 024 bits settle sooner, and a sweep cut short is followed by another at once: 256 MiB of this engine's stream 4.76 -> 4.24 ms, a zlib stream of 128 MiB 4.09 -> 3.66 ms against the limit of six)
#endif
template <int BITS, int Q, int MAXIT = ZA_PS_MAXIT>
struct ZaParBufT {
    static constexpr int kBits = BITS, kQ = Q, kMaxIt = MAXIT;
    uint32_t stage[64 * BITS / 32 + 8];   // the sweep's compressed bytes (also the sequential decoder's staging area and the header copy)
    uint32_t qa[Q];                       // position relative to the sweep's first output symbol (17 bits) | (distance - 1) << 17
    uint8_t ql[Q];                        // length - 3          (a counting kernel, MODE 1, never touches the queue: Q = 1)
};

// returns the number of lanes whose sub-sequences were decoded (0: nothing done, position untouched)
// what orders a wave's stores of output symbols in front of its later loads of them: a fence that waits for memory -- or, where
// `out` is an LDS image (OLDS: za_k_inflate_serial_small), nothing but the compiler's order: one wave's LDS accesses execute in order
template <bool OLDS> __device__ __forceinline__ void za_out_fence()
{
    if (OLDS) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __threadfence_block();
}

// (r06, za_k_inflate_serial_small) A SECOND wavefront runs the counting passes of the NEXT sweep while the first one stores and
// resolves the current one -- a sweep's passes need nothing of the sweep in front of it but where it ended.  ROLE 0: everything
// on the calling wave (all other kernels); ROLE 1: the main wave -- takes over the helper's result where it has one for this very
// position, asks for the next sweep in front of its storing pass and waits for it behind the resolution; ROLE 2: the helper --
// staging and counting passes into the buffer it was told, results to H, nothing else.  The two talk through LDS words (go /
// done sequence numbers); the second staging area lives in H.
struct ZaSweepHelp {
    uint32_t stage2[64 * 1024 / 32 + 8];
    uint32_t start[64], endp[64], cnt[64], nm[64];
    int st[64];
    uint64_t bitpos;            // what the result below is for
    uint32_t valid, buf;        // a result is there; the staging area it lies in (0: P->stage, 1: stage2)
    int nvalid, its;
    uint64_t req_bitpos; uint32_t req_buf, cmd;      // request: 1 run, 2 exit
    uint32_t go, done;          // sequence numbers
};

template <int MODE, typename SymT, typename PB, typename TT, bool OLDS = false, int ROLE = 0>
__device__ int za_par_sweep(const uint8_t *__restrict__ in, uint64_t in_len, const uint8_t *__restrict__ dict, uint32_t dict_len,
                            SymT *__restrict__ out, uint64_t out_cap, const TT &T, PB *P,
                            uint64_t &bitpos, uint64_t &op, uint32_t hist, uint32_t *far_io, bool &eob, ZaSweepHelp *H = nullptr)
{
    const int lane = za_lane();
    const uint64_t in_bits = in_len * 8ull;
    if (ROLE == 2) H->valid = 0u;                 // (uniform: every early return below leaves "no result")
    // Near the end of the input fewer lanes take part (L of them, sub-sequences of ZA_PS_MINBITS bits) instead of none: the last
    // 520 bytes of every member went through the sequential rounds, 65 of them (ZA_PS_SHORT_TAIL_LANES: the fewest lanes worth a sweep).
#ifndef ZA_PS_SHORT_TAIL_LANES
#define ZA_PS_SHORT_TAIL_LANES 12u
#endif
    if (bitpos + (uint64_t)ZA_PS_SHORT_TAIL_LANES * ZA_PS_MINBITS + 64ull > in_bits || out_cap - op < 256ull) return 0;
    const uint64_t room_bits = in_bits - bitpos - 64ull;
    uint32_t S = (uint32_t)((room_bits >> 6) > (uint64_t)PB::kBits ? (uint64_t)PB::kBits : (room_bits >> 6));
    uint32_t L = 64;                                                        // lanes that have a sub-sequence
    if (S < (uint32_t)ZA_PS_MINBITS) { S = ZA_PS_MINBITS; L = (uint32_t)(room_bits / ZA_PS_MINBITS); }      // ZA_PS_SHORT_TAIL_LANES <= L < 64
    const uint64_t sbyte = (bitpos >> 3) & ~3ull;
    const uint32_t b0 = (uint32_t)(bitpos - sbyte * 8ull);                 // 0..31
    const uint32_t ndw = ((b0 + L * S + 48u) >> 5) + 5u;                    // fits the staging area: a lane reads up to 64 bits ahead
    bool pre = false;                                                       // (ROLE 1) the helper has counted this very sweep
    uint32_t cur_buf = 0;                                                   // the staging area this sweep's bytes lie in
    if (ROLE == 1) {
        pre = H->valid != 0u && H->bitpos == bitpos;
        cur_buf = pre ? H->buf : 0u;
    }
    if (ROLE == 2) cur_buf = H->req_buf;
    uint32_t *const stg = cur_buf ? H->stage2 : P->stage;
    __builtin_amdgcn_wave_barrier();
    if (!pre)
    for (uint32_t i = (uint32_t)lane; i < ndw; i += 64) {
        const uint64_t o = sbyte + 4ull * i;
        uint32_t v = 0;
        if (o + 4 <= in_len + 8) v = za_ld32(in + o);                       // the buffer is padded by >= 8 bytes
        else for (int k = 0; k < 4; k++) if (o + (unsigned)k < in_len + 8) v |= (uint32_t)in[o + (unsigned)k] << (8 * k);
        stg[i] = v;
    }
    __builtin_amdgcn_wave_barrier();

    // one lane, one sub-sequence: tokens from bit `from` up to the first boundary >= `lim`.  The lane keeps 33..64 bits of
    // the stream in a register pair and takes one more dword from the staged bytes whenever 32 or fewer are left: a
    // literal / length code with its extra bits needs 20 at most, a distance code with its extra bits 28.
    uint32_t endp = 0, cnt = 0, nm = 0;
    int st = 0;                                 // 0 ran to the limit, 1 end of block, 2 invalid code, 3 no start
    int farrel = -(1 << 30);                    // (counting kernel) max over the lane's matches of distance - symbols before it
    bool bad_dist = false;                      // (storing pass) a distance reaches before the history
    uint32_t far_seen = 0;                      // (storing pass) furthest reach before this stream's first symbol
    auto run = [&](const bool emit, uint32_t from, uint32_t lim, uint64_t obase, uint32_t rbase, uint32_t qb) {
        uint32_t bp = from, c = 0, k = 0;
        int s = 0, fr = -(1 << 30);
        uint32_t w = from >> 5;
        const uint32_t sh0 = from & 31u;
        uint64_t bb = ((((uint64_t)stg[w + 1]) << 32) | (uint64_t)stg[w]) >> sh0;
        uint32_t nb = 64u - sh0;
        w += 2;
        while (bp < lim) {
            if (nb <= 32u) { bb |= (uint64_t)stg[w] << nb; nb += 32u; w++; }
            uint32_t e = T.lut_l[(uint32_t)bb & ((1u << TT::kLBits) - 1u)];
            if (!e) e = za_long_decode((uint32_t)bb, T.cnt_l, T.fst_l, T.idx_l, T.sym_l, TT::kLBits + 1);
            if (!e) { s = 2; break; }
            const int sym = (int)(e >> 4);
            uint32_t used = e & 15u;
            if (sym < 256) {
                if (emit) out[obase + c] = (SymT)sym;
                c++; bp += used; bb >>= used; nb -= used;
                continue;
            }
            if (sym == 256) { s = 1; bp += used; break; }
            const int ls = sym - 257;
            if (ls > 28) { s = 2; break; }
            int nx;
            int len = za_len_base(ls, nx);
            len += (int)((uint32_t)(bb >> used) & ((1u << nx) - 1u));
            used += (uint32_t)nx;
            bp += used; bb >>= used; nb -= used;
            if (nb <= 32u) { bb |= (uint64_t)stg[w] << nb; nb += 32u; w++; }
            e = T.lut_d[(uint32_t)bb & ((1u << TT::kDBits) - 1u)];
            if (!e) e = za_long_decode((uint32_t)bb, T.cnt_d, T.fst_d, T.idx_d, T.sym_d, TT::kDBits + 1);
            const int ds = (int)(e >> 4);
            if (!e || ds >= 30) { s = 2; break; }
            used = e & 15u;
            int dist = za_dist_base(ds, nx);
            dist += (int)((uint32_t)(bb >> used) & ((1u << nx) - 1u));
            used += (uint32_t)nx;
            bp += used; bb >>= used; nb -= used;
            if (emit) {
                P->qa[qb + k] = (rbase + c) | ((uint32_t)(dist - 1) << 17);
                P->ql[qb + k] = (uint8_t)(len - 3);
                const uint64_t at = obase + c;                               // symbols of this stream in front of the match
                if ((uint64_t)dist > at) {
                    if ((uint64_t)dist > at + (uint64_t)hist) bad_dist = true;
                    const uint32_t f = (uint32_t)((uint64_t)dist - at);
                    if (f > far_seen) far_seen = f;
                }
            }
            else if (MODE == 1) { const int f = dist - (int)c; if (f > fr) fr = f; }     // nothing is stored later: validity is checked from this
            c += (uint32_t)len; k++;
        }
        if (!emit) { endp = bp; cnt = c; nm = k; st = s; farrel = fr; }
    };

    // The counting passes (what a sweep mostly consists of: 4.3 of them, each as long as its slowest lane) take the tokens in
    // ROUNDS, like phase A of za_k_inflate_members: up to six literals and the token behind them on one straight path, looked
    // up in a 128-bit window of the staged stream that is fetched once per round (five LDS reads issued together) and moved
    // with v_alignbit -- no 64-bit shifts, no refill test per symbol, no branch per symbol kind.  Codes longer than the
    // first-level table (entry 0) and invalid ones take the bit-serial decode in the token stage; a literal run ends in front
    // of them.  Same results as run(false, ...): symbols that start before `lim` are counted, nothing else.
    auto run_count = [&](uint32_t from, uint32_t lim) {
        uint32_t bp = from, c = 0, k = 0;
        int s = 0, fr = -(1 << 30);
        auto lit3 = [&](uint32_t win, uint32_t room, bool on, uint32_t &bits) -> uint32_t {
            uint32_t u = 0, n = 0;
            bool l = on;
#pragma unroll
            for (int i = 0; i < ZA_PS_LITS; i++) {
                const uint32_t e = T.lut_l[__builtin_amdgcn_ubfe(win, u, TT::kLBits)];
                l = l && (e - 1u) < 0xFFFu && u < room;                     // assigned and a literal ((symbol << 4) | length, symbol < 256) that starts before the limit
                u += l ? (e & 15u) : 0u;
                n += l ? 1u : 0u;
            }
            bits = u;
            return n;
        };
#ifdef ZA_PS_STATS
        const unsigned long long cy0 = clock64();
        uint32_t wave_rounds = 0;
#endif
#pragma unroll 1
        while (s == 0 && bp < lim) {
#ifdef ZA_PS_STATS
            wave_rounds++;
#endif
            const uint32_t w = bp >> 5, sh = bp & 31u;
            const uint32_t d0 = stg[w], d1 = stg[w + 1], d2 = stg[w + 2], d3 = stg[w + 3], d4 = stg[w + 4];
            const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
            const uint32_t h2 = __builtin_amdgcn_alignbit(d3, d2, sh), h3 = __builtin_amdgcn_alignbit(d4, d3, sh);
            const uint32_t room = lim - bp;                                 // > 0
            uint32_t u1, u2;
            const uint32_t n1 = lit3(lo, room, true, u1);                   // first-level codes are <= 10 bits: three fit the 32 at hand
            const uint32_t lo1 = __builtin_amdgcn_alignbit(hi, lo, u1), hi1 = __builtin_amdgcn_alignbit(h2, hi, u1), h21 = __builtin_amdgcn_alignbit(h3, h2, u1);
#if ZA_PS_LIT_GROUPS == 1
            const uint32_t n2 = 0; u2 = 0;
#else
            const bool more = n1 == (uint32_t)ZA_PS_LITS && u1 < room;
            const uint32_t n2 = lit3(lo1, more ? room - u1 : 0u, more, u2);
#endif
            const uint32_t u = u1 + u2;                                     // <= 60
            c += n1 + n2;
            // -- the token behind them, whatever it is (a seventh literal, a literal with a long code, end of block, a match)
            const bool tok = u < room;
            const uint32_t m_lo = __builtin_amdgcn_alignbit(hi1, lo1, u2), m_hi = __builtin_amdgcn_alignbit(h21, hi1, u2);
            uint32_t em = T.lut_l[m_lo & ((1u << TT::kLBits) - 1u)];
            if (tok && em == 0u) em = za_long_decode(m_lo, T.cnt_l, T.fst_l, T.idx_l, T.sym_l, TT::kLBits + 1);
            const uint32_t sym = em >> 4, l = em & 15u;
            const bool is_len = tok && sym > 256u && sym <= 285u;
            int nx = 0, dnx = 0;
            const int lbase = za_len_base(is_len ? (int)sym - 257 : 0, nx);
            const uint32_t len = (uint32_t)lbase + __builtin_amdgcn_ubfe(m_lo, l, (uint32_t)nx);
            const uint32_t used = l + (uint32_t)nx;                          // <= 20
            const uint32_t x = __builtin_amdgcn_alignbit(m_hi, m_lo, used);  // the distance code and its extra bits: <= 28 bits
            uint32_t d = T.lut_d[x & ((1u << TT::kDBits) - 1u)];
            if (is_len && d == 0u) d = za_long_decode(x, T.cnt_d, T.fst_d, T.idx_d, T.sym_d, TT::kDBits + 1);
            const uint32_t ds = d >> 4, dl = d & 15u;
            const bool dist_ok = d != 0u && ds < 30u;
            const int dbase = za_dist_base(dist_ok ? (int)ds : 0, dnx);
            const uint32_t dist = (uint32_t)dbase + __builtin_amdgcn_ubfe(x, dl, (uint32_t)dnx);
            const bool bad = tok && (em == 0u || sym > 285u || (is_len && !dist_ok));
            if (bad) { s = 2; bp += u; }
            else if (tok) {
                uint32_t adv = l;
                if (sym < 256u) c++;
                else if (sym == 256u) s = 1;
                else {
                    if (MODE == 1) { const int f = (int)dist - (int)c; if (f > fr) fr = f; }
                    c += len; k++;
                    adv = used + dl + (uint32_t)dnx;
                }
                bp += u + adv;
            }
            else bp += u;
        }
        endp = bp; cnt = c; nm = k; st = s; farrel = fr;
#ifdef ZA_PS_STATS
        {
            uint32_t mx = 0;
            unsigned long long am = __ballot(1);
            const int first_lane = __builtin_ctzll(am);
            while (am) { const int j = __builtin_ctzll(am); am &= am - 1ull; const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)wave_rounds, j); mx = v > mx ? v : mx; }
            if (za_lane() == first_lane) {
                atomicAdd(&za_ps_stat[16], (unsigned long long)mx); atomicAdd(&za_ps_stat[17], (unsigned long long)(clock64() - cy0));
                atomicAdd(&za_ps_stat[19], 1ull);
            }
            atomicAdd(&za_ps_stat[20], (unsigned long long)wave_rounds);
            atomicAdd(&za_ps_stat[18], (unsigned long long)(k + 0u));
        }
#endif
    };

    uint32_t start = (uint32_t)lane < L ? b0 + (uint32_t)lane * S : ZA_PS_NONE;
    const uint32_t lim = b0 + ((uint32_t)lane + 1u) * S;
    bool dirty = true;
    int nvalid = (int)L;
    const unsigned long long t0 = ZA_STAT_T();
    int its = 0;
    if (pre) {
        // the helper's passes over this sweep: every lane's start, end, counts and state as its own passes would have left them
        start = H->start[lane]; endp = H->endp[lane]; cnt = H->cnt[lane]; nm = H->nm[lane]; st = H->st[lane];
        nvalid = H->nvalid; its = H->its;
        H->valid = 0u;
    } else
    for (int it = 1;; it++) {
        its = it;
        if (dirty) {
            if (start == ZA_PS_NONE) { st = 3; endp = 0; cnt = 0; nm = 0; farrel = -(1 << 30); }
            else {
#ifdef ZA_PS_OLD_COUNT
                run(false, start, lim, 0, 0, 0);
#else
                run_count(start, lim);
#endif
            }
        }
        const uint32_t handed = (uint32_t)lane < L ? (uint32_t)__shfl_up((int)(st == 0 ? endp : ZA_PS_NONE), 1, 64) : ZA_PS_NONE;     // (lanes without a sub-sequence stay out)
        const bool ch = lane > 0 && handed != start;
        if (lane > 0) start = handed;
        dirty = ch;
        const unsigned long long chm = __ballot(ch);
        if (!chm) break;
        if (it >= PB::kMaxIt) { nvalid = __builtin_ctzll(chm); break; }
    }
    if (ROLE == 2) {
        H->start[lane] = start; H->endp[lane] = endp; H->cnt[lane] = cnt; H->nm[lane] = nm; H->st[lane] = st;
        H->nvalid = nvalid; H->its = its; H->bitpos = bitpos; H->buf = cur_buf; H->valid = 1u;       // (uniform stores)
        return 1;
    }
    ZA_STAT_ADD(0, 1); ZA_STAT_ADD(2, its); ZA_STAT_ADD(3, pre ? 0ull : ZA_STAT_T() - t0); ZA_STAT_ADD(13, nvalid);
    // the chain ends at the first lane that did not run to its limit
    bool eob_hit = false;
    {
        const unsigned long long stopm = __ballot(st != 0) & (nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull));
        if (stopm) {
            const int j = __builtin_ctzll(stopm);
            if (__builtin_amdgcn_readlane(st, j) == 1) { nvalid = j + 1; eob_hit = true; }
            else nvalid = j;
        }
    }
    if (nvalid == 0) { ZA_STAT_ADD(6, 1); return 0; }
    const bool act0 = lane < nvalid;
    const uint32_t cinc = za_wave_incl_scan(act0 ? cnt : 0u), minc = za_wave_incl_scan(act0 ? nm : 0u);
    const uint32_t base = cinc - (act0 ? cnt : 0u), qb = minc - (act0 ? nm : 0u);
    {
        // a lane is kept if its output fits the buffer, the 17-bit positions and the queue (the counting kernel has no
        // queue, but checks here that no distance reaches before the history); the sweep ends in front of the first
        // lane that does not
        const bool fits = (uint64_t)cinc <= out_cap - op && cinc <= (1u << 17) && (MODE == 1 || minc <= (uint32_t)PB::kQ) &&
                          (MODE != 1 || (long long)farrel - (long long)(op + (uint64_t)base) <= (long long)hist);
        const unsigned long long badm = __ballot(act0 && !fits);
        if (badm) {
            const int j = __builtin_ctzll(badm);
            if (j < nvalid) { nvalid = j; eob_hit = false; }
        }
    }
    if (nvalid == 0) { ZA_STAT_ADD(6, 1); return 0; }
    ZA_STAT_ADD(1, nvalid);
    const bool act = lane < nvalid;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)cinc, nvalid - 1);
    const uint32_t M = (uint32_t)__builtin_amdgcn_readlane((int)minc, nvalid - 1);
    const uint32_t last_end = (uint32_t)__builtin_amdgcn_readlane((int)endp, nvalid - 1);
    bool posted = false;
    uint32_t seq = 0;
    auto wait_helper = [&]() {
        if (ROLE == 1 && posted) {
            while (__hip_atomic_load(&H->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
            posted = false;
        }
    };
    if (ROLE == 1 && !eob_hit) {
        // the next sweep starts where this one ends: the helper counts it while this wave stores and resolves
        H->req_bitpos = sbyte * 8ull + (uint64_t)last_end; H->req_buf = cur_buf ^ 1u; H->cmd = 1u;
        seq = H->go + 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&H->go, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        posted = true;
    }
    if (MODE != 1) {
        const unsigned long long t1 = ZA_STAT_T();
        if (act) run(true, start, lim, op + (uint64_t)base, base, qb);
        // a distance before the history is an error of the stream: nothing is committed, the sequential rounds find and
        // report it (what the storing pass wrote lies behind `op` and is overwritten)
        if (__ballot(bad_dist)) { ZA_STAT_ADD(6, 1); wait_helper(); if (ROLE == 1) H->valid = 0u; return 0; }
        if (far_io) {
            uint32_t fv = far_seen;
            for (int sft = 32; sft; sft >>= 1) { const uint32_t ov = (uint32_t)__shfl_xor((int)fv, sft, 64); fv = ov > fv ? ov : fv; }
            if (fv > *far_io) *far_io = fv;
        }
        za_out_fence<OLDS>();                   // literals (global) and the queue (LDS) are visible to the whole wave
        __builtin_amdgcn_wave_barrier();
        ZA_STAT_ADD(4, ZA_STAT_T() - t1);
        const unsigned long long t2 = ZA_STAT_T();
        // symbol at stream position q; q < 0 lies before the start: a marker (MODE 2) or a dictionary byte
        auto sym_at = [&](long long q) -> SymT {
            if (q >= 0) return out[q];
            return MODE == 2 ? (SymT)(256u + (uint32_t)(ZA_WIN + q)) : (SymT)dict[(long long)dict_len + q];
        };
        for (uint32_t g = 0; g < M; g += 64) {
            const bool has = g + (uint32_t)lane < M;
            uint32_t qa = 0, mlen = 0;
            if (has) { qa = P->qa[g + lane]; mlen = (uint32_t)P->ql[g + lane] + 3u; }
            const uint32_t mdst = qa & 0x1FFFFu, mdist = (qa >> 17) + 1u;
            bool done = !has;
            unsigned long long pending = __ballot(!done);
            // the matches of this group whose bytes mine reads are the lanes [jlo, jhi) (destinations ascend with the lane)
            const uint32_t sdst = has ? mdst : 0xFFFFFFFFu, send = has ? mdst + mlen : 0xFFFFFFFFu;
            const int src_a = (int)mdst - (int)mdist, src_b = src_a + (int)(mlen < mdist ? mlen : mdist);
            uint32_t jhi = 0, jlo = 0;
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1) {
                const uint32_t vd = (uint32_t)__shfl((int)sdst, (int)(jhi + step - 1u), 64);
                const uint32_t ve = (uint32_t)__shfl((int)send, (int)(jlo + step - 1u), 64);
                if ((long long)vd < (long long)src_b) jhi += step;
                if ((long long)ve <= (long long)src_a) jlo += step;
            }
            const unsigned long long deps = ((1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull);
            const uint64_t adst = op + (uint64_t)mdst;
            // short non-overlapping matches whose source is real output are copied by their own lane, 16 bytes per batch
            const uint32_t nbytes = mlen * (uint32_t)sizeof(SymT);
            const bool simple = has && mdist >= mlen && nbytes <= 64u && adst >= (uint64_t)mdist;
            const bool wide = (adst + (uint64_t)mlen) * sizeof(SymT) + 16ull <= out_cap * sizeof(SymT);      // (a 16-byte load behind the source stays inside the buffer)
            while (pending) {
                const bool ready = !done && (pending & deps) == 0ull;
                if (ready && simple) {
                    uint8_t *o = (uint8_t *)(out + adst);
                    const uint8_t *sp = (const uint8_t *)(out + (adst - (uint64_t)mdist));
                    // 16 bytes per load and store (64 lanes at 64 unrelated addresses: what an instruction costs the memory pipe is its
                    // lanes, not its bytes); a load may reach past the source's end into bytes of no meaning (never past the buffer:
                    // the source ends at or below the destination, which has `wide` bytes of room)
                    if (wide) {
                        for (uint32_t i = 0; i < nbytes; i += 16) {
                            const uint32_t rem = nbytes - i;
                            ZaU4u v = *(const ZaU4u *)(sp + i);
                            uint8_t *q = o + i;
                            if (rem >= 16u) *(ZaU4u *)q = v;
                            else {
                                if (rem & 8u) { ZaU2u t; t.x = v.x; t.y = v.y; *(ZaU2u *)q = t; q += 8; v.x = v.z; v.y = v.w; }
                                if (rem & 4u) { *(za_u32u *)q = v.x; q += 4; v.x = v.y; }
                                if (rem & 2u) { *(za_u16u *)q = (uint16_t)v.x; q += 2; v.x >>= 16; }
                                if (rem & 1u) *q = (uint8_t)v.x;
                            }
                        }
                    } else
                    for (uint32_t i = 0; i < nbytes; i += 16) {
                        const uint32_t rem = nbytes - i;
                        uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                        v0 = za_ld32(sp + i);
                        if (rem > 4) v1 = za_ld32(sp + i + 4);
                        if (rem > 8) v2 = za_ld32(sp + i + 8);
                        if (rem > 12) v3 = za_ld32(sp + i + 12);
                        const uint32_t vv[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int left = (int)rem - 4 * k;
                            if (left >= 4) *(za_u32u *)(o + i + 4 * k) = vv[k];
                            else if (left > 0) {
                                o[i + 4 * k] = (uint8_t)vv[k];
                                if (left > 1) o[i + 4 * k + 1] = (uint8_t)(vv[k] >> 8);
                                if (left > 2) o[i + 4 * k + 2] = (uint8_t)(vv[k] >> 16);
                            }
                        }
                    }
                }
                // The others by the whole wave.  The matches that are ready in one round do not read one another (that is what ready
                // means), so four of them at a time have their first 64 symbols LOADED before any is stored: one trip to memory for
                // four matches instead of four trips one behind the other (these matches are few, but each cost the wave a full
                // load-store round trip); what a match has beyond 64 symbols follows in blocks of 64.
                unsigned long long coop = __ballot(ready && !simple);
                while (coop) {
                    uint32_t cdv[4], clv[4], cdistv[4];
                    SymT val[4];
                    int cnt4 = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        cdv[u] = 0; clv[u] = 0; cdistv[u] = 1; val[u] = 0;
                        if (coop) {
                            const int j = __builtin_ctzll(coop);
                            coop &= coop - 1ull;
                            cdv[u] = (uint32_t)__builtin_amdgcn_readlane((int)mdst, j);
                            clv[u] = (uint32_t)__builtin_amdgcn_readlane((int)mlen, j);
                            cdistv[u] = (uint32_t)__builtin_amdgcn_readlane((int)mdist, j);
                            cnt4 = u + 1;
                        }
                    }
                    auto src_index = [&](uint32_t i, uint32_t cl, uint32_t cdist) -> int {
                        int k = (int)i;                     // byte i of a self-overlapping match is byte (i mod dist) of its period
                        if (cdist < cl) {
                            k = (int)i - (int)cdist * (int)((float)i * (1.0f / (float)cdist));
                            if (k < 0) k += (int)cdist;
                            if (k >= (int)cdist) k -= (int)cdist;
                        }
                        return k;
                    };
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (u < cnt4 && (uint32_t)lane < clv[u]) {
                            const long long dq = (long long)(op + (uint64_t)cdv[u]);
                            val[u] = sym_at(dq - (long long)cdistv[u] + src_index((uint32_t)lane, clv[u], cdistv[u]));
                        }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (u < cnt4 && (uint32_t)lane < clv[u]) out[(long long)(op + (uint64_t)cdv[u]) + lane] = val[u];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (u < cnt4 && clv[u] > 64u) {
                            const long long dq = (long long)(op + (uint64_t)cdv[u]);
                            for (uint32_t bs = 64; bs < clv[u]; bs += 64) {
                                const uint32_t i = bs + (uint32_t)lane;
                                if (i < clv[u]) out[dq + i] = sym_at(dq - (long long)cdistv[u] + src_index(i, clv[u], cdistv[u]));
                            }
                        }
                    }
                }
                za_out_fence<OLDS>();
                done = done || ready;
                pending = __ballot(!done);
                ZA_STAT_ADD(14, 1);
            }
        }
        ZA_STAT_ADD(5, ZA_STAT_T() - t2);
    }
    ZA_STAT_ADD(11, total); ZA_STAT_ADD(12, eob_hit ? 1 : 0); ZA_STAT_ADD(15, M);
    wait_helper();
    op += (uint64_t)total;
    bitpos = sbyte * 8ull + (uint64_t)last_end;
    if (eob_hit) eob = true;
    return nvalid;
}

// the helper wavefront's life (ROLE 2): the counting passes of the sweep it is asked for, into the staging area it is told, until it is sent home
template <typename SymT, typename PB, typename TT, bool OLDS>
__device__ void za_sweep_helper_loop(const uint8_t *__restrict__ in, uint64_t in_len, const uint8_t *__restrict__ dict, uint32_t dict_len,
                                     SymT *__restrict__ out, uint64_t out_cap, const TT &T, PB *P, ZaSweepHelp *H)
{
    uint32_t seen = 0;
    for (;;) {
        while (__hip_atomic_load(&H->go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == seen) __builtin_amdgcn_s_sleep(1);
        seen++;
        if (H->cmd == 2u) break;
        uint64_t bp = H->req_bitpos, opd = 0;
        bool eobd = false;
        (void)za_par_sweep<0, SymT, PB, TT, OLDS, 2>(in, in_len, dict, dict_len, out, out_cap, T, P, bp, opd, dict_len, nullptr, eobd, H);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&H->done, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// ... and the main wave sending it home (no request is open: every sweep waits for its helper before it returns)
__device__ __forceinline__ void za_sweep_helper_exit(ZaSweepHelp *H)
{
    H->cmd = 2u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(&H->go, H->go + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ------------------------------------------------------------------------------------------------
// sequential decoder: one wave per stream
// ------------------------------------------------------------------------------------------------
// Wave-uniform sequential RFC 1951 decode of one stream (all 64 lanes call with identical arguments).
// T / win / scratch / ibuf are the caller's LDS; returns a ZA_I_* status, bits consumed and bytes produced.
//   MODE 0  normal: SymT = u8, history ring `win` (32768 entries) and `out` hold bytes
//   MODE 1  count only: nothing is stored (win / out unused); sizes a chunk and validates it
//   MODE 2  markers: SymT = u16; the ring starts as 256 + j (j = index into the unknown previous 32 KiB), copies
//           move symbols, so every output symbol is a byte (< 256) or names the byte of the previous window it
//           equals; *max_back = farthest distance before the chunk start that was referenced
// hist = bytes of history that may be referenced before out[0] (dict_len, or 32768 for a chunk in mid-stream);
// P = LDS of the parallel sweeps (za_par_sweep), nullptr = sequential rounds only; with P, `ibuf` is P->stage;
// stop_at_sync ends the decode right after an empty stored block (sync-flush point) with ZA_I_SYNC.
#define ZA_I_SYNC 2
// MODE 0: bytes, 32 KiB ring in LDS.  MODE 1: count only.  MODE 2: 16-bit symbols (markers for bytes before the
// start), RING symbols in LDS; older sources are read back from `out` (written many rounds ago) or are markers.
template <int MODE, typename SymT, int RING = ZA_WIN, typename PB = ZaParBufT<1024, 3072>, typename TT = ZaInfTabs, bool OLDS = false, bool HELPED = false>
__device__ int za_inflate_serial_core(const uint8_t *__restrict__ in, uint64_t in_len,
                                      const uint8_t *__restrict__ dict, uint32_t dict_len,
                                      SymT *__restrict__ out, uint64_t out_cap,
                                      TT &T, SymT *win, int *scratch, uint32_t *ibuf, uint64_t &bits_used, uint64_t &out_len,
                                      uint32_t start_bit = 0, uint64_t *blk_bits = nullptr, uint64_t *blk_out = nullptr,
                                      uint32_t hist = 0xFFFFFFFFu, bool stop_at_sync = false, uint32_t *max_back = nullptr,
                                      const uint64_t *__restrict__ stops = nullptr, uint32_t nstops = 0, uint64_t abs_bit0 = 0,
                                      PB *P = nullptr, ZaSweepHelp *H = nullptr)
{
    const int lane = za_lane();
    const uint64_t in_bits = in_len * 8ull;
    uint64_t bitpos = start_bit, op = 0;
    uint64_t cp_bits = start_bit, cp_out = 0;
    uint32_t far = 0;
    int status = ZA_I_OK;
    bool ring_stale = false;       // a parallel sweep wrote `out` only: the LDS ring is refreshed before the next sequential round
    int par_wait = 0;              // sequential rounds to go before the next parallel sweep is tried
    int short_run = 0;             // sweeps in a row that kept fewer than 16 lanes
    if (hist == 0xFFFFFFFFu) hist = dict_len;
    if (MODE == 0) for (uint32_t i = (uint32_t)lane; i < dict_len; i += 64) if (dict_len - i <= (uint32_t)RING) win[(ZA_WIN - dict_len + i) & (RING - 1)] = (SymT)dict[i];
    // MODE 2: the ring starts out holding the markers of the RING positions before the start
    if (MODE == 2) for (uint32_t i = (uint32_t)lane; i < (uint32_t)RING; i += 64) win[i] = (SymT)(256u + (uint32_t)(ZA_WIN - RING) + i);
    // symbol at position q (may be negative: before the start) that is no longer in a small ring: read back from the
    // output (written many rounds ago); before the start it is a marker (MODE 2) or a dictionary byte (MODE 0)
    auto far_sym = [&](long long q) -> SymT {
        if (q >= 0) return out[q];
        return MODE == 2 ? (SymT)(256u + (uint32_t)(ZA_WIN + q)) : (SymT)dict[(long long)dict_len + q];
    };
    za_wave_sync();

    for (;;) {
        cp_bits = bitpos; cp_out = op;          // a decoder can restart here with the last 32 KiB of output as dictionary
        if (nstops && bitpos != start_bit) {
            // chunk mode: stop at a block boundary that is a listed chunk start (binary search, uniform)
            const uint64_t a = abs_bit0 + bitpos;
            uint32_t lo = 0, hi = nstops;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (stops[mid] < a) lo = mid + 1; else hi = mid; }
            if (lo < nstops && stops[lo] == a) { status = ZA_I_SYNC; break; }
        }
        if (bitpos + 3 > in_bits) { status = ZA_I_INPUT; break; }
        uint64_t bits = za_peek(in, bitpos);
        const int last = (int)(bits & 1u), type = (int)((bits >> 1) & 3u);
        bitpos += 3;
        if (type == 3) { status = ZA_I_DATA; break; }
        if (type == 0) {
            bitpos = (bitpos + 7ull) & ~7ull;
            if (bitpos + 32 > in_bits) { status = ZA_I_INPUT; break; }
            bits = za_peek(in, bitpos);
            const uint32_t len = (uint32_t)(bits & 0xFFFFu), nlen = (uint32_t)((bits >> 16) & 0xFFFFu);
            if ((len ^ 0xFFFFu) != nlen) { status = ZA_I_DATA; break; }
            bitpos += 32;
            uint32_t can = len;
            const uint64_t availb = (in_bits - bitpos) >> 3;
            bool short_in = false, short_out = false;
            if ((uint64_t)can > availb) { can = (uint32_t)availb; short_in = true; }
            if ((uint64_t)can > out_cap - op) { can = (uint32_t)(out_cap - op); short_out = true; short_in = false; }
            if (MODE != 1) {
                const uint8_t *src = in + (bitpos >> 3);
                for (uint32_t i = (uint32_t)lane; i < can; i += 64) {
                    const SymT b = (SymT)src[i];
                    win[(op + i) & (RING - 1)] = b;
                    out[op + i] = b;
                }
            }
            op += can; bitpos += 8ull * can;
            if (short_out) { status = ZA_I_OUTFULL; break; }
            if (short_in) { status = ZA_I_INPUT; break; }
            if (stop_at_sync && len == 0 && !last) { status = ZA_I_SYNC; break; }
        } else {
            const unsigned long long tt = ZA_STAT_T();
            status = za_read_tables(in, in_bits, bitpos, type, T, scratch, P ? ibuf : nullptr);     // with P, ibuf is the large staging area
            ZA_STAT_ADD(9, ZA_STAT_T() - tt); ZA_STAT_ADD(8, 1);
            if (status != ZA_I_OK) break;
            // Symbol loop.  The bitstream is staged through LDS (512 bytes at a time).  Per round: one 64-bit
            // window W at bitpos; lane l decodes, in one LDS round trip, the literal/length code and the
            // distance code that would start at bit l of W; a scalar walker then follows the real token chain
            // through the lanes' results with readlane, consuming every token that lies completely inside W.
            // A token is at most 48 bits, so the first one of a window always fits: each round makes progress.
            bool eob = false;
            uint64_t ibase = ~0ull;                 // byte offset of ibuf[0] (multiple of 4); ~0 = nothing staged
            while (!eob && status == ZA_I_OK) {
                if (P != nullptr) {
                    if (par_wait == 0) {
                        const int got = za_par_sweep<MODE, SymT, PB, TT, OLDS, HELPED ? 1 : 0>(in, in_len, dict, dict_len, out, out_cap, T, P, bitpos, op, hist,
                                                                     max_back ? &far : nullptr, eob, H);
                        ibase = ~0ull;                                   // the staging area was used by the sweep
                        // A sweep that kept few lanes (the passes had not settled, the queue was full) is followed by another sweep
                        // at once; only the second such sweep in a row sends the decoder to 32 sequential rounds (data on which the
                        // sub-sequences do not find their way costs a sweep five times what the same bits cost token by token).
                        // One that kept few lanes because the block ended in it says nothing about what follows.  Before: 32 rounds
                        // after every short sweep -- 69 sequential rounds per zlib-written member, 7 % of the kernel.
                        if (got > 0) {
                            ring_stale = true;
                            if (got < 16 && !eob) { if (++short_run >= 2) par_wait = 32; }
                            else short_run = 0;
                            if (HELPED && (eob || par_wait != 0)) H->valid = 0u;      // (no sweep follows at once: the staging areas are the sequential rounds' and the header's)
                            continue;
                        }
                        if (HELPED) H->valid = 0u;
                        par_wait = 32;
                    } else par_wait--;
                    if (MODE != 1 && ring_stale) {
                        // the last RING symbols go back into the LDS ring (positions before the start: dictionary / markers)
                        za_out_fence<OLDS>();
                        // (MODE 0: nothing can refer to what lies in front of the dictionary -- a stream of 1 KiB refills 1 KiB, not 32)
                        uint32_t i0 = 0;
                        if (MODE == 0 && op + (uint64_t)dict_len < (uint64_t)RING) i0 = (uint32_t)((uint64_t)RING - op - (uint64_t)dict_len) & ~63u;
                        for (uint32_t i = i0 + (uint32_t)lane; i < (uint32_t)RING; i += 64) {
                            const long long q = (long long)op - (long long)RING + (long long)i;
                            SymT v;
                            if (q >= 0) v = out[q];
                            else if (MODE == 2) v = (SymT)(256u + (uint32_t)(ZA_WIN + q));
                            else v = (-q <= (long long)dict_len) ? (SymT)dict[(long long)dict_len + q] : (SymT)0;
                            win[(uint64_t)q & (uint64_t)(RING - 1)] = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                        ring_stale = false;
                        ZA_STAT_ADD(10, 1);
                    }
                }
                ZA_STAT_ADD(7, 1);
                const uint64_t byte = bitpos >> 3;
                if (ibase == ~0ull || byte < ibase || byte + 24 > ibase + 4ull * ZA_IBUF_DW) {
                    ibase = byte & ~3ull;
                    __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < ZA_IBUF_DW; i += 64) {
                        const uint64_t o = ibase + 4ull * (unsigned)i;
                        uint32_t v = 0;
                        if (o + 4 <= in_len + 8) v = za_ld32(in + o);               // the buffer is padded by >= 8 bytes
                        else for (int k = 0; k < 4; k++) if (o + (unsigned)k < in_len + 8) v |= (uint32_t)in[o + (unsigned)k] << (8 * k);
                        ibuf[i] = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // ---- fast round: every lane decodes the COMPLETE token (literal, end of block, or length + distance
                // with their extra bits: at most 48 bits) that would start at bit `lane`; a short scalar loop then
                // follows the real chain 0 -> next -> ... through the lanes (one readlane per token) and everything
                // on the chain is emitted together: literals with one store, matches one after the other.  A round
                // consumes at least 64 bits.  Windows that touch the end of the input or of the output, codes longer
                // than the LUT and invalid data are left to the careful walker below, which also produces the status.
                if (bitpos + 128ull <= in_bits && out_cap - op >= 64ull * 258ull) {
                    const uint32_t relbit = (uint32_t)(bitpos - ibase * 8ull) + (uint32_t)lane;
                    const uint32_t fw = relbit >> 5, fsh = relbit & 31u;
                    const uint64_t flo = ((uint64_t)ibuf[fw + 1] << 32) | ibuf[fw];
                    const uint64_t mine = fsh ? ((flo >> fsh) | ((uint64_t)ibuf[fw + 2] << (64u - fsh))) : flo;
                    const uint32_t eL = T.lut_l[mine & ((1u << TT::kLBits) - 1u)];
                    const uint32_t l = eL & 15u, sym = eL >> 4;
                    uint32_t nk = (uint32_t)lane + l;            // (kind << 8) | offset of the next token; kind 0 literal
                    uint32_t olen = 1, dist = 0;
                    if (eL == 0) nk = 0xFFFFu;                   // not in the LUT: careful walker
                    else if (sym == 256u) { nk |= 2u << 8; olen = 0; }
                    else if (sym > 256u) {
                        const int ls = (int)sym - 257;
                        if (ls >= 29) nk = 0xFFFFu;
                        else {
                            int nx;
                            int len = za_len_base(ls, nx);
                            len += (int)((mine >> l) & ((1u << nx) - 1u));
                            const uint32_t o2 = l + (uint32_t)nx;                        // <= 20
                            const uint32_t eD = T.lut_d[(mine >> o2) & ((1u << TT::kDBits) - 1u)];
                            const int ds = (int)(eD >> 4);
                            if (eD == 0 || ds >= 30) nk = 0xFFFFu;
                            else {
                                int dnx;
                                int d = za_dist_base(ds, dnx);
                                const uint32_t o3 = o2 + (eD & 15u);                     // <= 35
                                d += (int)((mine >> o3) & ((1u << dnx) - 1u));
                                nk = (1u << 8) | ((uint32_t)lane + o3 + (uint32_t)dnx);  // offset <= 63 + 48
                                olen = (uint32_t)len; dist = (uint32_t)d;
                            }
                        }
                    }
                    uint64_t vis = 0;
                    uint32_t fo = 0;
                    bool feob = false;
                    for (;;) {
                        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)nk, (int)fo);
                        if (n == 0xFFFFu) break;
                        vis |= 1ull << fo;
                        fo = n & 0xFFu;
                        if ((n >> 8) == 2u) { feob = true; break; }
                        if (fo >= 64u) break;
                    }
                    if (vis) {
                        bool on = (vis >> lane) & 1ull;
                        uint32_t incl = za_wave_incl_scan(on ? olen : 0u);
                        uint32_t pre = incl - (on ? olen : 0u);
                        const bool ismatch = (nk >> 8) == 1u && nk != 0xFFFFu;
                        // a distance that reaches before the start of the history ends the round in front of it
                        const uint64_t bad = __ballot(on && ismatch && (uint64_t)dist > op + (uint64_t)pre + (uint64_t)hist);
                        if (bad) {
                            const uint32_t b = (uint32_t)__builtin_ctzll(bad);
                            vis &= (1ull << b) - 1ull;
                            fo = b; feob = false;
                            on = (vis >> lane) & 1ull;
                        }
                        if (vis) {
                            uint32_t total = bad ? (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)fo)
                                                 : (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            // Literals are put into the 32 KiB ring before the matches are copied: a literal behind a
                            // match whose distance is within this round's output of 32 KiB would overwrite bytes that
                            // match still has to read.  Such a match ends the round (or is a round of its own).
                            // (with a small ring a match that far back reads `out` instead; then only the round's
                            // output is bounded so that whatever is read there was written in an earlier round)
                            const uint64_t hz = RING == ZA_WIN ? __ballot(on && ismatch && dist + (total - pre) > (uint32_t)ZA_WIN)
                                                               : __ballot(on && pre + olen > (uint32_t)(RING / 4));
                            if (hz) {
                                const uint32_t h = (uint32_t)__builtin_ctzll(hz);
                                feob = false;
                                if (h == 0) {
                                    vis = 1ull; fo = (uint32_t)__builtin_amdgcn_readlane((int)nk, 0) & 0xFFu;
                                    total = (uint32_t)__builtin_amdgcn_readlane((int)olen, 0);
                                } else {
                                    vis &= (1ull << h) - 1ull; fo = h;
                                    total = (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)h);
                                }
                                on = (vis >> lane) & 1ull;
                            }
                            if (max_back && op < (uint64_t)ZA_WIN) {
                                uint32_t fv = (on && ismatch && (uint64_t)dist > op + pre) ? (uint32_t)((uint64_t)dist - op - pre) : 0u;
                                for (int sft = 32; sft; sft >>= 1) { const uint32_t ov = (uint32_t)__shfl_xor((int)fv, sft, 64); fv = ov > fv ? ov : fv; }
                                if (fv > far) far = fv;
                            }
                            if (MODE != 1) {
                                if (on && !ismatch && olen) {
                                    const uint64_t q = op + (uint64_t)pre;
                                win[q & (RING - 1)] = (SymT)sym; out[q] = (SymT)sym;
                                }
                                __builtin_amdgcn_wave_barrier();
                                uint64_t mm = __ballot(on && ismatch);
                                while (mm) {
                                    const int ml = __builtin_ctzll(mm);
                                    mm &= mm - 1ull;
                                    const int len = __builtin_amdgcn_readlane((int)olen, ml);
                                    const int d = __builtin_amdgcn_readlane((int)dist, ml);
                                    const uint32_t mpre = (uint32_t)__builtin_amdgcn_readlane((int)pre, ml);
                                    const uint64_t mp = op + (uint64_t)mpre;
                                    if (RING != ZA_WIN && (uint32_t)d + (total - mpre) > (uint32_t)RING) {
                                        // source older than the ring (d > 3/4 RING, so d > len: no overlap)
                                        for (int base = 0; base < len; base += 64) {
                                            const int i = base + lane;
                                            if (i < len) {
                                                const SymT b = far_sym((long long)mp - d + i);
                                                win[(mp + (uint64_t)i) & (RING - 1)] = b; out[mp + (uint64_t)i] = b;
                                            }
                                        }
                                    } else if (d >= len) {
                                        for (int base = 0; base < len; base += 64) {
                                            const int i = base + lane;
                                            if (i < len) {
                                                const SymT b = win[(mp - (uint64_t)d + (uint64_t)i) & (RING - 1)];
                                                win[(mp + (uint64_t)i) & (RING - 1)] = b; out[mp + (uint64_t)i] = b;
                                            }
                                        }
                                    } else {
                                        // overlapping copy: byte i comes from position i mod d of the period; the period
                                        // itself lies completely before the match, so every lane can read at once
                                        const float rd = 1.0f / (float)d;
                                        for (int base = 0; base < len; base += 64) {
                                            const int i = base + lane;
                                            if (i < len) {
                                                int k = i - d * (int)((float)i * rd);
                                                if (k < 0) k += d;
                                                if (k >= d) k -= d;
                                                const SymT b = win[(mp - (uint64_t)d + (uint64_t)k) & (RING - 1)];
                                                out[mp + (uint64_t)i] = b;
                                                __builtin_amdgcn_wave_barrier();
                                                win[(mp + (uint64_t)i) & (RING - 1)] = b;
                                            }
                                        }
                                    }
                                    __builtin_amdgcn_wave_barrier();
                                }
                            }
                            op += (uint64_t)total;
                            bitpos += (uint64_t)fo;
                            if (feob) eob = true;
                            continue;
                        }
                    }
                }
                const uint32_t rel = (uint32_t)(byte - ibase), w = rel >> 2, sh = (rel & 3u) * 8u + (uint32_t)(bitpos & 7u);
                const uint64_t lo64 = ((uint64_t)ibuf[w + 1] << 32) | ibuf[w];
                const uint64_t Wv = sh ? ((lo64 >> sh) | ((uint64_t)ibuf[w + 2] << (64 - sh))) : lo64;
                const uint64_t W = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(Wv >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)Wv);
                // speculative decode at bit offset `lane`; LUT only: a code longer than the LUT (rare) is resolved by
                // the walker when it is really met
                const uint64_t mine = W >> lane;
                const uint32_t eL = T.lut_l[mine & ((1u << TT::kLBits) - 1u)];
                const uint32_t eD = T.lut_d[mine & ((1u << TT::kDBits) - 1u)];
                uint32_t o = 0;
                for (;;) {
                    if (o > 64u - 15u) break;                                        // next code may be cut: new window
                    uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)eL, (int)o);
                    if (!e) e = za_long_decode((uint32_t)(W >> o), T.cnt_l, T.fst_l, T.idx_l, T.sym_l, TT::kLBits + 1);         // o <= 49: 15 bits are there
                    if (!e) { status = (bitpos + o + 15 > in_bits) ? ZA_I_INPUT : ZA_I_DATA; break; }
                    int sym = (int)(e >> 4);
                    const uint32_t l = e & 15u;
                    if (bitpos + o + l > in_bits) { status = ZA_I_INPUT; break; }
                    if (sym < 256) {
                        if (op >= out_cap) { status = ZA_I_OUTFULL; break; }
                        if (MODE != 1 && lane == 0) { win[op & (RING - 1)] = (SymT)sym; out[op] = (SymT)sym; }
                        op++; o += l;
                        continue;
                    }
                    if (sym == 256) { o += l; eob = true; break; }
                    sym -= 257;
                    if (sym >= 29) { status = ZA_I_DATA; break; }
                    int nx;
                    int len = za_len_base(sym, nx);
                    const uint32_t o2 = o + l + (uint32_t)nx;
                    if (o2 > 64u - 15u) break;                                       // distance code may be cut: new window
                    if (nx) len += (int)((W >> (o + l)) & ((1u << nx) - 1u));       // o + l < 64 here
                    uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)eD, (int)o2);
                    if (!e2) e2 = za_long_decode((uint32_t)(W >> o2), T.cnt_d, T.fst_d, T.idx_d, T.sym_d, TT::kDBits + 1);
                    if (!e2) { status = (bitpos + o2 + 15 > in_bits) ? ZA_I_INPUT : ZA_I_DATA; break; }
                    const int ds = (int)(e2 >> 4);
                    if (ds >= 30) { status = ZA_I_DATA; break; }
                    int dnx;
                    int dist = za_dist_base(ds, dnx);
                    const uint32_t o3 = o2 + (e2 & 15u) + (uint32_t)dnx;
                    if (o3 > 64u) break;                                             // extra bits cut: new window
                    if (dnx) dist += (int)((W >> (o2 + (e2 & 15u))) & ((1u << dnx) - 1u));   // shift < 64 when dnx > 0
                    if (bitpos + o3 > in_bits) { status = ZA_I_INPUT; break; }
                    if ((uint64_t)dist > op + hist) { status = ZA_I_DATA; break; }
                    if ((uint64_t)dist > op && (uint32_t)((uint64_t)dist - op) > far) far = (uint32_t)((uint64_t)dist - op);
                    bool short_out = false;
                    if ((uint64_t)len > out_cap - op) { len = (int)(out_cap - op); short_out = true; }
                    if (MODE != 1) {
                        for (int base = 0; base < len; base += 64) {
                            const int i = base + lane;
                            SymT b = 0;
                            if (i < len) {
                                const int k = dist < len ? i % dist : i;
                                if (RING != ZA_WIN && dist > RING) b = far_sym((long long)op - dist + k);
                                else b = win[(op - (uint64_t)dist + (uint64_t)k) & (RING - 1)];
                            }
                            __builtin_amdgcn_wave_barrier();
                            if (i < len) { win[(op + (uint64_t)i) & (RING - 1)] = b; out[op + (uint64_t)i] = b; }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    op += (uint64_t)len;
                    if (short_out) { status = ZA_I_OUTFULL; break; }
                    o = o3;
                }
                bitpos += o;
            }
            if (status != ZA_I_OK) break;
        }
        if (last) { status = ZA_I_END; break; }
    }
    bits_used = bitpos; out_len = op;
    if (blk_bits) *blk_bits = cp_bits;
    if (blk_out) *blk_out = cp_out;
    if (max_back) *max_back = far;
    return status;
}


// (r06) two wavefronts: the second one counts the next sweep while the first stores and resolves the current one (ZaSweepHelp)
#define ZA_SERIAL_THREADS 128
__global__ __launch_bounds__(ZA_SERIAL_THREADS) void za_k_inflate_serial(const uint8_t *__restrict__ in, uint64_t in_len, uint32_t start_bit,
                                                          const uint8_t *__restrict__ dict, uint32_t dict_len,
                                                          uint8_t *__restrict__ out, uint64_t out_cap,
                                                          ZaInfResult *__restrict__ res)
{
    __shared__ ZaInfTabs T;
    __shared__ uint8_t win[ZA_WIN];
    __shared__ int scratch[2];
    __shared__ ZaParBufT<1024, 3072> P;
    __shared__ ZaSweepHelp H;
    if (threadIdx.x == 0) { H.valid = 0u; H.go = 0u; H.done = 0u; H.cmd = 0u; }
    __syncthreads();                                               // (the one barrier both waves meet)
    if (threadIdx.x >= 64) {
        za_sweep_helper_loop<uint8_t, ZaParBufT<1024, 3072>, ZaInfTabs, false>(in, in_len, dict, dict_len, out, out_cap, T, &P, &H);
        return;
    }
    uint64_t bits = 0, op = 0, cpb = 0, cpo = 0;
    const int status = za_inflate_serial_core<0, uint8_t, ZA_WIN, ZaParBufT<1024, 3072>, ZaInfTabs, false, true>(in, in_len, dict, dict_len, out, out_cap, T, win, scratch, P.stage, bits, op, start_bit, &cpb, &cpo,
                                                         0xFFFFFFFFu, false, nullptr, nullptr, 0, 0, &P, &H);
    za_sweep_helper_exit(&H);
    if (za_lane() == 0) { res->status = status; res->pad = 0; res->out_len = op; res->in_bits = bits; res->block_bits = cpb; res->block_out = cpo; }
}

// The same decoder for a SMALL stream (r06): the output is assembled in LDS and leaves in one coalesced pass at the end.  On a
// lone wavefront every round of the sweeps' match resolution was a load and a store round trip to memory behind a fence
// (1.15 us a round, 400 of a 64 KiB stream's 1 100 us), every literal of the storing pass and of the sequential rounds a store
// to memory, the ring refill behind a sweep a read-back of it: with `out` in LDS they are LDS round trips.  The image holds
// ZA_SMALL_IMG bytes; a stream that produces more comes back as "output full" with exactly that many, and the host runs the
// ordinary kernel (zngamd_inflate_raw's small path).
#ifndef ZA_SMALL_IMG
#define ZA_SMALL_IMG 86016        // (84 KiB: with the helper's staging area the kernel takes 155 of the CU's 160 KiB)
#endif
__global__ __launch_bounds__(128) void za_k_inflate_serial_small(const uint8_t *__restrict__ in, uint64_t in_len, uint32_t start_bit,
                                                                 const uint8_t *__restrict__ dict, uint32_t dict_len,
                                                                 uint8_t *__restrict__ out, uint64_t out_cap,
                                                                 ZaInfResult *__restrict__ res)
{
    __shared__ ZaInfTabs T;
    __shared__ uint8_t win[ZA_WIN];
    __shared__ int scratch[2];
    __shared__ ZaParBufT<1024, 3072> P;
    __shared__ ZaSweepHelp H;
    __shared__ __attribute__((aligned(16))) uint8_t img[ZA_SMALL_IMG];
    const uint64_t cap = out_cap < (uint64_t)ZA_SMALL_IMG ? out_cap : (uint64_t)ZA_SMALL_IMG;
    const int lane = za_lane();
    if (threadIdx.x < 64 && lane == 0) { H.valid = 0u; H.go = 0u; H.done = 0u; H.cmd = 0u; }
    __syncthreads();                                               // (the one barrier both waves meet)
    if (threadIdx.x >= 64) {
        za_sweep_helper_loop<uint8_t, ZaParBufT<1024, 3072>, ZaInfTabs, true>(in, in_len, dict, dict_len, img, cap, T, &P, &H);
        return;
    }
    uint64_t bits = 0, op = 0, cpb = 0, cpo = 0;
    const int status = za_inflate_serial_core<0, uint8_t, ZA_WIN, ZaParBufT<1024, 3072>, ZaInfTabs, true, true>(in, in_len, dict, dict_len, img, cap, T, win, scratch, P.stage, bits, op, start_bit, &cpb, &cpo,
                                                         0xFFFFFFFFu, false, nullptr, nullptr, 0, 0, &P, &H);
    za_sweep_helper_exit(&H);
    za_wave_sync();
    for (uint64_t i = 16ull * (uint64_t)lane; i < op; i += 1024ull) {
        if (i + 16ull <= op) { const uint4 v = *(const uint4 *)(img + i); ZaU4u t = {v.x, v.y, v.z, v.w}; *(ZaU4u *)(out + i) = t; }
        else for (uint64_t k = i; k < op; k++) out[k] = img[k];
    }
    if (lane == 0) { res->status = status; res->pad = (uint32_t)ZA_SMALL_IMG; res->out_len = op; res->in_bits = bits; res->block_bits = cpb; res->block_out = cpo; }
}

// ------------------------------------------------------------------------------------------------
// indexed gzip members
// ------------------------------------------------------------------------------------------------
// Member layout written by za_k_assemble_members (all little endian):
//   0  1f 8b 08 04 | 4 mtime=0 | 8 xfl | 9 os=ff | 10 XLEN(u16) | 12 'Z' 'A' | 14 SLEN(u16)
//   16 member_size(u32) | 20 isize(u32) | 24 nchunk(u16) = ceil(isize / 256) | 26 version(u8) = 2 | 27 chunk shift(u8) = 8
//   28 (nchunk + 1) x u32 index entries: bits 0..22 = bit offset (from the first deflate byte) of the first token that
//      starts at or behind output byte 256 c, bits 23..31 = how many bytes behind (0..257: a match may run across the
//      boundary -- token boundaries are NOT forced here); entry nchunk = bit offset of the end-of-block code
//   32 + 4 nchunk: deflate bytes (one final block; a dynamic header in its flat form) ... | crc32(u32) | isize(u32)
#define ZA_MEMBER_FIXED 32                                     // header bytes besides the 4 per chunk
#define ZA_MEMBER_HDR(nchunk) (ZA_MEMBER_FIXED + 4u * (nchunk))

struct ZaMember {          // mirrors zngamd_member
    uint64_t in_off, in_len, out_off;
    uint32_t out_len, crc, index_off, nseg;
};

struct ZaCand { uint64_t off; uint32_t size; uint32_t isize; };

__global__ __launch_bounds__(256) void za_k_scan_members(const uint8_t *__restrict__ in, uint64_t in_len,
                                                         ZaCand *__restrict__ cands, uint32_t max_cands,
                                                         uint32_t *__restrict__ n_cands)
{
    // every thread owns 16 consecutive byte positions; loads are coalesced 16 B per lane
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = t * 16ull;
    if (base >= in_len) return;
    uint32_t w[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint64_t o = base + 4ull * k;
        w[k] = (o + 4 <= in_len) ? *(const uint32_t *)(in + o) : 0u;     // in is 16-byte aligned (hipMalloc)
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int wi = k >> 2, sh = (k & 3) * 8;
        const uint32_t v = sh ? ((w[wi] >> sh) | (w[wi + 1] << (32 - sh))) : w[wi];
        if (v != 0x04088b1fu) continue;
        const uint64_t o = base + (uint64_t)k;
        if (o + ZA_MEMBER_FIXED + 8 > in_len) continue;
        const uint8_t *h = in + o;
        if (h[9] != 0xFF || h[12] != 'Z' || h[13] != 'A' || h[26] != 2 || h[27] != ZA_CHUNK_SHIFT) continue;
        const uint32_t size = za_ld32(h + 16), isize = za_ld32(h + 20), nchunk = za_ld16(h + 24), xlen = za_ld16(h + 10), slen = za_ld16(h + 14);
        const uint32_t hdr = ZA_MEMBER_HDR(nchunk);
        if (isize > ZA_MAX_UNIT || nchunk != ((isize + (1u << ZA_CHUNK_SHIFT) - 1u) >> ZA_CHUNK_SHIFT) || xlen != hdr - 12u || slen != hdr - 16u) continue;
        if (size < hdr + 8u || o + size > in_len) continue;
        const uint32_t idx = atomicAdd(n_cands, 1u);
        if (idx < max_cands) { ZaCand c; c.off = o; c.size = size; c.isize = isize; cands[idx] = c; }
    }
}

#define ZA_MATCHQ_PER_SEG 688      // queue entries per segment: >= 2048/3 matches (one entry per round at most), a multiple of 4
#ifndef ZA_IROW_LOADS
#define ZA_IROW_LOADS 4            // 16-byte loads per row of staged input
#endif
#define ZA_IROW (4 * ZA_IROW_LOADS + 1)            // dwords per lane row (+ 1: odd stride)
#define ZA_IROW_BYTES (16 * ZA_IROW_LOADS - 16)    // bytes consumed per row; the last 16 are look-ahead (a round of the member decoder takes up to 97 bits)
#ifndef ZA_ILITS
#define ZA_ILITS 3                 // literals a lane takes per round at most
#endif
#define ZA_ML_BITS 10              // literal/length table: every code of an indexed member is at most 10 bits long (ZA_LIMIT_L)
#define ZA_MD_BITS 9               // distance table (ZA_LIMIT_D)

// Tables of the member decoder.  One table read decodes any symbol (the encoder limits the code lengths, and a member with
// longer codes is left to the sequential decoder); the entries carry what the token needs:
//   lut_l  bit 15 = 0: literal, bits 4..11 the byte;  bit 15 = 1: length, bits 4..11 = base - 3, bits 12..14 = extra bits
//          (7 = end of block);  bits 0..3 = code length, 0 = invalid code
//   lut_d  bits 8..23 = base, bits 4..7 = extra bits, bits 0..3 = code length, 0 = invalid code
struct ZaMemTabs {
    uint16_t lut_l[1 << ZA_ML_BITS];
    uint32_t lut_d[1 << ZA_MD_BITS];
};
struct ZaMemBuild {                 // only while the tables are built: lives in the row area
    uint16_t tmp_d[1 << ZA_MD_BITS];
    uint16_t cnt_l[16], cnt_d[16];
    uint16_t sym_l[288], sym_d[32];
    uint8_t lens[320];
};

__global__ __launch_bounds__(64, 5) void za_k_inflate_members(const uint8_t *__restrict__ in, uint64_t in_total,
                                                              const ZaMember *__restrict__ members,
                                                              uint8_t *__restrict__ out, uint64_t out_cap,
                                                              uint32_t *__restrict__ matchq,       // [grid][64][ZA_MATCHQ_PER_SEG]
                                                              const uint32_t *__restrict__ crc_tabs,     // [4][256] slice-by-4 tables | [8][16] advance over 2 016 zero bytes | [128] x^(8 * 32 k)
                                                              const uint32_t *__restrict__ x8k_table,
                                                              int32_t *__restrict__ status_out)
{
    __shared__ ZaMemTabs T;
    __shared__ int scratch[2];
    __shared__ __attribute__((aligned(16))) uint32_t rows[64 * ZA_IROW];      // table build: ZaMemBuild; phase A: staged input; then the CRC table
    static_assert(sizeof(ZaMemBuild) <= sizeof(uint32_t) * 64 * ZA_IROW, "build area");
    ZaMemBuild &B = *(ZaMemBuild *)rows;
    const int lane = za_lane();
    const ZaMember m = members[blockIdx.x];
    const uint8_t *src = in + m.in_off;
    const uint64_t in_bits = m.in_len * 8ull;
    uint8_t *dst = out + m.out_off;
    const int n = (int)m.out_len;
    const int nseg = (int)m.nseg;
    if (m.in_off + m.in_len + 8 > in_total || m.out_off + m.out_len > out_cap || n > ZA_MAX_UNIT || m.in_len > (1u << 20) ||
        nseg != ((n + ZA_SEG - 1) >> ZA_SEG_SHIFT) || m.index_off != 4u * ((uint32_t)nseg + 1u) || m.index_off > m.in_off || n == 0) {
        if (lane == 0) status_out[blockIdx.x] = ZA_I_INDEX;
        return;
    }
    const uint32_t *index = (const uint32_t *)(src - m.index_off);      // member starts are byte aligned only: unaligned loads
    // index entries: bit offset | overshoot << 23; at this granularity (one entry per 2 KiB segment, where the codec forces a
    // token boundary) the overshoot is zero
    const uint32_t my_start = za_ld32((const uint8_t *)(index + (lane < nseg ? lane : nseg)));
    const uint32_t my_stop = za_ld32((const uint8_t *)(index + (lane < nseg ? lane + 1 : nseg)));
    if (__ballot((my_start >> 23) != 0u || (my_stop >> 23) != 0u) != 0ull || in_bits < 3) { if (lane == 0) status_out[blockIdx.x] = ZA_I_INDEX; return; }
    // ---- block header (uniform).  Members written by this engine are one final block, fixed or dynamic with the header in its
    // flat form: HCLEN = 19, the code-length code is the fixed 4-bit code of the symbols 0..15, so code length k sits in the
    // 4 bits at 74 + 4 k (bit-reversed) and all lanes read the header at once.  Anything else: sequential decoder.
    const uint64_t bits = za_peek(src, 0);
    const int last = (int)(bits & 1u), type = (int)((bits >> 1) & 3u);
    uint32_t nlen = 288, ndist = 30, hdr_end = 3;
    bool hdr_ok = last && (type == 1 || type == 2);
    if (hdr_ok && type == 2) {
        nlen = (uint32_t)((bits >> 3) & 31u) + 257u; ndist = (uint32_t)((bits >> 8) & 31u) + 1u;
        uint64_t want = 0;
        for (int i = 3; i < 19; i++) want |= 4ull << (3 * i);
        hdr_end = 74u + 4u * (nlen + ndist);
        hdr_ok = in_bits >= 74 && ((bits >> 13) & 15u) == 15u && (za_peek(src, 17) & ((1ull << 57) - 1ull)) == want && nlen <= 286 && ndist <= 30 &&
                 hdr_end <= in_bits;
    }
    if (!hdr_ok || __shfl(my_start, 0, 64) != hdr_end) { if (lane == 0) status_out[blockIdx.x] = ZA_I_INDEX; return; }
    {
        bool toolong = false;
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
        if (__ballot(toolong) != 0ull) { if (lane == 0) status_out[blockIdx.x] = ZA_I_INDEX; return; }
        __syncthreads();
        // plain tables ((symbol << 4) | length) first -- the distance one in the row area -- then the entries are rewritten
        uint16_t *tmp_d = B.tmp_d;
        int ok = B.lens[256] != 0;
        int st = za_build_table(B.lens, (int)nlen, B.cnt_l, B.sym_l, T.lut_l, ZA_ML_BITS, &scratch[0], &scratch[1]);
        if (st < 0 || (st > 0 && scratch[1] != 1)) ok = 0;
        st = za_build_table(B.lens + 288, (int)(type == 1 ? 32u : ndist), B.cnt_d, B.sym_d, tmp_d, ZA_MD_BITS, &scratch[0], &scratch[1]);
        if (st < 0 || (st > 0 && scratch[1] != 1 && type != 1)) ok = 0;       // (the fixed block's 30 five-bit distance codes are incomplete by design)
        if (!ok) { if (lane == 0) status_out[blockIdx.x] = ZA_I_INDEX; return; }
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
        uint8_t *litp = dst + seg0;
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
                    const bool bad_data = em == 0u || ((em & 0x7000u) != 0x7000u && (d == 0u || (int)dist > pos1));
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
            if (seg0 + (int)b + 16 <= n) { ZaU4u v; v.x = (uint32_t)blk_lo; v.y = (uint32_t)(blk_lo >> 32); v.z = (uint32_t)blk_hi; v.w = (uint32_t)(blk_hi >> 32); *(ZaU4u *)(litp + b) = v; }
            else for (uint32_t k = b; k < nlit; k++) { const uint32_t o = k - b; litp[k] = (uint8_t)(o < 8 ? blk_lo >> (8 * o) : blk_hi >> (8 * (o - 8))); }
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
            if ((e & 0xF000u) != 0xF000u) lane_err = 1;
            else if (((bp + (e & 15u) + 7u) >> 3) != (uint32_t)m.in_len) lane_err = 1;
        }
    }
    const unsigned long long e1 = __ballot(lane_err == 1), e2 = __ballot(lane_err == 2);
    if (e1 || e2) { if (lane == 0) status_out[blockIdx.x] = e2 ? ZA_I_DATA : ZA_I_INDEX; return; }
    __threadfence_block();       // the literal bytes and the match queues are visible to the whole wave

    // ---- phase B: expand, segment by segment in output order, inside an LDS image of the segment.
    // The image (in the row area) is the segment's 2 KiB behind the last 272 bytes of the output in front of it.  The segment's
    // literal bytes are loaded RIGHT-ALIGNED into it: literal k of nlit sits at seg_len - nlit + k, at or behind its final place,
    // and the expansion moves every run of literals down to where it belongs while the matches fill the gaps -- in ascending
    // order, so a run never lands on bytes that are still to be moved (all lanes of a group read before any of them writes).
    // 64 queue entries at a time: destinations from a wave prefix sum, the literal runs move (two 16-byte copies each at most),
    // then the matches: LDS to LDS when the source starts inside the image, with one or two 16-byte global loads when it lies
    // further back (final bytes in memory).  Inside a group a match waits for the lanes that write what it reads (exact masks
    // from two shuffle binary searches).  CRC-32 is taken from the finished image as it leaves (32 bytes per lane, slice-by-4,
    // the state carried over the 2 016 bytes of the other lanes by eight table reads), so the output is never read back.
    uint32_t crc_r = lane == 0 ? 0xFFFFFFFFu : 0u;              // raw CRC state of "my bytes, zeros elsewhere" (lane 0 carries the preset)
#ifndef ZA_ABL_NO_B
    {
        uint8_t *img = (uint8_t *)rows;                            // [0, 272): output in front of the segment, [272, 272 + 2048): the segment
        const uint32_t TAIL = 272u;
        uint32_t *crct = (uint32_t *)&T;                           // slice-by-4 tables take the place of the decode tables
        uint32_t *advt = rows + (TAIL + ZA_SEG + 32) / 4;          // [8][16]: state advanced over 2 016 zero bytes, by nibble
        static_assert(sizeof(uint32_t) * 64 * ZA_IROW >= 272 + ZA_SEG + 32 + 512, "segment image + advance table");
        static_assert(sizeof(ZaMemTabs) >= 4096, "slice tables");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < 1024; i += 64) crct[i] = crc_tabs[i];
        for (int i = lane; i < 128; i += 64) advt[i] = crc_tabs[1024 + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int s = 0; s < nseg; s++) {
            const uint32_t cnt = __shfl(nmatch, s, 64), lits = __shfl(nlit, s, 64);
            const uint32_t *q = matchq + ((size_t)blockIdx.x * 64 + (size_t)s) * ZA_MATCHQ_PER_SEG;
            const uint32_t seg_start = (uint32_t)s << ZA_SEG_SHIFT;
            const uint32_t seg_len = (uint32_t)n - seg_start < (uint32_t)ZA_SEG ? (uint32_t)n - seg_start : (uint32_t)ZA_SEG;
            uint32_t ent_next = (uint32_t)lane < cnt ? q[lane] : 0u;
            // image: the tail of the previous segment moves to the front (it is final); the literal bytes come from memory,
            // piece by piece of 16 bytes where a piece holds any (image byte x is literal byte x - shift)
            __builtin_amdgcn_wave_barrier();
            uint32_t t0 = 0, t1 = 0;
            if (s > 0) { t0 = ((const uint32_t *)(img + ZA_SEG))[lane]; if (lane < 4) t1 = ((const uint32_t *)(img + ZA_SEG))[64 + lane]; }
            {
                const int shift = (int)(seg_len - lits);             // match bytes of the segment
                ZaU4u pc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int x = lane * 32 + 16 * j, o = x - shift;      // image offset of the piece, and where it starts in the literal bytes
                    if (x < (int)seg_len && o + 16 > 0) {
                        // (bytes in front of the first literal or behind the last are of no meaning; they must only be readable)
                        if ((o >= 0 || m.out_off + seg_start >= 16u) && m.out_off + seg_start + (uint64_t)(o + 16) <= out_cap) pc[j] = *(const ZaU4u *)(dst + seg_start + o);
                        else {
                            uint32_t t[4] = {0, 0, 0, 0};
                            for (int k = 0; k < 16; k++) if (o + k >= 0 && o + k < (int)lits) t[k >> 2] |= (uint32_t)dst[seg_start + o + k] << (8 * (k & 3));
                            pc[j].x = t[0]; pc[j].y = t[1]; pc[j].z = t[2]; pc[j].w = t[3];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (s > 0) { ((uint32_t *)img)[lane] = t0; if (lane < 4) ((uint32_t *)img)[64 + lane] = t1; }
                *(uint4 *)(img + TAIL + lane * 32) = make_uint4(pc[0].x, pc[0].y, pc[0].z, pc[0].w);
                *(uint4 *)(img + TAIL + lane * 32 + 16) = make_uint4(pc[1].x, pc[1].y, pc[1].z, pc[1].w);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t segpos = seg_start;                               // output position behind the entries handled so far
            uint32_t mrem = seg_len - lits;                            // match bytes at or behind segpos: how far the literals there sit above their place
            for (uint32_t g = 0; g < cnt; g += 64) {
                const bool hasq = g + (uint32_t)lane < cnt;
                const uint32_t ent = ent_next;
                ent_next = g + 64u + (uint32_t)lane < cnt ? q[g + 64u + lane] : 0u;      // the next group's entries travel while this group is resolved
                const uint32_t l2 = (ent >> 15) & 0x1FFu;
                const bool has = hasq && l2 != 0u;                      // a real match (length field 0: a run of literals)
                const uint32_t mlen = has ? l2 + 2u : 0u, mdist = (ent & 0x7FFFu) + 1u;
                const uint32_t glit = !hasq ? 0u : has ? (ent >> 24) : ent;       // literals in front of the match (<= 32)
                const uint32_t incl = za_wave_incl_scan(glit + mlen), incm = za_wave_incl_scan(mlen);
                const uint32_t mdst = segpos + incl - mlen;
                const uint32_t up = mrem - (incm - mlen);               // my literals sit `up` bytes above their place
                segpos += (uint32_t)__shfl((int)incl, 63, 64);
                mrem -= (uint32_t)__shfl((int)incm, 63, 64);
                uint8_t *od = img + TAIL + (mdst - seg_start);          // my match's destination inside the image; my literals end there
                // the runs of literals move down: every lane reads its run, then every lane writes it
                if (__ballot(glit != 0u && up != 0u) != 0ull) {
                    const uint8_t *sp = od - glit + up;
                    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
                    const bool mv = glit != 0u && up != 0u;
                    if (mv) {
                        a0 = *(const za_u32u *)sp; a1 = *(const za_u32u *)(sp + 4); a2 = *(const za_u32u *)(sp + 8); a3 = *(const za_u32u *)(sp + 12);
                        if (glit > 16u) { b0 = *(const za_u32u *)(sp + 16); b1 = *(const za_u32u *)(sp + 20); b2 = *(const za_u32u *)(sp + 24); b3 = *(const za_u32u *)(sp + 28); }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (mv) {
                        uint8_t *o = od - glit;
                        uint32_t rem = glit;
                        if (glit > 16u) {
                            *(za_u32u *)o = a0; *(za_u32u *)(o + 4) = a1; *(za_u32u *)(o + 8) = a2; *(za_u32u *)(o + 12) = a3;
                            o += 16; a0 = b0; a1 = b1; a2 = b2; a3 = b3; rem = glit - 16u;
                        }
                        if (rem & 16u) { *(za_u32u *)o = a0; *(za_u32u *)(o + 4) = a1; *(za_u32u *)(o + 8) = a2; *(za_u32u *)(o + 12) = a3; }
                        else {
                            if (rem & 8u) { *(za_u32u *)o = a0; *(za_u32u *)(o + 4) = a1; o += 8; a0 = a2; a1 = a3; }
                            if (rem & 4u) { *(za_u32u *)o = a0; o += 4; a0 = a1; }
                            if (rem & 2u) { *(za_u16u *)o = (uint16_t)a0; o += 2; a0 >>= 16; }
                            if (rem & 1u) *o = (uint8_t)a0;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                bool done = !has;
                unsigned long long pending = __ballot(!done);
                // Which matches of this group write bytes that mine reads?  Destinations are disjoint and ascending with
                // the lane, so they are the lanes [jlo, jhi): jhi = matches that start below the end of my source,
                // jlo = matches that end at or below its start (two 6-step binary searches with shuffles; both counts
                // are at most my own lane).  A match is ready as soon as none of those is pending -- the lowest
                // pending one always is.  (Lanes without a match sit at their position with length 0: they order correctly.)
                const uint32_t sdst = hasq ? mdst : 0xFFFFFFFFu, send = hasq ? mdst + mlen : 0xFFFFFFFFu;
                const uint32_t src_a = mdst - mdist, src_b = src_a + (mlen < mdist ? mlen : mdist);
                uint32_t jhi = 0, jlo = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                    const uint32_t vd = (uint32_t)__shfl((int)sdst, (int)(jhi + step - 1u), 64);
                    const uint32_t ve = (uint32_t)__shfl((int)send, (int)(jlo + step - 1u), 64);
                    if (vd < src_b) jhi += step;
                    if (ve <= src_a) jlo += step;
                }
                const unsigned long long deps = ((1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull);
                // far: the source ends below the segment (it starts more than 272 bytes in front of it): final bytes in memory
#ifdef ZA_ABL_NO_FAR
                const bool far = false;                                  // (timing only: far sources read garbage inside the image)
#else
                const bool far = src_a + TAIL < seg_start;
#endif
                const bool simple = mdist >= mlen && mlen <= 32u;      // copied by its own lane, at most two 16-byte batches
                // the far sources of the whole group are fetched at once (nothing in this group can change them)
                ZaU4u fv = {0, 0, 0, 0}, fv2 = {0, 0, 0, 0};
                if (has && far && simple) {
                    fv = *(const ZaU4u *)(dst + src_a);
                    if (mlen > 16u) fv2 = *(const ZaU4u *)(dst + src_a + 16);
                }
                while (pending) {
                    const bool ready = !done && (pending & deps) == 0ull;
                    if (ready && simple) {
                        ZaU4u v = fv, v2 = fv2;
                        if (!far) {
                            const uint8_t *sp = od - mdist;                 // inside the image: src_a >= seg_start - 272
#ifdef ZA_ABL_NO_FAR
                            if (sp < img) sp = img;
#endif
                            v.x = *(const za_u32u *)sp; v.y = *(const za_u32u *)(sp + 4); v.z = *(const za_u32u *)(sp + 8); v.w = *(const za_u32u *)(sp + 12);
                            if (mlen > 16u) { v2.x = *(const za_u32u *)(sp + 16); v2.y = *(const za_u32u *)(sp + 20); v2.z = *(const za_u32u *)(sp + 24); v2.w = *(const za_u32u *)(sp + 28); }
                        }
                        uint8_t *o = od;
                        uint32_t rem = mlen;
                        if (mlen > 16u) {
                            *(za_u32u *)o = v.x; *(za_u32u *)(o + 4) = v.y; *(za_u32u *)(o + 8) = v.z; *(za_u32u *)(o + 12) = v.w;
                            o += 16; v = v2; rem = mlen - 16u;
                        }
                        if (rem & 16u) { *(za_u32u *)o = v.x; *(za_u32u *)(o + 4) = v.y; *(za_u32u *)(o + 8) = v.z; *(za_u32u *)(o + 12) = v.w; }
                        else {
                            if (rem & 8u) { *(za_u32u *)o = v.x; *(za_u32u *)(o + 4) = v.y; o += 8; v.x = v.z; v.y = v.w; }
                            if (rem & 4u) { *(za_u32u *)o = v.x; o += 4; v.x = v.y; }
                            if (rem & 2u) { *(za_u16u *)o = (uint16_t)v.x; o += 2; v.x >>= 16; }
                            if (rem & 1u) *o = (uint8_t)v.x;
                        }
                    }
                    // long or self-overlapping matches: the whole wave copies them, one at a time
                    unsigned long long coop = __ballot(ready && !simple);
                    while (coop) {
                        const int j = __builtin_ctzll(coop);
                        coop &= coop - 1ull;
                        const uint32_t cd = (uint32_t)__builtin_amdgcn_readlane((int)mdst, j);
                        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)mlen, j);
                        const uint32_t cdist = (uint32_t)__builtin_amdgcn_readlane((int)mdist, j);
#ifdef ZA_ABL_NO_FAR
                        const bool cfar = false;
                        if (cd - cdist + TAIL < seg_start) continue;
#else
                        const bool cfar = cd - cdist + TAIL < seg_start;      // (then cdist > cl: no overlap)
#endif
                        uint8_t *o = img + TAIL + (cd - seg_start);
                        const float rd = 1.0f / (float)cdist;
                        for (uint32_t base = 0; base < cl; base += 64) {
                            const uint32_t i = base + (uint32_t)lane;
                            if (i < cl) {
                                // byte i of a self-overlapping match is byte (i mod dist) of its period, which lies below it
                                int k = (int)i;
                                if (cdist < cl) {
                                    k = (int)i - (int)cdist * (int)((float)i * rd);
                                    if (k < 0) k += (int)cdist;
                                    if (k >= (int)cdist) k -= (int)cdist;
                                }
                                o[i] = cfar ? dst[cd - cdist + (uint32_t)k] : (o - cdist)[k];
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the image is LDS and one wave's LDS accesses execute in order: only the compiler must keep the order (no wait, no trip to L2)
                    __builtin_amdgcn_wave_barrier();
                    done = done || ready;
                    pending = __ballot(!done);
                }
            }
            // the finished segment: 32 bytes per lane, coalesced (its last, partial piece bytewise), CRC-32 on the way out
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t o = (uint32_t)lane * 32u;
#ifndef ZA_ABL_NO_CRC
                if (s > 0 && o < seg_len) {                              // my state crosses the 2 016 bytes of the other lanes
                    uint32_t a = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) a ^= advt[16 * k + ((crc_r >> (4 * k)) & 15u)];
                    crc_r = a;
                }
#endif
                if (o + 32u <= seg_len) {
                    const uint4 a = *(const uint4 *)(img + TAIL + o), b2 = *(const uint4 *)(img + TAIL + o + 16);
                    ZaU4u va = {a.x, a.y, a.z, a.w}, vb = {b2.x, b2.y, b2.z, b2.w};
                    *(ZaU4u *)(dst + seg_start + o) = va; *(ZaU4u *)(dst + seg_start + o + 16) = vb;
#ifndef ZA_ABL_NO_CRC
                    const uint32_t wv[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        crc_r ^= wv[k];
                        crc_r = crct[768 + (crc_r & 0xFFu)] ^ crct[512 + ((crc_r >> 8) & 0xFFu)] ^ crct[256 + ((crc_r >> 16) & 0xFFu)] ^ crct[crc_r >> 24];
                    }
#endif
                } else for (uint32_t k = 0; k < 32u && o + k < seg_len; k++) {
                    const uint8_t bv = img[TAIL + o + k];
                    dst[seg_start + o + k] = bv;
#ifndef ZA_ABL_NO_CRC
                    crc_r = crct[(crc_r ^ bv) & 0xFFu] ^ (crc_r >> 8);
#endif
                }
            }
            __threadfence_block();       // later segments read these bytes from memory
        }
    }
#endif

    // ---- verify against the member trailer (CRC32, ISIZE), zlib_ngmodule.c:2577-2599
#ifdef ZA_ABL_NO_CRC
    const uint32_t c = za_ld32(src + m.in_len);
#else
    uint32_t c = 0;
    {
        // my state stands behind my last byte: the bytes between there and the end of the member are zeros to it -- a few
        // table steps for the odd bytes, one GF(2) product with x^(8 * 32 k) for the rest (k < 128: two segments at most)
        const uint32_t first = (uint32_t)lane * 32u;
        if (first < (uint32_t)n) {
            const uint32_t s_last = ((uint32_t)n - 1u - first) >> ZA_SEG_SHIFT;
            uint32_t endp = (s_last << ZA_SEG_SHIFT) + first + 32u;
            if (endp > (uint32_t)n) endp = (uint32_t)n;
            const uint32_t after = (uint32_t)n - endp;
            const uint32_t *crct = (const uint32_t *)&T;
            uint32_t r = crc_r;
            for (uint32_t k = 0; k < (after & 31u); k++) r = crct[r & 0xFFu] ^ (r >> 8);
            c = (after >> 5) ? za_multmodp(crc_tabs[1152 + (after >> 5)], r) : r;
        }
        c = za_wave_xor_reduce(c) ^ 0xFFFFFFFFu;
    }
#endif
    const uint32_t want_crc = za_ld32(src + m.in_len), want_len = za_ld32(src + m.in_len + 4);
    if (lane == 0) status_out[blockIdx.x] = (c != want_crc) ? ZA_I_CRC : (want_len != (uint32_t)n) ? ZA_I_LENGTH : ZA_I_OK;
}

// Members whose extent is known up front without this engine's index (BGZF: 'B','C' subfield with the
// block size; reference fixture tests/data/test.fastq.bgzip.gz): one wavefront per member runs the
// sequential decoder, then checks CRC-32 and ISIZE against the trailer (zlib_ngmodule.c:2577-2599).
// history ring and match queue of the one-wavefront-per-member decoder (16 KiB of LDS per wavefront, 9 per CU).  A queue of 512
// entries (11 per CU) is 6 % faster on 8 192 zlib-written members of 128 KiB and 10 % slower on 4 097 BGZF members of 64 KiB
// (fewer wavefronts than the GPU holds: occupancy is no help, shorter sweeps hurt); 256 entries cost 30 %; kept at 1 024
// (`profiles/abl_foreign.sh`, `profiles/abl_bgzf.sh`)
// (r03, after a short sweep stopped costing 32 sequential rounds: a queue of 768 and 512 bytes of history -- the sequential
// rounds that use the ring are 4 per member now -- leave room for 12 wavefronts per CU instead of 9: 10.96 -> 9.53 ms per GiB;
// queue 1 024 with the small ring 10.6, queue 512 13.8, 896 entries with 416-bit sub-sequences 9.48)
#ifndef ZA_MEMBER_LBITS
#define ZA_MEMBER_LBITS 9               // index bits of the first-level tables (see ZaInfTabsT): 9 / 8 -> 7.89 ms per GiB, 9 / 7 8.14, 8 / 8 8.02, 8 / 7 8.16, 10 / 8 9.29, 10 / 9 9.42
#endif
#ifndef ZA_MEMBER_DBITS
#define ZA_MEMBER_DBITS 8
#endif
#ifndef ZA_MEMBER_RING
#define ZA_MEMBER_RING 512
#endif
#ifndef ZA_MEMBER_Q
#define ZA_MEMBER_Q 768
#endif
#ifndef ZA_MEMBER_BITS
#define ZA_MEMBER_BITS 384             // bits per sub-sequence: 384 is 5 % faster than 256 on zlib-written 128 KiB members (fewer passes per
#endif                                 // output byte, still 9 wavefronts per CU), equal on 64 KiB BGZF members; 512 needs a longer queue and loses
__global__ __launch_bounds__(64) void za_k_inflate_serial_members(const uint8_t *__restrict__ in, uint64_t in_total,
                                                                  const ZaMember *__restrict__ members,
                                                                  uint8_t *__restrict__ out, uint64_t out_cap,
                                                                  const uint32_t *__restrict__ crc_table,
                                                                  const uint32_t *__restrict__ x8k_table,
                                                                  int32_t *__restrict__ status_out)
{
    __shared__ ZaInfTabsT<ZA_MEMBER_LBITS, ZA_MEMBER_DBITS> T;
    __shared__ uint8_t win[ZA_MEMBER_RING];       // the last bytes of history in LDS; older sources come from the output
    __shared__ int scratch[2];
    __shared__ ZaParBufT<ZA_MEMBER_BITS, ZA_MEMBER_Q> P;
    static_assert(sizeof(P.stage) >= 256 * sizeof(uint32_t), "the CRC table takes the place of the staged stream once the member is decoded");
    uint32_t *crct = P.stage;                      // (loaded behind the decode: a KiB less of LDS per wavefront, 13 per CU instead of 12)
    const int lane = za_lane();
    const ZaMember m = members[blockIdx.x];
    if (m.in_off + m.in_len + 8 > in_total || m.out_off + m.out_len > out_cap) {
        if (lane == 0) status_out[blockIdx.x] = ZA_I_DATA;
        return;
    }
    const uint8_t *src = in + m.in_off;
    uint8_t *dst = out + m.out_off;
    uint64_t bits = 0, op = 0;
    int status = za_inflate_serial_core<0, uint8_t, ZA_MEMBER_RING, ZaParBufT<ZA_MEMBER_BITS, ZA_MEMBER_Q>>(src, m.in_len, nullptr, 0, dst, m.out_len, T, win, scratch, P.stage, bits, op,
                                                                       0, nullptr, nullptr, 0xFFFFFFFFu, false, nullptr, nullptr, 0, 0, &P);
    if (status == ZA_I_END) {
        status = ZA_I_OK;
        if (((bits + 7) >> 3) != m.in_len) status = ZA_I_DATA;          // the member must end where its size says
        else if (op != m.out_len) status = ZA_I_LENGTH;
        else {
            __threadfence_block();
            __syncthreads();
            for (int i = lane; i < 256; i += 64) crct[i] = crc_table[i];
            __syncthreads();
            uint32_t crc = 0;
            for (uint64_t o = 0; o < op; o += ZA_MAX_UNIT) {
                const int len = (int)((op - o) > ZA_MAX_UNIT ? ZA_MAX_UNIT : (op - o));
                const uint32_t c = za_wave_crc32(dst + o, len, crct, x8k_table);
                uint32_t xp = 0x80000000u, sq = 0x00800000u;            // crc = crc * x^(8 len) ^ c
                for (int k = len; k; k >>= 1) { if (k & 1) xp = za_multmodp(sq, xp); sq = za_multmodp(sq, sq); }
                crc = za_multmodp(xp, crc) ^ c;
            }
            const uint32_t want_crc = za_ld32(src + m.in_len), want_len = za_ld32(src + m.in_len + 4);
            if (crc != want_crc) status = ZA_I_CRC;
            else if (want_len != (uint32_t)op) status = ZA_I_LENGTH;
        }
    } else if (status == ZA_I_OUTFULL) status = ZA_I_LENGTH;
    if (lane == 0) status_out[blockIdx.x] = status;
}

// ------------------------------------------------------------------------------------------------
// parallel inflate of ONE deflate stream that contains sync-flush points (what block-parallel writers emit:
// the reference's gzip_ng_threaded, pigz, this engine's own writer).  SURVEY.md section 8f-3.
//   za_k_scan_sync        byte positions that follow `00 00 FF FF` = candidate chunk starts
//   za_k_chunk_count      one wave per candidate: decode without storing anything until the next in-stream sync
//                         point / the final block; gives the chunk's compressed and uncompressed size and
//                         weeds out false candidates (they fail to decode or are never reached by the chain)
//   za_k_chunk_decode     one wave per chunk of the chain: decode with 16-bit symbols; what a chunk copies from
//                         the (still unknown) 32 KiB before its start stays a marker naming that byte
//   za_k_chunk_compose / _chain   the 32 KiB window before every chunk, as a blocked scan over the chunk chain
//   za_k_chunk_resolve    one workgroup per chunk: markers -> bytes
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void za_k_scan_sync(const uint8_t *__restrict__ in, uint64_t n,
                                                      uint64_t *__restrict__ cands, uint32_t max_cands,
                                                      uint32_t *__restrict__ n_cands)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = t * 16ull;
    if (base >= n) return;
    uint32_t w[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint64_t o = base + 4ull * k;
        uint32_t v = 0;
        if (o + 4 <= n) v = za_ld32(in + o);
        else for (int j = 0; j < 4; j++) if (o + (unsigned)j < n) v |= (uint32_t)in[o + (unsigned)j] << (8 * j);
        w[k] = v;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int wi = k >> 2, sh = (k & 3) * 8;
        const uint32_t v = sh ? ((w[wi] >> sh) | (w[wi + 1] << (32 - sh))) : w[wi];
        if (v != 0xFFFF0000u) continue;
        const uint64_t q = base + (uint64_t)k + 4ull;          // first byte after the marker
        if (q > n) continue;
        const uint32_t idx = atomicAdd(n_cands, 1u);
        if (idx < max_cands) cands[idx] = q;
    }
}

// the count pass gives up on a candidate after this much output without reaching a listed boundary: bounds the work a
// false candidate can cause; a real block that large (none of the common encoders emits one) sends the member to the
// sequential decoder
#define ZA_COUNT_CAP (64ull << 20)
// Both chunk kernels exist in two sizes (template argument BITS = sub-sequence length of the block decoder): with more
// chunks than fit the GPU at the large size (4 workgroups per CU) the small one wins by occupancy (11 per CU); with fewer,
// every wavefront is resident either way and the long sub-sequences, which re-synchronise in fewer passes, are faster.
// The host picks by the number of chunks (ZNGAMD_CHUNKS_SMALL_FROM in zng_amd.hip).
struct ZaChunkRes { int32_t status; uint32_t max_back; uint64_t bits; uint64_t out_len; };
struct ZaChunk { uint64_t in_bit; uint64_t out_off; uint64_t out_len; uint64_t end_bit; uint64_t src_off; };   // absolute bit offsets in the deflate stream; src_off = where the chunk's 16-bit symbols lie (out_off when they were decoded at their final place)

template <int BITS>
__global__ __launch_bounds__(64) void za_k_chunk_count(const uint8_t *__restrict__ in, uint64_t in_len,
                                                       const uint64_t *__restrict__ cands, uint32_t ncands,
                                                       ZaChunkRes *__restrict__ res,
                                                       uint64_t first_bit, uint32_t first_hist,   // where the stream (re)starts, and how much history it has there
                                                       int whole_streams = 0)                     // candidates are starts of separate streams (gzip members): each runs to its own end, no history
{
    __shared__ ZaInfTabs T;
    __shared__ int scratch[2];
    __shared__ ZaParBufT<BITS, 1, ZA_CHUNK_MAXIT> PS;
    const uint64_t abit = cands[blockIdx.x];                 // absolute bit offset of a possible block header
    const uint64_t off = abit >> 3;
    uint64_t bits = 0, op = 0;
    int status = ZA_I_DATA;
    if (off <= in_len)
        status = za_inflate_serial_core<1, uint8_t, ZA_WIN, ZaParBufT<BITS, 1, ZA_CHUNK_MAXIT>>(in + off, in_len - off, nullptr, 0, nullptr, ZA_COUNT_CAP, T, nullptr, scratch, PS.stage,
                                                    bits, op, (uint32_t)(abit & 7u), nullptr, nullptr,
                                                    whole_streams ? 0u : (abit == first_bit ? first_hist : (uint32_t)ZA_WIN), !whole_streams, nullptr,
                                                    cands, whole_streams ? 0u : ncands, off * 8ull, &PS);
    // bits = position relative to byte `off`; report the absolute end
    if (za_lane() == 0) { ZaChunkRes r; r.status = status; r.max_back = 0; r.bits = off * 8ull + bits; r.out_len = op; res[blockIdx.x] = r; }
}

template <int BITS, int Q, int RINGSYMS, int LB = ZA_LUT_L_BITS, int DB = ZA_LUT_D_BITS>       // RINGSYMS symbols of history in LDS; older sources are re-read from the chunk's own output; LB / DB: index bits of the decode tables
__global__ __launch_bounds__(64) void za_k_chunk_decode(const uint8_t *__restrict__ in, uint64_t in_len,
                                                        const ZaChunk *__restrict__ chunks, uint16_t *__restrict__ out16,
                                                        ZaChunkRes *__restrict__ res, uint64_t first_bit, uint32_t first_hist,
                                                        const uint64_t *__restrict__ stops = nullptr, uint32_t nstops = 0)
{
    // Two uses.  After a count pass: a chunk is a run of blocks whose extent and output size are known (out_len, end_bit), its
    // symbols go to their final place.  Without one (stops = every listed boundary, as the count pass has them): a chunk is ONE
    // candidate decoded to the next listed boundary or sync point into a scratch area of out_len symbols (src_off) -- sizes and
    // ends are found out here, and the host places the chunks afterwards.
    __shared__ ZaInfTabsT<LB, DB> T;
    __shared__ uint16_t win[RINGSYMS];
    __shared__ int scratch[2];
    __shared__ ZaParBufT<BITS, Q, ZA_CHUNK_MAXIT> P;
    const ZaChunk ch = chunks[blockIdx.x];
    const uint64_t off = ch.in_bit >> 3;
    uint64_t bits = 0, op = 0;
    uint32_t far = 0;
    int status = ZA_I_DATA;
    if (off <= in_len)
        status = za_inflate_serial_core<2, uint16_t, RINGSYMS, ZaParBufT<BITS, Q, ZA_CHUNK_MAXIT>>(in + off, in_len - off, nullptr, 0, out16 + ch.src_off, ch.out_len, T, win,
                                                     scratch, P.stage, bits, op, (uint32_t)(ch.in_bit & 7u), nullptr, nullptr,
                                                     ch.in_bit == first_bit ? first_hist : (uint32_t)ZA_WIN, stops != nullptr, &far,
                                                     stops ? stops : &chunks[blockIdx.x].end_bit, stops ? nstops : 1u, off * 8ull, &P);
    if (za_lane() == 0) { ZaChunkRes r; r.status = status; r.max_back = far; r.bits = off * 8ull + bits; r.out_len = op; res[blockIdx.x] = r; }
}

// ---- block finder: where could a dynamic-Huffman block header start? ----------------------------------------
// Phase A tests every bit offset cheaply: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29 and a COMPLETE code-length
// code (Kraft sum exactly 1 over the HCLEN+4 three-bit lengths).  Survivors (about 1 offset in 10^3) are compacted.
#define ZA_FINDA_THREADS 1024
#define ZA_FINDA_LOCAL   512          // survivors a workgroup collects in LDS before it reserves room for them with ONE global atomic
__global__ __launch_bounds__(ZA_FINDA_THREADS) void za_k_find_blocks_a(const uint8_t *__restrict__ in, uint64_t n,
                                                          uint64_t *__restrict__ surv, uint32_t max_surv, uint32_t *__restrict__ n_surv)
{
    // (an atomic per survivor on the one global counter -- 2 * 10^5 of them -- was what this kernel's time consisted of)
    __shared__ uint64_t found[ZA_FINDA_LOCAL];
    __shared__ uint32_t nfound, gbase;
    if (threadIdx.x == 0) nfound = 0;
    // Kraft sum of three 3-bit code lengths at once (units of 2^-7; a length of 0 = unused symbol adds nothing): the up to
    // nineteen lengths of a header are seven table reads instead of a data-dependent loop
    __shared__ uint8_t k3[512];
    for (uint32_t i = threadIdx.x; i < 512u; i += ZA_FINDA_THREADS) {
        const uint32_t a = i & 7u, b = (i >> 3) & 7u, c = i >> 6;
        k3[i] = (uint8_t)((a ? 128u >> a : 0u) + (b ? 128u >> b : 0u) + (c ? 128u >> c : 0u));
    }
    __syncthreads();
    // one thread per aligned dword of the input = 32 bit offsets; its four dword loads are coalesced across the wave
    // (a thread per byte with an unaligned 12-byte window made 64 separate accesses per wave instruction)
    const uint32_t mis = (uint32_t)((uintptr_t)in & 3u);
    const uint8_t *base = in - mis;                              // 4-byte aligned; bytes before `in` are never reported
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t first = 4ull * t;                             // offset from `base` of this thread's first byte
    const bool live = first + 16 <= n + mis;                     // the last few bytes of the buffer have no room for a header anyway
    const uint32_t *p32 = (const uint32_t *)(base + (live ? first : 0));
    const uint32_t d0 = p32[0], d1 = p32[1], d2 = p32[2], d3 = p32[3];
    const uint64_t q0 = ((uint64_t)d1 << 32) | d0, q2 = ((uint64_t)d3 << 32) | d2;       // the 16 bytes as two little-endian halves
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (!live || first + k < mis) continue;                  // in front of the stream
        const uint64_t byte = first + k - mis;
        if (byte + 12 > n) continue;                             // a real header is followed by far more than 12 bytes
        // 96-bit window at byte k of the 16 loaded bytes: lo = bits 0..63, hi = bits 64..95
        const uint32_t sh = 8u * k;
        const uint64_t lo = sh ? ((q0 >> sh) | ((uint64_t)d2 << (64u - sh))) : q0;
        const uint32_t hi = (uint32_t)(q2 >> sh);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const uint32_t head = (uint32_t)(lo >> b);               // 17 bits: BFINAL, BTYPE, HLIT, HDIST, HCLEN
            if ((head & 7u) != 4u) continue;                         // BFINAL = 0, BTYPE = 10b
            if (((head >> 3) & 31u) > 29u || ((head >> 8) & 31u) > 29u) continue;
            const uint32_t hclen = ((head >> 13) & 15u) + 4u;
            // the 3 * hclen <= 57 bits behind the 17 (bit b + 17 .. b + 73 of the 96-bit window)
            uint64_t w = (lo >> (b + 17u)) | ((uint64_t)hi << (47u - b));
            w &= (1ull << (3u * hclen)) - 1ull;
            uint32_t kraft = 0;
#pragma unroll
            for (uint32_t j = 0; j < 7; j++) kraft += k3[(uint32_t)(w >> (9u * j)) & 511u];
            if (kraft != 128u) continue;
            const uint32_t li = atomicAdd(&nfound, 1u);
            if (li < (uint32_t)ZA_FINDA_LOCAL) found[li] = byte * 8ull + (uint64_t)b;
            else { const uint32_t idx = atomicAdd(n_surv, 1u); if (idx < max_surv) surv[idx] = byte * 8ull + (uint64_t)b; }     // list full (rare)
        }
    }
    __syncthreads();
    const uint32_t cntl = nfound < (uint32_t)ZA_FINDA_LOCAL ? nfound : (uint32_t)ZA_FINDA_LOCAL;
    if (threadIdx.x == 0 && cntl) gbase = atomicAdd(n_surv, cntl);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cntl; i += ZA_FINDA_THREADS) if (gbase + i < max_surv) surv[gbase + i] = found[i];
}

// Phase B, one lane per survivor: decode the HLIT+HDIST code lengths with the code-length code and require what
// a decoder requires (no bad repeats, end-of-block present, both codes complete -- or a single 1-bit code).
__global__ __launch_bounds__(64) void za_k_find_blocks_b(const uint8_t *__restrict__ in, uint64_t n,
                                                         const uint64_t *__restrict__ surv, uint32_t n_surv,
                                                         uint64_t *__restrict__ cands, uint32_t max_cands, uint32_t *__restrict__ n_cands)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n_surv) return;
    const uint64_t abit = surv[i];
    const uint64_t in_bits = n * 8ull;
    uint64_t bp = abit + 3;
    const uint64_t hdr = za_peek(in, bp);
    const int nlen = (int)(hdr & 31u) + 257, ndist = (int)((hdr >> 5) & 31u) + 1, ncode = (int)((hdr >> 10) & 15u) + 4;
    bp += 14;
    // code-length code: canonical (count, first code, first index) per length from the 3-bit lengths
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t cll = 0, clh = 0;                                    // 19 x 3 bits packed: symbol s at bits 3s
    // a window of >= 57 bits is fetched once and used until fewer than 17 of them are left (a symbol with its repeat
    // count needs 14): one dependent load per ~8 symbols instead of one per symbol
    uint64_t wbase = bp, wcur = za_peek(in, bp);
    auto bits_at = [&](uint64_t at) -> uint32_t {
        if (at - wbase > 40ull) { wbase = at; wcur = za_peek(in, at); }
        return (uint32_t)(wcur >> (at - wbase));
    };
    for (int k = 0; k < ncode; k++) {
        const uint32_t v = bits_at(bp) & 7u; bp += 3;
        const int sft = 3 * order[k];
        if (sft < 30) cll |= v << sft; else clh |= v << (sft - 30);
    }
    auto cl_len = [&](int s) -> uint32_t { const int sft = 3 * s; return sft < 30 ? (cll >> sft) & 7u : (clh >> (sft - 30)) & 7u; };
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s2 = 0; s2 < 19; s2++) cnt[cl_len(s2)]++;
    // decode nlen + ndist lengths; track Kraft sums (units 2^-15) and the presence of symbol 256
    uint32_t kl = 0, kd = 0, nl_nonzero = 0, nd_nonzero = 0, maxl = 0, maxd = 0;
    bool eob = false, bad = false;
    int idx = 0, prev = 0;
    while (idx < nlen + ndist && !bad) {
        if (bp + 7 > in_bits) { bad = true; break; }
        const uint32_t bitsv = bits_at(bp);
        // canonical decode over the 19-symbol code, bit by bit
        int code = 0, first = 0, index = 0, sym = -1, l = 1;
        uint32_t v = bitsv;
        for (; l <= 7; l++) {
            code |= (int)(v & 1u); v >>= 1;
            const int c = (int)cnt[l];
            if (code - c < first) {
                // the (code - first)-th symbol, in symbol order, among those of length l
                const int want = code - first;
                int seen = 0;
                for (int s2 = 0; s2 < 19; s2++) if ((int)cl_len(s2) == l) { if (seen == want) { sym = s2; break; } seen++; }
                break;
            }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        if (sym < 0) { bad = true; break; }
        bp += (unsigned)l;
        int rep = 1, val = sym;
        if (sym == 16) { if (idx == 0) { bad = true; break; } val = prev; rep = 3 + (int)((bitsv >> l) & 3u); bp += 2; }
        else if (sym == 17) { val = 0; rep = 3 + (int)((bitsv >> l) & 7u); bp += 3; }
        else if (sym == 18) { val = 0; rep = 11 + (int)((bitsv >> l) & 127u); bp += 7; }
        if (idx + rep > nlen + ndist) { bad = true; break; }
        for (int r = 0; r < rep; r++, idx++) {
            if (val) {
                if (idx < nlen) { kl += 32768u >> val; nl_nonzero++; if ((uint32_t)val > maxl) maxl = (uint32_t)val; if (idx == 256) eob = true; }
                else { kd += 32768u >> val; nd_nonzero++; if ((uint32_t)val > maxd) maxd = (uint32_t)val; }
            }
        }
        prev = val;
        if (kl > 32768u || kd > 32768u) { bad = true; break; }      // over-subscribed already: most false survivors end here, after a few symbols
    }
    if (bad || !eob || bp > in_bits) return;
    const bool lit_ok = kl == 32768u || (kl < 32768u && maxl == 1);
    const bool dist_ok = nd_nonzero == 0 || kd == 32768u || (kd < 32768u && maxd == 1);
    if (!lit_ok || !dist_ok) return;
    const uint32_t o = atomicAdd(n_cands, 1u);
    if (o < max_cands) cands[o] = abit;
}

// Window propagation.  W(k) = the 32 KiB before chunk k.  W(k+1) = tail(k) applied to W(k), where tail(k) is the
// last 32 Ki symbols of chunk k (markers name bytes of W(k)).  Walking that chain chunk by chunk is serial, so it
// is done as a blocked scan over groups of ZA_CHUNK_GROUP chunks:
//   za_k_chunk_compose  one workgroup per group: comp[k] = tail(k) o ... o tail(first of group), still in markers
//                       of W(first of group)
//   za_k_chunk_chain    one workgroup: W(first of group g+1) = comp[last of group g] applied to W(first of group g)
//   za_k_chunk_resolve  one workgroup per chunk: W(k) from comp[k-1] and its group's window, then markers -> bytes
#define ZA_CHUNK_GROUP 64

__device__ __forceinline__ uint16_t za_tail_sym(const uint16_t *__restrict__ out16, const ZaChunk &ch, uint32_t j)
{
    // symbol at window position j of the window that follows chunk `ch`, in terms of the window before it
    const long long p = (long long)ch.out_len - ZA_WIN + (long long)j;
    return p < 0 ? (uint16_t)(256u + j + (uint32_t)ch.out_len) : out16[ch.src_off + (uint64_t)p];
}

__global__ __launch_bounds__(1024) void za_k_chunk_compose(const uint16_t *__restrict__ out16, const ZaChunk *__restrict__ chunks,
                                                           uint32_t n, uint16_t *__restrict__ comp)
{
    __shared__ __attribute__((aligned(16))) uint16_t cur[ZA_WIN];     // 64 KiB: composed map so far, updated in place
    const uint32_t tid = threadIdx.x;
    const uint32_t k0 = blockIdx.x * ZA_CHUNK_GROUP;
    const uint32_t k1 = min(n, k0 + ZA_CHUNK_GROUP);
    // The chunks of a group follow one another (each map is looked up in the one before), so what a step costs is its memory
    // round trip: every thread takes 32 CONSECUTIVE window positions (four 16-byte loads of the chunk's tail, four 16-byte stores
    // of the composed map) and has the next chunk's tail on its way while it looks this one's markers up.
    constexpr uint32_t PER = ZA_WIN / 1024;                           // 32 symbols = 4 x 16 bytes per thread
    ZaU4u raw[PER / 8], nxt[PER / 8];
    auto fetch = [&](uint32_t k, ZaU4u *dst) {
        const ZaChunk ch = chunks[k];
        if (ch.out_len >= (uint64_t)ZA_WIN) {
            const uint16_t *tail = out16 + ch.src_off + (ch.out_len - ZA_WIN) + (uint64_t)tid * PER;
#pragma unroll
            for (uint32_t t = 0; t < PER / 8; t++) dst[t] = *(const ZaU4u *)(tail + 8u * t);
        } else {                                                       // a chunk shorter than the window (a short stream's only one)
            uint32_t w[PER / 2];
#pragma unroll
            for (uint32_t t = 0; t < PER; t++) {
                const uint32_t sy = za_tail_sym(out16, ch, tid * PER + t);
                if (t & 1u) w[t >> 1] |= sy << 16; else w[t >> 1] = sy;
            }
#pragma unroll
            for (uint32_t t = 0; t < PER / 8; t++) { dst[t].x = w[4 * t]; dst[t].y = w[4 * t + 1]; dst[t].z = w[4 * t + 2]; dst[t].w = w[4 * t + 3]; }
        }
    };
    fetch(k0, raw);
    for (uint32_t k = k0; k < k1; k++) {
        if (k + 1 < k1) fetch(k + 1, nxt);
        uint16_t *dst = comp + (size_t)k * ZA_WIN + (size_t)tid * PER;
        uint4 vals[PER / 8];
#pragma unroll
        for (uint32_t t = 0; t < PER / 8; t++) {
            const uint32_t q[4] = {raw[t].x, raw[t].y, raw[t].z, raw[t].w};
            uint32_t r[4];
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                uint32_t lo = q[e] & 0xFFFFu, hi = q[e] >> 16;
                if (k != k0) {
                    if (lo >= 256u) lo = cur[lo - 256u];
                    if (hi >= 256u) hi = cur[hi - 256u];
                }
                r[e] = lo | (hi << 16);
            }
            vals[t] = make_uint4(r[0], r[1], r[2], r[3]);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t t = 0; t < PER / 8; t++) { *(uint4 *)(cur + tid * PER + 8u * t) = vals[t]; *(uint4 *)(dst + 8u * t) = vals[t]; }
        __syncthreads();
#pragma unroll
        for (uint32_t t = 0; t < PER / 8; t++) raw[t] = nxt[t];
    }
}

__global__ __launch_bounds__(1024) void za_k_chunk_chain(const uint16_t *__restrict__ comp, uint32_t n, uint8_t *__restrict__ winbuf,
                                                         const uint8_t *__restrict__ dict, uint32_t dict_len)   // history before chunk 0 (a resumed stream)
{
    __shared__ __attribute__((aligned(16))) uint8_t wa[ZA_WIN];
    __shared__ __attribute__((aligned(16))) uint8_t wb[ZA_WIN];
    uint8_t *cur = wa, *nxt = wb;
    const uint32_t tid = threadIdx.x;
    const uint32_t groups = (n + ZA_CHUNK_GROUP - 1) / ZA_CHUNK_GROUP;
    for (uint32_t j = tid; j < ZA_WIN; j += 1024) cur[j] = j >= ZA_WIN - dict_len ? dict[j - (ZA_WIN - dict_len)] : (uint8_t)0;
    __syncthreads();
    // one step per group, each waiting for the one before: the next group's map (64 KiB, 32 consecutive symbols per thread) is on
    // its way while this one's window is written out
    constexpr uint32_t PER = ZA_WIN / 1024;
    uint4 raw[PER / 8], ahead[PER / 8];
    auto fetch = [&](uint32_t g, uint4 *dst) {
        const uint16_t *cm = comp + (size_t)((g + 1) * ZA_CHUNK_GROUP - 1) * ZA_WIN + (size_t)tid * PER;
#pragma unroll
        for (uint32_t t = 0; t < PER / 8; t++) dst[t] = *(const uint4 *)(cm + 8u * t);
    };
    if (groups > 1) fetch(0, raw);
    for (uint32_t g = 0; g < groups; g++) {
        if (g + 2 < groups) fetch(g + 1, ahead);
        uint8_t *wout = winbuf + (size_t)g * ZA_WIN;
        for (uint32_t j = tid * 16; j < ZA_WIN; j += 16 * 1024) *(uint4 *)(wout + j) = *(const uint4 *)(cur + j);
        if (g + 1 == groups) break;
#pragma unroll
        for (uint32_t t = 0; t < PER / 8; t++) {
            const uint32_t q[4] = {raw[t].x, raw[t].y, raw[t].z, raw[t].w};
            uint32_t r[2] = {0, 0};
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) {
                const uint32_t sy = (q[e >> 1] >> (16u * (e & 1u))) & 0xFFFFu;
                r[e >> 2] |= (sy < 256u ? sy : (uint32_t)cur[sy - 256u]) << (8u * (e & 3u));
            }
            *(uint2 *)(nxt + tid * PER + 8u * t) = make_uint2(r[0], r[1]);
        }
        __syncthreads();
        uint8_t *t2 = cur; cur = nxt; nxt = t2;
#pragma unroll
        for (uint32_t t = 0; t < PER / 8; t++) raw[t] = ahead[t];
    }
}

#define ZA_RESOLVE_THREADS 512
__global__ __launch_bounds__(ZA_RESOLVE_THREADS) void za_k_chunk_resolve(const uint16_t *__restrict__ out16, const ZaChunk *__restrict__ chunks,
                                                          const uint16_t *__restrict__ comp, const uint8_t *__restrict__ winbuf,
                                                          uint8_t *__restrict__ out8)
{
    __shared__ __attribute__((aligned(16))) uint8_t wg[ZA_WIN];     // window before the group's first chunk
    __shared__ __attribute__((aligned(16))) uint8_t w[ZA_WIN];      // window before this chunk
    const uint32_t k = blockIdx.x, g = k / ZA_CHUNK_GROUP;
    const ZaChunk ch = chunks[k];
    const uint8_t *wsrc = winbuf + (size_t)g * ZA_WIN;
    const bool first = k == g * ZA_CHUNK_GROUP;
    uint8_t *wdst = first ? w : wg;
    for (uint32_t j = threadIdx.x * 16; j < ZA_WIN; j += 16 * ZA_RESOLVE_THREADS) *(uint4 *)(wdst + j) = *(const uint4 *)(wsrc + j);
    __syncthreads();
    if (!first) {
        const uint16_t *cm = comp + (size_t)(k - 1) * ZA_WIN;      // (64 KiB per chunk: 16-byte aligned)
        for (uint32_t j = threadIdx.x * 8; j < ZA_WIN; j += 8 * ZA_RESOLVE_THREADS) {
            const uint4 v = *(const uint4 *)(cm + j);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
            uint32_t r[2] = {0, 0};
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t sy = (q[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
                r[t >> 2] |= (sy < 256u ? sy : (uint32_t)wg[sy - 256u]) << (8 * (t & 3));
            }
            *(uint2 *)(w + j) = make_uint2(r[0], r[1]);
        }
        __syncthreads();
    }
    // The chunk's symbols become bytes, eight per thread and step: one aligned 16-byte load, a look-up in the window only for the
    // symbols that are markers (few: what a chunk copies from before its start), one 8-byte store.  (One symbol per thread and
    // step -- a two-byte load, a one-byte store -- made this the longest of the three window kernels: 512 dependent steps per
    // thread, two workgroups per CU by their LDS.)
    const uint16_t *src = out16 + ch.src_off;
    uint8_t *dst = out8 + ch.out_off;
    const uint64_t n = ch.out_len;
    const uint64_t head = min(n, (uint64_t)((8u - (uint32_t)(ch.src_off & 7ull)) & 7u));        // symbols in front of the first 16-byte aligned one
    if (threadIdx.x < head) { const uint32_t sym = src[threadIdx.x]; dst[threadIdx.x] = sym < 256u ? (uint8_t)sym : w[(sym - 256u) & (ZA_WIN - 1)]; }
    const uint64_t groups8 = (n - head) >> 3;
    for (uint64_t i = threadIdx.x; i < groups8; i += ZA_RESOLVE_THREADS) {
        const uint4 v = *(const uint4 *)(src + head + 8 * i);
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
        uint32_t r[2];
        if (((v.x | v.y | v.z | v.w) & 0xFF00FF00u) == 0u) {                                      // eight literals: pack the low bytes
            r[0] = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u);
            r[1] = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u);
        } else {
            r[0] = 0; r[1] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t sy = (q[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
                r[t >> 2] |= (sy < 256u ? sy : (uint32_t)w[(sy - 256u) & (ZA_WIN - 1)]) << (8 * (t & 3));
            }
        }
        ZaU2u o; o.x = r[0]; o.y = r[1];
        *(ZaU2u *)(dst + head + 8 * i) = o;
    }
    const uint64_t done = head + 8 * groups8;
    if (done + threadIdx.x < n) { const uint32_t sym = src[done + threadIdx.x]; dst[done + threadIdx.x] = sym < 256u ? (uint8_t)sym : w[(sym - 256u) & (ZA_WIN - 1)]; }
}

// One workgroup per member: header with the chunk index, deflate bytes from the unit slot, trailer.
__global__ __launch_bounds__(256) void za_k_assemble_members(const uint8_t *__restrict__ slots, uint32_t slot_stride,
                                                             const uint32_t *__restrict__ unit_len,
                                                             const uint32_t *__restrict__ unit_crc,
                                                             const uint32_t *__restrict__ cidx_ws,
                                                             const ZaUnit *__restrict__ units,
                                                             const uint64_t *__restrict__ member_off,   // offsets with header + trailer per member
                                                             uint8_t *__restrict__ dst, uint8_t xfl)
{
    const uint32_t u = blockIdx.x;
    const uint32_t dlen = unit_len[u];
    const uint32_t n = units[u].in_len;
    const uint32_t nchunk = (n + (1u << ZA_CHUNK_SHIFT) - 1u) >> ZA_CHUNK_SHIFT;
    const uint32_t hdr = ZA_MEMBER_HDR(nchunk);
    uint8_t *d = dst + member_off[u];
    const uint32_t size = hdr + dlen + 8;
    const uint32_t *ci = cidx_ws + (size_t)u * ZA_CIDX_STRIDE;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < hdr; i += blockDim.x) {
        uint8_t b = 0;
        switch (i) {
        case 0: b = 0x1f; break; case 1: b = 0x8b; break; case 2: b = 8; break; case 3: b = 4; break;
        case 8: b = xfl; break; case 9: b = 0xff; break;
        case 10: b = (uint8_t)((hdr - 12u) & 0xFF); break; case 11: b = (uint8_t)((hdr - 12u) >> 8); break;
        case 12: b = 'Z'; break; case 13: b = 'A'; break;
        case 14: b = (uint8_t)((hdr - 16u) & 0xFF); break; case 15: b = (uint8_t)((hdr - 16u) >> 8); break;
        case 24: b = (uint8_t)(nchunk & 0xFF); break; case 25: b = (uint8_t)(nchunk >> 8); break;
        case 26: b = 2; break; case 27: b = ZA_CHUNK_SHIFT; break;
        default:
            if (i >= 16 && i < 20) b = (uint8_t)(size >> (8 * (i - 16)));
            else if (i >= 20 && i < 24) b = (uint8_t)(n >> (8 * (i - 20)));
            else if (i >= 28) b = (uint8_t)(ci[(i - 28) >> 2] >> (8 * ((i - 28) & 3)));
        }
        d[i] = b;
    }
    const uint8_t *src = slots + (size_t)u * slot_stride;
    uint8_t *p = d + hdr;
    for (uint32_t i = tid; i < dlen; i += blockDim.x) p[i] = src[i];
    if (tid < 8) {
        const uint32_t v = tid < 4 ? unit_crc[u] : n;
        p[dlen + tid] = (uint8_t)(v >> (8 * (tid & 3)));
    }
}

// byte-wise equality of two device buffers (used by the round-trip checks of bench.py / smoke)
__global__ __launch_bounds__(256) void za_k_compare(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint64_t n,
                                                    unsigned long long *__restrict__ mismatches)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16ull;
    unsigned long long bad = 0;
    for (uint64_t o = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16ull; o < n; o += stride) {
        if (o + 16 <= n) {
            const uint4 x = *(const uint4 *)(a + o), y = *(const uint4 *)(b + o);
            bad += (x.x != y.x) + (x.y != y.y) + (x.z != y.z) + (x.w != y.w);
        } else for (uint64_t i = o; i < n; i++) bad += a[i] != b[i];
    }
    if (bad) atomicAdd(mismatches, bad);
}
