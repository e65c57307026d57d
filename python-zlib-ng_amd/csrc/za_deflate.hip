// DEFLATE kernels for MI355X (gfx950).  Product code; hand-written HIP, wave64.
//
// Replaces, for many independent blocks at once, what the reference does per block in
// ParallelCompress_compress_and_crc (reference src/zlib_ng/zlib_ngmodule.c:1696-1782):
// zng_deflateReset + zng_deflateSetDictionary + zng_crc32_z + zng_deflate(Z_SYNC_FLUSH).
//
// Pipeline over a batch of units (unit = <=128 KiB of input + <=32 KiB dictionary before it; at most 64 segments: 2 KiB each for
// full units, 32 bytes .. 1 KiB for the units of small calls):
//   k_chains<A|B|C>  one 4-wave workgroup per run of units, launched once per link table (5-, 3-, 12-byte contexts); 32-bit head
//             table (64 KiB) in LDS; the 64 positions of a step are inserted by ONE returning LDS atomic maximum, which also
//             yields every position's link
//   k_search  one 1024-thread workgroup per run; links of table A and the bytes of the sliding window staged in LDS rings;
//             every position searched in parallel: a walk over chain A, then its own links in tables B and C
//   k_dpstats + k_optparse (levels 4-9)  the unit's cost table from a quarter of its entries; then one wave per unit, one lane per
//             segment: backward dynamic programme over estimated bit costs, rewrites the search results so that the greedy parse
//             follows its choices
//   k_parse   one wave per unit, one lane per segment: greedy selection, symbol histogram with LDS atomics, CRC-32 per segment
//             folded with GF(2) products
//   k_plan    one wave per unit: length-limited canonical Huffman, block-type choice, header bits, the unit's exact size
//   k_pack    one wave per unit, token-parallel: bit lengths -> wave prefix sum -> an LDS ring of the stream's open dwords
#include "za_common.h"
#include "za_crc.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// k_chains
// ------------------------------------------------------------------------------------------------
// One stream (a run of consecutive units) per workgroup, and the whole insert of 64 consecutive positions is ONE LDS
// instruction: the head table holds 32-bit run-absolute positions, and every lane does a returning atomic maximum
// of its position into its bucket.  The LDS executes the lanes of an instruction that name one address one after the other,
// in ascending lane order, so what comes back to a lane is the largest position its bucket held before it: the nearest
// earlier position of the bucket -- an earlier lane of this very instruction or the table's entry from earlier steps --
// which is exactly the sequential insertion order of the spec (oracle stage 1), and the bucket is left with its last
// position.  That order is not promised anywhere, so it is checked: served in any other order, some lane of a bucket gets
// back a position that is not below its own, and the step's links are then worked out lane by lane (za_chains_fix; the
// table itself is right in any order, a maximum does not depend on it).  A wave's LDS operations execute in program order,
// so the step behind needs no wait: the atomics of consecutive steps go out back to back and their results are used a group
// of steps later -- no hand-over of positions between owners of bucket classes, no ordering of same-bucket lanes by hand.
// (Until round 3 a 256-thread workgroup split the bucket space over four waves, handed every position to the owning wave
// through LDS rings -- three ballots, eight mbcnt and two barriers per 256 positions -- and ordered the same-bucket lanes of a
// 64-entry insert with an exchange on a side array: 108 lane-instructions per position, 14.6 ms per 4 GiB.)
// Positions are 32 bits and absolute in the run, so nothing ages and nothing wraps: a link is valid if it reaches back at most
// 32 768.  The table is 2^14 x 4 bytes = 64 KiB: two streams per CU.  A wave issues an instruction every eight cycles or so
// however many others share its SIMD (profiles/ubench_occ.hip), so a stream is dealt to FOUR waves, a pipeline with one barrier
// per group of 1 024 positions (see the kernel): two hash, one issues the atomics, one makes the links and stores them.  (With
// a 32 KiB table and four streams per CU two waves per stream did as well -- the kernel is bound by the instructions a CU can
// issue, about 510 per group however they are dealt: 3.7 ms per 4 GiB with 13 bits and two waves, 4.2 with four; with 14 bits
// 6.7 with two waves and 4.8 with four.)
// No wave talks to memory position by position: 64 lanes fetching 6 bytes each at 64 consecutive byte addresses, and 64
// two-byte stores per step, kept the kernel waiting for the vector-memory pipe (three such instructions per step: 7.4 ms per
// 4 GiB whether one wave issued the rest or two).  The bytes of a group arrive as ONE 16-byte load per lane and pass through an
// LDS ring, from which every lane takes its own six (three aligned dwords, `v_alignbyte`); the links of a group are collected
// in LDS and leave as two 16-byte stores per lane.
#define ZA_CH_GROUP 16                         // steps (of 64 positions) per group
// the byte offset of a bucket's table entry out of the full 32-bit product: (x >> 19) << 2 with two full-rate operations
// (a right shift and an AND; a left shift runs at half rate on this chip, profiles/ubench_issue2.hip)
#define ZA_CH_OFFS(x) (((x) >> (32 - ZA_HASH_BITS - 2)) & (((1u << ZA_HASH_BITS) - 1u) << 2))
#define ZA_CH_CHUNK (64 * ZA_CH_GROUP)         // positions = bytes of a group: 16 per lane

// A 16-byte load whose wait is written by hand.  The compiler's own waits were the chain kernel's whole time: it staged a chunk
// behind `s_waitcnt vmcnt(0)` -- every load in flight, the one issued a moment ago included, because it had shuffled that load's
// registers -- so an iteration lasted as long as a trip to memory whatever the waves did in it.  Issued through an asm statement
// a load is invisible to that bookkeeping: the destination must not be touched before za_wait_loads names it (nothing else of
// the wave's loop is a vector-memory operation, so `newer` = the loads issued after the one that is needed), and every load
// must have been waited for before its registers can be anything else's (za_wait_loads(0) over all sets at the end).
typedef uint32_t za_v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void za_issue_load16(za_v4u32 &dst, const uint8_t *addr)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory");
}
template <int NEWER>
__device__ __forceinline__ void za_wait_loads(za_v4u32 &a, za_v4u32 &b, za_v4u32 &c, za_v4u32 &d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(NEWER) : "memory");
}

// the links of one step worked out lane by lane (only if the LDS ever served an atomic's lanes out of order): what the
// atomic should have returned
__device__ __noinline__ uint32_t za_chains_fix(uint32_t h, uint32_t A, uint32_t old, bool ins)
{
    uint32_t near = 0, pre = 0xFFFFFFFFu;
    const int lane = za_lane();
    const unsigned long long insm = __ballot(ins);
    for (int j = 0; j < 64; j++) {
        const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j), Aj = (uint32_t)__builtin_amdgcn_readlane((int)A, j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)old, j);
        if (((insm >> j) & 1ull) != 0ull && hj == h) {
            if (j < lane) near = Aj;               // ascending j: the last one kept is the nearest earlier lane of my bucket
            pre = oj < pre ? oj : pre;             // the bucket's entry in front of the step: what its first-served lane got back
        }
    }
    return near ? near : pre;
}

#ifdef ZA_CH_STATS
// profiling build only (profiles/abl_deflate.py prints them): per table and wave of the chain kernel, clocks spent at the barrier
// and in all ([8 * table + 2 * wave]); [24..27]: table A's inserting wave per group -- reads, atomics, writes (with a wait behind each), groups
__device__ unsigned long long za_ch_stat[32];
#endif
// TABLE: which of the three link tables (za_common.h): the context's length and hash are all that differs
template <int TABLE>
__global__ __launch_bounds__(256) void za_k_chains(const uint8_t *__restrict__ in, const ZaUnit *__restrict__ units,
                                                   const uint32_t *__restrict__ run_start,
                                                   uint16_t *__restrict__ prev_ws,
                                                   uint32_t *__restrict__ cost_ws = nullptr)     // (table A only) the units' "has a long match" words are cleared for the search
{
    // the bucket's last position as a run-absolute number that starts ABOVE the window size, 0 = none: `position - entry` is then
    // a valid link exactly if it is at most 32 768 (an empty bucket gives more), and it is below 1 exactly if the LDS served
    // the lanes of an atomic out of order (see above)
    __shared__ uint32_t head[1 << ZA_HASH_BITS];
    __shared__ __attribute__((aligned(16))) uint32_t ring[(2 * ZA_CH_CHUNK + 16) / 4];      // the bytes of two groups (+ the first 16 again behind the end)
    __shared__ __attribute__((aligned(16))) uint16_t hbuf[2][ZA_CH_CHUNK];                  // byte offsets of the buckets (bucket * 4) of a group's positions, two groups
    __shared__ __attribute__((aligned(16))) uint32_t obuf[2][ZA_CH_CHUNK];                  // what the atomics returned for a group (the links' raw form), two groups; the
                                                               // storing wave turns a group into links in place
    constexpr int HB = ZaTableBytes<TABLE>::value;               // bytes of a context: a position with fewer left in its unit is not inserted
    const uint32_t lane = (uint32_t)za_lane();
    // FOUR wavefronts per stream, a pipeline of three stages with one barrier per tick (a group of 1 024 positions per tick):
    // waves 0 and 1 hash group t (wave 0 also stages the bytes of group t + 1: it takes the second half of the group, which is
    // the half that needs them; wave 1 the first), wave 2 issues the atomics of group t - 1 (and checks their order), wave 3 turns
    // what they returned for group t - 2 into links and stores them.  The kernel is bound by the instructions a CU issues, and a
    // wave issues one every eight cycles or so whatever shares its SIMD: with a table of 64 KiB only two streams fit a CU, and
    // two waves per stream would leave its SIMDs half idle.
    const uint32_t role = threadIdx.x >> 6;
    const uint32_t u0 = run_start[blockIdx.x], u1 = run_start[blockIdx.x + 1];
#ifdef ZA_CH_STATS
    unsigned long long st_wait = 0, st_ph[4] = {0, 0, 0, 0};
    const unsigned long long st_t0 = clock64();
#endif
    uint32_t goff = 0;                                // positions of the run in front of the current unit
    uint32_t n_prev = 0;
#pragma unroll 1
    for (uint32_t ui = u0; ui < u1; ui++) {
        const ZaUnit u = units[ui];
        const int n = (int)u.in_len, dict_len = (int)u.dict_len;
        if (cost_ws && threadIdx.x == 0) cost_ws[(size_t)ui * ZA_DP_COSTS + 258] = 0u;
        const uint8_t *row = in + u.in_off - dict_len;                 // byte of row index 0 (row index i = position + dictionary length)
        uint16_t *prevdist = prev_ws + (size_t)ui * ZA_PREV_STRIDE;
#ifdef ZA_ABL_NO_CARRY
        const bool carry = false;
#else
        const bool carry = ui > u0 && (u.flags & ZA_FLAG_CARRY) != 0u;
#endif
        if (carry) goff += n_prev;
        else {
            // a fresh table: the unit's dictionary is inserted like the unit itself (what deflateSetDictionary does per block)
            goff = 0;
            for (uint32_t i = threadIdx.x * 4u; i < (1u << ZA_HASH_BITS); i += 1024u) *(uint4 *)&head[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();                                       // (the table; and all waves are through with the unit in front)
        n_prev = (uint32_t)n;
        const int total = dict_len + n;
        // a carried unit starts with the last HB - 1 positions of the unit in front of it: they had fewer than HB bytes left
        // there and have them now (their links go to this unit's own row: that unit's links say "never inserted", which is
        // what a search of THAT unit must see)
        const int first = carry ? dict_len - (HB - 1) : 0;
        const uint32_t abase = goff + (uint32_t)(2 * ZA_WIN - dict_len) + 1u;     // table value of row index i: abase + i (> 32 768)
        if (total < HB) {                           // (uniform) not one whole context: nothing to insert, every link is 0
            if (role == 2u && (int)lane >= first && (int)lane < total) prevdist[lane] = 0;
            continue;
        }
        if (total < 16) {
            // (uniform) a row shorter than one 16-byte load -- a stream of up to 15 bytes without a dictionary, nothing carried: lane i
            // of the inserting wave takes row index i and looks at the lanes below it (the table is left alone: nothing follows it)
            if (role == 2u) {
                uint32_t h = 0xFFFFFFFFu;
                const bool ins = (int)lane >= first && (int)lane <= total - HB;
                if (ins) {
                    // exactly the context's bytes, one by one (the row may end at the caller's last byte: no dword loads here)
                    uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
                    for (int b = 0; b < HB; b++) w[b >> 2] |= (uint32_t)row[lane + b] << (8 * (b & 3));
                    h = za_hash_x<TABLE>(w[0], w[1], w[2]) >> (32 - ZA_HASH_BITS);
                }
                uint32_t d = 0;
                for (int j = 0; j < 16; j++) {
                    const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j);
                    if (ins && j < (int)lane && hj == h) d = lane - (uint32_t)j;
                }
                if ((int)lane >= first && (int)lane < total) prevdist[lane] = (uint16_t)d;
            }
            continue;
        }
        const int iclamp_hi = total - HB;          // last row index with a whole context available
        const int t0 = first & ~63;                             // (groups start at multiples of 64: whole lines of links)
        const int ngroups = (total - t0 + ZA_CH_CHUNK - 1) / ZA_CH_CHUNK;
        // ---- wave 0.  Chunk c = the ZA_CH_CHUNK bytes from row index t0 + c * ZA_CH_CHUNK, 8 per lane; bytes behind the row's end
        // are zeros (the positions that would need them are never inserted)
        // (every fetch is ONE plain 16-byte load -- behind the row's end from a place moved back into the row, put right when
        // the chunk is staged)
        const int last16 = total - 16;                              // last row index a 16-byte load may start at (total >= 16 here)
        auto fetch = [&](int c, za_v4u32 &dst) {
            int at = t0 + c * ZA_CH_CHUNK + 16 * (int)lane;
            at = at > last16 ? last16 : at;
            za_issue_load16(dst, row + at);
        };
        auto stage = [&](int c, uint4 v) {                      // chunk c into its half of the ring
            const int cbase = t0 + c * ZA_CH_CHUNK;
            if (cbase + ZA_CH_CHUNK > total) {                  // (uniform) the row's last chunk and those behind it: zeros behind the row's end
                const int at = cbase + 16 * (int)lane;
                const int delta = at > last16 ? at - last16 : 0;    // the load started this many bytes in front of my place
                const uint32_t w[8] = {v.x, v.y, v.z, v.w, 0u, 0u, 0u, 0u};
                uint32_t r[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) { lo = (i + (delta >> 2)) == j ? w[j] : lo; hi = (i + (delta >> 2) + 1) == j ? w[j] : hi; }
                    r[i] = delta >= 16 ? 0u : __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)delta & 3u);
                }
                v = make_uint4(r[0], r[1], r[2], r[3]);
            }
            *(uint4 *)(ring + (c & 1) * (ZA_CH_CHUNK / 4) + 4 * lane) = v;
            if ((c & 1) == 0 && lane == 0) *(uint4 *)(ring + 2 * ZA_CH_CHUNK / 4) = v;
        };
        // half a group's positions hashed (steps g0 .. g0 + 7: 512 positions): needs chunk k and, for the second half, the head of
        // chunk k + 1.  A lane takes FOUR consecutive positions at a time out of one set of three aligned dwords (the four byte
        // offsets are constants: the first needs no alignment at all) and leaves their four bucket offsets as one 8-byte store --
        // three reads and one write per four positions instead of per position (a quarter fewer instructions in this stage).
        auto hash_half = [&](int k, int g0, uint16_t *hb) {
#ifdef ZA_ABL_CH_NOHASH
            return;
#endif
            const uint32_t *w = ring + (k & 1) * (ZA_CH_CHUNK / 4) + 16 * g0 + lane;     // dword of position 64 g0 + 4 lane
            uint16_t *o = hb + 64 * g0 + 4 * lane;
#pragma unroll
            for (int j = 0; j < ZA_CH_GROUP / 8; j++) {                                   // positions 64 g0 + 256 j + 4 lane + {0, 1, 2, 3}
                const uint32_t d0 = w[64 * j], d1 = w[64 * j + 1];
                uint32_t hh[4];
                if constexpr (TABLE == ZA_TABLE_A) {                                      // bytes 0..3 and byte 4 of each position
                    hh[0] = za_hash_x<TABLE>(d0, d1, 0u);
#pragma unroll
                    for (int q = 1; q < 4; q++) hh[q] = za_hash_x<TABLE>(__builtin_amdgcn_alignbyte(d1, d0, q), d1 >> (8 * q), 0u);
                } else if constexpr (TABLE == ZA_TABLE_B) {
                    hh[0] = za_hash_x<TABLE>(d0, 0u, 0u);
#pragma unroll
                    for (int q = 1; q < 4; q++) hh[q] = za_hash_x<TABLE>(__builtin_amdgcn_alignbyte(d1, d0, q), 0u, 0u);
                } else {
                    const uint32_t d2 = w[64 * j + 2], d3 = w[64 * j + 3];
                    hh[0] = za_hash_x<TABLE>(d0, d1, d2);
#pragma unroll
                    for (int q = 1; q < 4; q++)
                        hh[q] = za_hash_x<TABLE>(__builtin_amdgcn_alignbyte(d1, d0, q), __builtin_amdgcn_alignbyte(d2, d1, q), __builtin_amdgcn_alignbyte(d3, d2, q));
                }
                *(uint2 *)(o + 256 * j) = make_uint2(ZA_CH_OFFS(hh[0]) | (ZA_CH_OFFS(hh[1]) << 16), ZA_CH_OFFS(hh[2]) | (ZA_CH_OFFS(hh[3]) << 16));
            }
        };
        auto group_inner = [&](int k) -> bool {                 // (uniform) group k lies wholly inside the row: no test per lane
            const int tbase = t0 + k * ZA_CH_CHUNK;
            return tbase >= first && tbase + ZA_CH_CHUNK - 1 <= iclamp_hi;
        };
        // wave 2: a group's atomics go out back to back; what they return goes to the storing wave as it is (`inner`: the group
        // lies wholly inside the row -- all but a unit's first and last -- and needs no test per lane).  Only the order check is
        // made here, where the buckets still are in registers: a lane served out of order got back a position at or above its own
        auto insert_as = [&](auto inner_tag, int tbase, const uint16_t *hb, uint32_t *ob) {
            constexpr bool inner = decltype(inner_tag)::value;
            const uint32_t li = (uint32_t)tbase + lane;                 // my row index in the group's first step
            uint32_t hh[ZA_CH_GROUP], old[ZA_CH_GROUP];
            const uint32_t A0 = abase + li;
#ifdef ZA_CH_STATS
            const unsigned long long T0 = clock64();
#endif
#ifdef ZA_ABL_CH_HALFREADS
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g += 2) { const uint32_t t = *(const uint32_t *)&hb[64 * g + 2 * (lane >> 1)]; hh[g] = t & 0xFFFCu; hh[g + 1] = (t >> 16) & 0xFFFCu; }      // (timing only: the offsets two at a time)
#else
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g++) hh[g] = hb[64 * g + lane];
#endif
#ifdef ZA_CH_STATS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long T1 = clock64();
#endif
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g++) {
                const int i = (int)li + 64 * g;
                const bool ins = inner || (i >= first && i <= iclamp_hi);
                old[g] = 0u;
#ifdef ZA_ABL_CH_NOATOMIC
                if (ins) old[g] = *(const uint32_t *)((const uint8_t *)head + hh[g]);       // (timing only)
#else
                if (ins) old[g] = atomicMax((uint32_t *)((uint8_t *)head + hh[g]), A0 + 64u * (uint32_t)g);
#endif
            }
#ifdef ZA_CH_STATS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long T2 = clock64();
#endif
            uint32_t top = 0;                                          // bit 31 set: some step got back an entry at or above its own position
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g++) top |= (A0 + 64u * (uint32_t)g - old[g] - 1u);
#ifndef ZA_ABL_CH_NOCHECK
            if (__builtin_expect(__ballot((int)top < 0) != 0ull, 0)) {
#pragma unroll 1
                for (int g = 0; g < ZA_CH_GROUP; g++) {
                    const int i = (int)li + 64 * g;
                    const bool ins = inner || (i >= first && i <= iclamp_hi);
                    const uint32_t A = A0 + 64u * (uint32_t)g;
                    uint32_t hg = hh[0], og = old[0];
#pragma unroll
                    for (int q = 1; q < ZA_CH_GROUP; q++) { hg = g == q ? hh[q] : hg; og = g == q ? old[q] : og; }
                    const uint32_t ofix = za_chains_fix(hg >> 2, A, og, ins);
#pragma unroll
                    for (int q = 0; q < ZA_CH_GROUP; q++) old[q] = g == q ? ofix : old[q];
                }
            }
#endif
#ifdef ZA_ABL_CH_HALFWRITES
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g += 2) ob[64 * g + lane] = old[g] + old[g + 1];       // (timing only: half the inserting wave's writes)
#else
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g++) ob[64 * g + lane] = old[g];
#endif
#ifdef ZA_CH_STATS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            st_ph[0] += T1 - T0; st_ph[1] += T2 - T1; st_ph[2] += clock64() - T2; st_ph[3] += 1;
#endif
        };
        // wave 3: what the atomics returned becomes links (position - entry if that is at most 32 768; positions with fewer than HB
        // bytes left were never inserted: 0), which leave 16 bytes per lane twice where the group lies wholly inside the row (rows
        // and groups start at multiples of 128 bytes)
        auto store_links = [&](int k, uint32_t *ob) {
            const int tbase = t0 + k * ZA_CH_CHUNK;
            const uint32_t A0 = abase + (uint32_t)tbase + lane;
            uint32_t dd[ZA_CH_GROUP];
#pragma unroll
            for (int g = 0; g < ZA_CH_GROUP; g++) dd[g] = A0 + 64u * (uint32_t)g - ob[64 * g + lane];
            if (group_inner(k)) {
                uint16_t *lb = (uint16_t *)ob;                              // the links take the front half of the group's own buffer
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int g = 0; g < ZA_CH_GROUP; g++) lb[64 * g + lane] = (uint16_t)(dd[g] <= (uint32_t)ZA_WIN ? dd[g] : 0u);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint4 v0 = *(const uint4 *)&lb[8 * lane], v1 = *(const uint4 *)&lb[512 + 8 * lane];
#ifndef ZA_ABL_CH_NOSTORE
                *(uint4 *)(prevdist + tbase + 8 * (int)lane) = v0;
                *(uint4 *)(prevdist + tbase + 512 + 8 * (int)lane) = v1;
#endif
            } else {
#pragma unroll
                for (int g = 0; g < ZA_CH_GROUP; g++) {
                    const int i = tbase + (int)lane + 64 * g;
                    if (i >= first && i < total) prevdist[i] = (uint16_t)((i <= iclamp_hi && dd[g] <= (uint32_t)ZA_WIN) ? dd[g] : 0u);
                }
            }
        };
        auto wg_barrier = [&] {
#ifdef ZA_CH_STATS
            const unsigned long long b0 = clock64();
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef ZA_CH_STATS
            st_wait += clock64() - b0;
#endif
        };
        // Ticks t = 0 .. ngroups + 1, one barrier behind each (and one behind wave 0's prologue).  Wave 0 asks for the bytes FOUR
        // groups ahead -- chunk c travels in register set c & 3, its loop is unrolled four times so that no set is ever copied --
        // and its loop holds no other memory operation, so the wait in front of a chunk's staging is for that chunk alone.
        if (role == 0u) {
            za_v4u32 p0 = {0u, 0u, 0u, 0u}, p1 = p0, p2 = p0, p3 = p0;
            fetch(0, p0);
            za_wait_loads<0>(p0, p1, p2, p3);
            stage(0, make_uint4(p0.x, p0.y, p0.z, p0.w));
            fetch(1, p1); fetch(2, p2); fetch(3, p3); fetch(4, p0);
            wg_barrier();
            auto tick = [&](int t, za_v4u32 &pend) {             // pend: chunk t + 1 on arrival (the oldest of four in flight), chunk t + 5 on return
                if (t < ngroups) {
                    za_wait_loads<3>(p0, p1, p2, p3);
                    stage(t + 1, make_uint4(pend.x, pend.y, pend.z, pend.w));
                    fetch(t + 5, pend);
                    hash_half(t, ZA_CH_GROUP / 2, hbuf[t & 1]);
                }
                wg_barrier();
            };
#pragma unroll 1
            for (int t = 0; t <= ngroups + 1; t += 4) {
                tick(t, p1);
                if (t + 1 > ngroups + 1) break;
                tick(t + 1, p2);
                if (t + 2 > ngroups + 1) break;
                tick(t + 2, p3);
                if (t + 3 > ngroups + 1) break;
                tick(t + 3, p0);
            }
            za_wait_loads<0>(p0, p1, p2, p3);                    // (the chunks asked for behind the row's end: their registers are free only now)
        } else if (role == 1u) {
            wg_barrier();
#pragma unroll 1
            for (int t = 0; t <= ngroups + 1; t++) {
                if (t < ngroups) hash_half(t, 0, hbuf[t & 1]);
                wg_barrier();
            }
        } else if (role == 2u) {
            wg_barrier();
#pragma unroll 1
            for (int t = 0; t <= ngroups + 1; t++) {
                if (t >= 1 && t - 1 < ngroups) {
                    const int k = t - 1, tbase = t0 + k * ZA_CH_CHUNK;
                    if (group_inner(k)) insert_as(std::true_type{}, tbase, hbuf[k & 1], obuf[k & 1]);
                    else insert_as(std::false_type{}, tbase, hbuf[k & 1], obuf[k & 1]);
                }
                wg_barrier();
            }
        } else {
            wg_barrier();
#pragma unroll 1
            for (int t = 0; t <= ngroups + 1; t++) {
                if (t >= 2) store_links(t - 2, obuf[(t - 2) & 1]);
                wg_barrier();
            }
        }
    }
#ifdef ZA_CH_STATS
    if (lane == 0 && role == 2u && TABLE == 0) for (int i = 0; i < 4; i++) atomicAdd(&za_ch_stat[24 + i], st_ph[i]);
    if (lane == 0) { atomicAdd(&za_ch_stat[8 * TABLE + 2 * role], st_wait); atomicAdd(&za_ch_stat[8 * TABLE + 2 * role + 1], clock64() - st_t0); }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_search
// ------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per unit.  The sliding window lives in LDS: the chain links of the last
// 40960 positions (u16 ring) and the input bytes of the last 65536 positions (byte ring), both indexed
// by the absolute position P = 32768 + p.  A tile of 4096 positions is staged per barrier; every
// position of the tile is then searched in parallel with LDS traffic only.
#ifndef ZA_SEARCH_CONSEC
#define ZA_SEARCH_CONSEC 1         // levels 1-6: a thread takes four CONSECUTIVE positions of a tile (0: positions 1 024 apart, as until r06 -- see the tile loop)
#endif
#ifndef ZA_STATS_FOLD
#define ZA_STATS_FOLD 0            // 1: the dynamic programme's cost statistics taken inside the search of levels 4-6 (measured, lost: see the kernel)
#endif
#define ZA_SEARCH_THREADS 1024
#define ZA_SEARCH_TILE    4096
#define ZA_RING           40960            // chain-link ring: window 32768 + two tiles
#define ZA_BYTES          65536            // byte ring (power of two)
#define ZA_LOOKAHEAD      272              // bytes staged beyond the tile: max match 258 + wide compares
// Levels that compare in full walk 8-16 chain steps, and most positions are done long before the last one (text at level 9:
// 7.8 steps on average, 31 % of the positions take all 16): a wave walks ZA_WL_STEPS steps at a time, and the positions that
// are not done yet go to a work list of the wave's own in LDS (two dwords each), from which the wave takes 64 at a time as
// soon as it holds that many -- the later steps run on full waves.  At most 63 entries wait when 64 more arrive.
#ifndef ZA_WL_STEPS
#define ZA_WL_STEPS       4
#endif
#define ZA_WL_ENTRIES     127
#define ZA_WL_BYTES       (16 * ZA_WL_ENTRIES * 8)

// byte-aligned dword of the byte ring at absolute position `addr` (the ring has mirrored pad dwords behind its end)
__device__ __forceinline__ uint32_t za_lds_ld32(const uint32_t *win32, uint32_t addr)
{
    const uint32_t idx = addr & (ZA_BYTES - 1), w = idx >> 2;
    return __builtin_amdgcn_alignbyte(win32[w + 1], win32[w], idx & 3u);     // win32 has 4 mirrored pad dwords
}

// `best` entry of a position: distance - 1 (bits 0..14) | length (bits 15..23, 0: no match) | the position's own byte << 24 --
// the parse kernel then needs nothing but these entries (no second pass over the input)

// (Measured this round and dropped -- the kernel is bound by its instruction count, neither by LDS latency nor by LDS bank
// cycles: two or four positions of a thread searched at once, their chain walks interleaved by hand, 6.82 -> 6.80 / 7.13 ms per
// GiB; a candidate's 16 bytes as ONE unaligned ds_read_b128 -- the LDS takes wide reads at any byte address -- 9.4 ms; as three
// aligned ds_read_b64 and a select per dword 7.57 ms.)
// the match of `len` bytes between the window positions qb and P, extended to its true length (at most maxlen): 8 bytes per round,
// each side from three aligned dwords (the ring's mirrored pad covers the overrun); most matches end in the first round
__device__ __forceinline__ int za_search_extend(const uint32_t *win32, uint32_t qb, uint32_t P, int len, int maxlen)
{
    {   // first round: 8 bytes (most matches end here)
        const uint32_t o = (uint32_t)len;
        const uint32_t is = (qb + o) & (ZA_BYTES - 1), ws = is >> 2, ip = (P + o) & (ZA_BYTES - 1), wp = ip >> 2;
        const uint32_t s0 = win32[ws], s1 = win32[ws + 1], s2 = win32[ws + 2], p0 = win32[wp], p1 = win32[wp + 1], p2 = win32[wp + 2];
        const uint32_t a0 = __builtin_amdgcn_alignbyte(s1, s0, is & 3u) ^ __builtin_amdgcn_alignbyte(p1, p0, ip & 3u);
        const uint32_t a1 = __builtin_amdgcn_alignbyte(s2, s1, is & 3u) ^ __builtin_amdgcn_alignbyte(p2, p1, ip & 3u);
        uint32_t g0, g1;
        asm("v_ffbl_b32 %0, %2\n\tv_ffbl_b32 %1, %3\n\tv_add_u32_e64 %1, %1, 32 clamp\n\tv_min_u32_e32 %0, %0, %1"
            : "=&v"(g0), "=&v"(g1) : "v"(a0), "v"(a1));
        const int nb = (int)min(g0 >> 3, 8u);
        len += nb;
        if (nb < 8 || len >= maxlen) return len < maxlen ? len : maxlen;
    }
    for (;;) {   // a long match: 16 bytes per round (five aligned dwords a side)
        const uint32_t o = (uint32_t)len;
        const uint32_t is = (qb + o) & (ZA_BYTES - 1), ws = is >> 2, ip = (P + o) & (ZA_BYTES - 1), wp = ip >> 2;
        const uint32_t s0 = win32[ws], s1 = win32[ws + 1], s2 = win32[ws + 2], s3 = win32[ws + 3], s4 = win32[ws + 4];
        const uint32_t p0 = win32[wp], p1 = win32[wp + 1], p2 = win32[wp + 2], p3 = win32[wp + 3], p4 = win32[wp + 4];
        const uint32_t a0 = __builtin_amdgcn_alignbyte(s1, s0, is & 3u) ^ __builtin_amdgcn_alignbyte(p1, p0, ip & 3u);
        const uint32_t a1 = __builtin_amdgcn_alignbyte(s2, s1, is & 3u) ^ __builtin_amdgcn_alignbyte(p2, p1, ip & 3u);
        const uint32_t a2 = __builtin_amdgcn_alignbyte(s3, s2, is & 3u) ^ __builtin_amdgcn_alignbyte(p3, p2, ip & 3u);
        const uint32_t a3 = __builtin_amdgcn_alignbyte(s4, s3, is & 3u) ^ __builtin_amdgcn_alignbyte(p4, p3, ip & 3u);
        uint32_t g0, g1, g2, g3;
        asm("v_ffbl_b32 %0, %4\n\tv_ffbl_b32 %1, %5\n\tv_ffbl_b32 %2, %6\n\tv_ffbl_b32 %3, %7\n\t"
            "v_add_u32_e64 %1, %1, 32 clamp\n\tv_add_u32_e64 %2, %2, 64 clamp\n\tv_add_u32_e64 %3, %3, %8 clamp\n\t"
            "v_min3_u32 %0, %0, %1, %2\n\tv_min_u32_e32 %0, %0, %3"
            : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(96u));
        const int nb = (int)min(g0 >> 3, 16u);
        len += nb;
        if (nb < 16 || len >= maxlen) break;
    }
    return len < maxlen ? len : maxlen;
}

// 4 * log2(a / b) in whole quarter bits, a >= b >= 1, a < 2^22
__device__ __forceinline__ int za_ilog4(uint32_t a, uint32_t b)
{
    const uint32_t q = (a << 8) / b;                  // >= 256
    const int lg = 23 - (int)__builtin_clz(q);
    const uint32_t t = q >> lg;                       // 256 .. 511
    return 4 * lg + (t >= 304u ? 1 : 0) + (t >= 362u ? 1 : 0) + (t >= 431u ? 1 : 0);
}

// FULL: candidates are compared in full (levels with cap 258); otherwise on 16 bytes, winner extended afterwards
// STEPS: the chain steps of the level as a constant (1 .. 3: the walk is unrolled, no loop counter, no loop) or 0 = L.chain
// USEC: table C's candidate is tried too (levels 5-9)
// Candidates of a position: the first L.chain entries of its chain in table A (a walk through the link ring in LDS), then its own
// link in table B and in table C (the nearest earlier 3- / 12-byte context: one global 2-byte load each, no walk).  Longest wins,
// nearest wins ties; the levels that compare in full stop at L.nice equal bytes.  On the 16-byte levels the candidates of B and C
// are only TESTED for their context (3 / 12 equal bytes = a match of that length) and a winner among them is extended afterwards.
template <bool FULL, int STEPS = 0, bool USEC = false>
__global__ __launch_bounds__(ZA_SEARCH_THREADS) void za_k_search(const uint8_t *__restrict__ in, uint64_t in_total,
                                                                 const ZaUnit *__restrict__ units,
                                                                 const uint32_t *__restrict__ run_start,
                                                                 const uint16_t *__restrict__ prev_ws,
                                                                 const uint16_t *__restrict__ linkb_ws,
                                                                 const uint16_t *__restrict__ linkc_ws,
                                                                 uint32_t *__restrict__ best_ws, uint32_t *__restrict__ cost_ws, ZaLevel L)
{
    // One block of LDS with the byte window FIRST: a candidate's LDS address is then its position's low 16 bits (no base to add),
    // and the link ring is walked with byte addresses that have its base in them (no shift and no add per step).  A link of
    // 0xFFFF in the ring = end of the chain (what the chain kernel writes as 0): the walk's "too far" test ends it, one compare
    // instead of two per step.  (8 % fewer vector instructions per step, 1.5 % of the kernel's time: the candidates' LDS reads --
    // five dwords at a random address, 26 LDS cycles per wave -- weigh as much as the instructions.)
    __shared__ __attribute__((aligned(16))) uint8_t lds[ZA_BYTES + 32 + 2 * ZA_RING + (FULL ? ZA_WL_BYTES : 0)];
    // (r06) the dynamic programme's cost statistics (za_k_dpstats until round 5: a kernel and a pass over a quarter of the entries
    // of its own) are taken here, from the entries this workgroup has just written: levels 4-6.  The levels that compare in full
    // have no LDS left for the histogram (their work lists take the CU's last 16 KiB) and keep the separate kernel.
    constexpr bool STATS = !FULL && ZA_STATS_FOLD;
    __shared__ uint32_t st_hist[STATS ? 256 : 1];
    __shared__ uint32_t st_cnt[4];                               // U, NM, the sum T of the smoothed histogram
    uint32_t *win32 = (uint32_t *)lds;
    uint16_t *ring = (uint16_t *)(lds + ZA_BYTES + 32);
    constexpr uint32_t RING_B0 = ZA_BYTES + 32, RING_BYTES = 2 * ZA_RING;      // the ring's byte range in the block
    const int tid = (int)threadIdx.x;
    auto link_in = [](uint16_t v) -> uint16_t { return v ? v : (uint16_t)0xFFFFu; };
    // Links travel FOUR at a time: a thread loads the links of four consecutive positions as 8 bytes and puts them into the ring
    // with one 8-byte store (the ring's size is a multiple of four entries, so a group of four never wraps inside); 0 -> 0xFFFF
    // in both halves of a dword at once: (x - 1) + 1 with the addition saturating.  One position of a ring slot, one load and one
    // store per four links instead of four of each: 30 fewer instructions per thread and tile (4 % of the kernel's).
    auto quad_in = [](uint2 q) -> uint2 {
        uint2 r;
        asm("v_pk_sub_u16 %0, %2, %4\n\tv_pk_sub_u16 %1, %3, %4\n\tv_pk_add_u16 %0, %0, %4 clamp\n\tv_pk_add_u16 %1, %1, %4 clamp"
            : "=&v"(r.x), "=&v"(r.y) : "v"(q.x), "v"(q.y), "s"(0x00010001u));
        return r;
    };
    // One workgroup per RUN of consecutive units -- the chain kernel's runs.  Behind the head of a run every unit's dictionary
    // is the tail of the unit in front of it, and the window of that unit is still in the rings: the walk goes on where it
    // stood instead of staging 32 KiB of links and bytes again with all sixteen waves waiting.  `goff` = the run's positions
    // in front of the current unit: the rings are indexed by goff + 32768 + p, everything else by p.
    const uint32_t u0 = run_start[blockIdx.x], u1 = run_start[blockIdx.x + 1];
    uint32_t goff = 0;
    int carry_bytes = 0, n_prev = 0;
    const bool stats = STATS && L.dp != 0;                         // (uniform)
    if (stats) { if (tid < 256) st_hist[tid] = 0u; if (tid < 4) st_cnt[tid] = 0u; }     // (the first tile's staging barrier orders these)
#pragma unroll 1
    for (uint32_t ui = u0; ui < u1; ui++) {
    const ZaUnit u = units[ui];
    const uint8_t *data = in + u.in_off;
    const int n = (int)u.in_len, dict_len = (int)u.dict_len;
    const uint16_t *prevdist = prev_ws + (size_t)ui * ZA_PREV_STRIDE;     // index p + dict_len
    const uint16_t *linkb = linkb_ws + (size_t)ui * ZA_PREV_STRIDE + dict_len;      // index p (own positions only: no walk through these)
    const uint16_t *linkc = linkc_ws + (size_t)ui * ZA_PREV_STRIDE + dict_len;
    const int sshift = ZA_UNIT_SEG_SHIFT(u.flags);      // log2 of the unit's segment size (matches end at segment ends)
    const bool carried = ui > u0;           // (the host cuts a run wherever a unit's dictionary is not the tail of its predecessor)
    uint32_t *best = best_ws + (size_t)ui * ZA_BEST_STRIDE;
    // bytes that may be read starting at data[0] without leaving the caller's buffer
    const long long readable = (long long)(in_total - u.in_off);
    goff = carried ? goff + (uint32_t)n_prev : 0u;
    bool longm = false;                              // some match of the unit is longer than the dynamic programme's ring (ZA_DP_NEAR): it then keeps acc[] in memory too
    int links_loaded = carried ? 0 : -dict_len;                        // positions p < links_loaded have their chain link in the ring
    int bytes_loaded = carried ? carry_bytes - n_prev : (-dict_len) & ~3;   // positions p < bytes_loaded have their byte in the byte ring (aligned dwords)
    if (carried && tid < ZA_HASH_BYTES_A - 1)
        // the last four positions of the unit in front: never inserted there (their links in the ring say so), inserted by
        // this unit's chain pass (its own row)
        ring[(goff + (uint32_t)(ZA_WIN - (ZA_HASH_BYTES_A - 1) + tid)) % ZA_RING] = link_in(prevdist[dict_len - (ZA_HASH_BYTES_A - 1) + tid]);
    auto load_quad = [&](int p, int limit) -> uint2 {    // links of positions p .. p + 3 (what lies at and behind `limit` is not used)
        // (one plain predicated load: a branch with narrower loads in it makes the compiler wait for every load in flight at the
        // join.  The last group of a unit reads up to three entries past its links: inside the row, or the workspace's slack)
        uint2 q = make_uint2(0u, 0u);
        if (p < limit) { const ZaU2u t = *(const ZaU2u *)(prevdist + p + dict_len); q.x = t.x; q.y = t.y; }
        return q;
    };
    auto store_quad = [&](int p, int limit, uint2 q) {   // ... into the ring; entries at and behind `limit` are left alone
        const uint32_t X = goff + (uint32_t)(ZA_WIN + p);
        q = quad_in(q);
        if ((X & 3u) == 0u && p + 4 <= limit) *(uint2 *)(lds + RING_B0 + 2u * (X % ZA_RING)) = q;
        else {
            for (int k = 0; k < 4; k++)
                if (p + k < limit) ring[(X + (uint32_t)k) % ZA_RING] = (uint16_t)((k < 2 ? q.x : q.y) >> (16 * (k & 1)));
        }
    };
    auto load_bytes = [&](int p) -> uint32_t {          // dword of input at p (a multiple of 4), zero outside the unit
        if (p >= -dict_len && (long long)p + 4 <= readable) return za_ld32(data + p);
        uint32_t v = 0;                                   // edges: before the dictionary start or past the caller's buffer
        for (int k = 0; k < 4; k++)
            if (p + k >= -dict_len && (long long)(p + k) < readable) v |= (uint32_t)data[p + k] << (8 * k);
        return v;
    };
    auto store_bytes = [&](int p, uint32_t v) {
        const uint32_t w = ((goff + (uint32_t)(ZA_WIN + p)) & (ZA_BYTES - 1)) >> 2;
        win32[w] = v;
        if (w < 8) win32[w + ZA_BYTES / 4] = v;
    };
    // ---- the first tile is staged up front: chain links up to the tile end, bytes up to tile end + lookahead
    {
        int need_links = ZA_SEARCH_TILE < n ? ZA_SEARCH_TILE : n;
        for (int p = links_loaded + 4 * tid; p < need_links; p += 4 * ZA_SEARCH_THREADS) store_quad(p, need_links, load_quad(p, need_links));
        links_loaded = need_links;
        int need_bytes = ZA_SEARCH_TILE + ZA_LOOKAHEAD;
        if (need_bytes > n) need_bytes = n;
        need_bytes = (need_bytes + 3) & ~3;
        // (the plain loop -- every dword inside the caller's buffer and the dictionary: all but a unit at an end of the buffer or
        // with a dictionary of odd length -- keeps eight loads in flight; the careful one, with its branches, one)
        if ((carried || bytes_loaded == -dict_len) && (long long)need_bytes <= readable)
            for (int p = bytes_loaded + 4 * tid; p < need_bytes; p += 4 * ZA_SEARCH_THREADS) store_bytes(p, za_ld32(data + p));
        else
            for (int p = bytes_loaded + 4 * tid; p < need_bytes; p += 4 * ZA_SEARCH_THREADS) store_bytes(p, load_bytes(p));
        bytes_loaded = need_bytes;
    }
    // The positions' own links in tables B and C travel ONE TILE AHEAD in registers (fetched where they are used, behind the first
    // position's walk, every tile stalled for their trip to memory).  Eight 2-byte loads per thread and tile: 12 % of the kernel
    // (`-DZA_ABL_NO_LINKLOADS`, r06: 7.68 -> 6.76 ms per GiB) with everything they fetch sitting in the L2 -- a third of it the
    // loads' address arithmetic, which is why they are unconditional now (a position behind the unit's end reads the unit's last
    // link, and is never searched) with 32-bit offsets from a scalar base.  Measured and dropped: one 8-byte load per table and
    // thread, each lane picking its four links out of its neighbours' registers with ds_bpermute -- 2 loads and 16 permutes
    // instead of 8 loads: 7.74 -> 7.95 ms per GiB, the permutes alone cost what the loads did.
    // (r06, ZA_OWN_STAGE) ... and through LDS: a thread fetches EIGHT consecutive links of one table with one 16-byte load (threads
    // 0-511 table B, 512-1023 table C: one vector-memory instruction per thread and tile instead of eight), writes them at the
    // tile's end into a staging area, and at the top of the next tile every thread reads its own eight back (2-byte LDS reads at
    // consecutive addresses: conflict-free).  The area needs 16 KiB and the kernel has 15.9 left: it lies in the BYTE RING, in the
    // part no walk can reach at that time -- positions P + 4 608 .. P + 20 992 behind the tile's first position P (the ring holds
    // valid bytes from P - 36 864 up to P + 8 468 while the area lives: written at the end of the tile in front, read at the top
    // of this one; 65 536 - 45 332 = 20 204 bytes are nobody's).  One more barrier per tile, right behind the reads (the waves
    // have just left the tile's barrier: it costs what its few instructions cost), keeps a fast wave's next write off a slow
    // wave's reads.
    // MEASURED AND NOT TAKEN (the switch stays for the comparison): level 6 7.69 against 7.69 ms per GiB, levels 1 and 4 3.5 % SLOWER
    // (4.71 / 4.54, 5.71 / 5.52) -- the second barrier and the nine LDS operations cost what the seven loads saved; what the
    // ablation without the loads had promised (0.9 ms per GiB) was its synthetic candidates' doing, not the loads'.
#ifndef ZA_OWN_STAGE
#define ZA_OWN_STAGE 0
#endif
    za_v4u32 oq = {0u, 0u, 0u, 0u};
    auto own_area = [&](int tile_base) -> uint32_t { return (goff + (uint32_t)(ZA_WIN + tile_base) + 4608u + 15u) & 0xFFF0u; };      // (16-byte aligned, inside the ring)
    auto own_load = [&](int tile_base) {                           // my eight links of that tile
        const bool forc = tid >= ZA_SEARCH_THREADS / 2;
        const int p = tile_base + 8 * (tid & (ZA_SEARCH_THREADS / 2 - 1));
        oq = za_v4u32{0u, 0u, 0u, 0u};
#ifdef ZA_ABL_NO_LINKLOADS
        oq.x = oq.y = oq.z = oq.w = (((uint32_t)p * 7u & 0xFFu) | 1u) * 0x00010001u;
#else
        // (the last piece of a unit reads up to seven entries past its links: inside the row -- B and C rows are as long as A's)
        if (p < n && (USEC || !forc)) { const ZaU4u t = *(const ZaU4u *)((forc ? linkc : linkb) + p); oq.x = t.x; oq.y = t.y; oq.z = t.z; oq.w = t.w; }
#endif
    };
    auto own_store = [&](int tile_base) {
        const uint32_t a = (own_area(tile_base) + (tid >= ZA_SEARCH_THREADS / 2 ? 8192u : 0u) + 16u * (uint32_t)(tid & (ZA_SEARCH_THREADS / 2 - 1))) & 0xFFFFu;
        *(uint4 *)(lds + a) = make_uint4(oq.x, oq.y, oq.z, oq.w);
    };
    uint32_t nlkb[ZA_SEARCH_TILE / ZA_SEARCH_THREADS], nlkc[ZA_SEARCH_TILE / ZA_SEARCH_THREADS];
    constexpr bool CONSEC = !FULL && ZA_SEARCH_CONSEC && !ZA_OWN_STAGE && !ZA_STATS_FOLD;
    uint2 nob = make_uint2(0u, 0u), noc = make_uint2(0u, 0u);     // (CONSEC) my four positions' links of the next tile as they were loaded: taken apart a tile later
    auto load_own_links = [&](int tile_base) {
        if (ZA_OWN_STAGE) { own_load(tile_base); return; }
        if (CONSEC) {
            // four consecutive links of a table are ONE 8-byte load (a position behind the unit's end reads the unit's last links,
            // and is never searched; the load may reach three entries past the unit's links: inside the row)
            const int p = tile_base + 4 * tid;
            const uint32_t off = 2u * (uint32_t)(p < n ? p : (n - 1) & ~3);
#ifdef ZA_ABL_NO_LINKLOADS
            nob = make_uint2(0x00010001u, 0x00010001u); noc = nob;
#else
            if (n > 0) { const ZaU2u t = *(const ZaU2u *)((const uint8_t *)linkb + off); nob.x = t.x; nob.y = t.y; }
            if (USEC && n > 0) { const ZaU2u t = *(const ZaU2u *)((const uint8_t *)linkc + off); noc.x = t.x; noc.y = t.y; }
#endif
            return;
        }
#pragma unroll
        for (int k = 0; k < ZA_SEARCH_TILE / ZA_SEARCH_THREADS; k++) {
            const int p = tile_base + k * ZA_SEARCH_THREADS + tid;
#ifdef ZA_ABL_NO_LINKLOADS
            nlkb[k] = ((uint32_t)p * 7u & 0xFFu) | 1u; nlkc[k] = ((uint32_t)p * 13u & 0x3FFu) | 1u;      // (timing only: candidates without the loads)
#else
            const uint32_t off = 2u * (uint32_t)(p < n ? p : n - 1);        // (n >= 1 inside the tile loop; the prologue's call guards itself)
            nlkb[k] = n > 0 ? (uint32_t)*(const uint16_t *)((const uint8_t *)linkb + off) : 0u;
            nlkc[k] = USEC && n > 0 ? (uint32_t)*(const uint16_t *)((const uint8_t *)linkc + off) : 0u;
#endif
        }
    };
    load_own_links(0);
    if (ZA_OWN_STAGE) own_store(0);                               // (the barrier in front of the tile loop makes it visible)
    // (r06, levels 1-6) A tile's four results per thread stay in registers and are stored at the top of the NEXT tile, behind the
    // take-over of the own links and in front of the new loads (the unit's last tile: behind the loop).  Stored where they were
    // made, the last of them was in flight when the tile's end asked for the links and bytes fetched at its top -- stores count
    // in vmcnt on this chip, and behind the divergent search the compiler's wait is for everything: every wave sat out a store's
    // round trip per tile.
    uint32_t res[ZA_SEARCH_TILE / ZA_SEARCH_THREADS];
    int res_base = -1;                                             // tile whose results wait in res[], -1: none
    auto flush_results = [&]() {
        if (res_base < 0) return;
        if (CONSEC) {
            const int p = res_base + 4 * tid;                         // four consecutive entries: one 16-byte store
            if (p + 4 <= n) *(uint4 *)(best + p) = make_uint4(res[0], res[1], res[2], res[3]);
            else {
#pragma unroll
                for (int k = 0; k < 4; k++) if (p + k < n) best[p + k] = res[k];
            }
            res_base = -1;
            return;
        }
#pragma unroll
        for (int k = 0; k < ZA_SEARCH_TILE / ZA_SEARCH_THREADS; k++) {
            const int p = res_base + k * ZA_SEARCH_THREADS + tid;
            if (p < n) best[p] = res[k];
        }
        res_base = -1;
    };
    // ---- cost statistics (oracle dp_costs): the SAMPLE is every fourth block of 256 positions, i.e. four blocks per
    // tile and one position per thread.  The thread reads its position's entry and the one in front of it back from memory -- this
    // workgroup wrote them, the tile's barrier lies in between -- one tile LATE: the loads are issued at the top of the next
    // tile and used at its end, so that nobody waits for them.  U / NM in registers until the unit ends, the bytes by LDS atomics.
    uint32_t st_e = 0u, st_pe = 0u, st_U = 0u, st_NM = 0u;
    bool st_have = false;
    auto stats_issue = [&](int tile_base) {
        const int p = tile_base + ((tid >> 8) << 10) + (tid & 255);
        st_have = p < n;
        st_e = 0u; st_pe = 0u;
        if (st_have) { st_e = best[p]; if (p > 0) st_pe = best[p - 1]; }
    };
    auto stats_use = [&]() {
        const uint32_t len = ZA_ELEN(st_e), lp = ZA_ELEN(st_pe);
        if (st_have) {
            if (len == 0u || (len == 3u && ZA_EDIST(st_e) > (uint32_t)ZA_DP_WEAK_DIST)) { atomicAdd(&st_hist[st_e >> 24], 1u); st_U++; }
            else if (len + 1u != lp) st_NM++;
        }
        st_have = false;
    };
    __syncthreads();
    for (int base = 0; base < n; base += ZA_SEARCH_TILE) {
        // ---- (r06) FIRST this tile's own links in tables B and C are taken over from the registers they were fetched into a tile
        // ago -- before any new load goes out.  The compiler cannot count loads across the loop's back edge: its wait in front of
        // the first use of something loaded in the last iteration is for EVERY load in flight, and with the take-over behind the
        // next tile's link and byte loads (where it stood until r05) that wait covered loads issued a moment ago: all sixteen
        // waves sat through a trip to memory at the top of every tile.
        uint32_t lkb[ZA_SEARCH_TILE / ZA_SEARCH_THREADS], lkc[ZA_SEARCH_TILE / ZA_SEARCH_THREADS];
#pragma unroll
        for (int k = 0; k < ZA_SEARCH_TILE / ZA_SEARCH_THREADS; k++) {
            if (ZA_OWN_STAGE) {
                const uint32_t a = own_area(base) + 2u * (uint32_t)(k * ZA_SEARCH_THREADS + tid);
                lkb[k] = *(const uint16_t *)(lds + (a & 0xFFFFu));
                lkc[k] = USEC ? (uint32_t)*(const uint16_t *)(lds + ((a + 8192u) & 0xFFFFu)) : 0u;
            } else if (CONSEC) {
                lkb[k] = ((k < 2 ? nob.x : nob.y) >> (16 * (k & 1))) & 0xFFFFu;
                lkc[k] = ((k < 2 ? noc.x : noc.y) >> (16 * (k & 1))) & 0xFFFFu;
            } else { lkb[k] = nlkb[k]; lkc[k] = nlkc[k]; }
            asm volatile("" : "+v"(lkb[k]), "+v"(lkc[k]) : : "memory");       // (here and now: nothing is moved across)
        }
        if (ZA_OWN_STAGE) za_lds_barrier();                       // (every thread has its links: the area may be written at this tile's end)
        if constexpr (!FULL) flush_results();
        // ---- the NEXT tile's links and bytes are fetched into registers now and put into the rings after this tile's
        // search: the global-memory latency hides behind the search instead of stalling all 16 waves in front of it
        // (at most 4 links and 2 dwords per thread: one tile of each)
        int need_links = base + 2 * ZA_SEARCH_TILE;
        if (need_links > n) need_links = n;
        int need_bytes = base + 2 * ZA_SEARCH_TILE + ZA_LOOKAHEAD;
        if (need_bytes > n) need_bytes = n;
        need_bytes = (need_bytes + 3) & ~3;
        uint64_t nbv; uint32_t nsh;
        const int pq = links_loaded + 4 * tid;                  // my four links of the next tile
        const uint2 nq = load_quad(pq, need_links);
        // (r06) ONE branch-free form for every tile: a dword that reaches over the end of the caller's buffer (the last unit's last
        // one to three bytes) is loaded from as far in front as it reaches over, and shifted down -- zeros behind the buffer's end
        // as before.  In front of the dictionary nothing is ever fetched here (the loop stages from the second tile on).  Until r05
        // the edge had a careful path of its own, byte loads in branches, and at the join of the two paths the compiler waited for
        // every load in flight -- the link and byte loads issued a moment before: all sixteen waves sat through a trip to memory
        // at the top of every tile (a ninth of the kernel; the ablation without the own-link loads had shown it and blamed them).
        // A thread takes EIGHT consecutive bytes (one load instruction instead of two: vector-memory instructions are what this
        // kernel's waves queue for -- every one of them is worth about a percent of the tile).
        {
            const int rd_i = (int)(readable < (long long)0x7FFFFFF0 ? readable : (long long)0x7FFFFFF0);
            const int p = bytes_loaded + 8 * tid;
            int over = p + 8 - rd_i;
            over = over > 0 ? over : 0;                           // 0 .. 7 for every p < need_bytes (p < n <= readable)
            nbv = 0ull; nsh = 8u * (uint32_t)over;                // (the shift is applied where the bytes are stored: not here, where it would wait for the load)
            if (p < need_bytes) nbv = za_ld64(data + p - over);
        }
        if constexpr (FULL) {
        // ---- levels 7-9: ZA_WL_STEPS chain steps per visit, the unfinished positions through the wave's work list
        const int lane = tid & 63;
        uint32_t *wl = (uint32_t *)(lds + ZA_BYTES + 32 + 2 * ZA_RING) + (tid >> 6) * (2 * ZA_WL_ENTRIES);
        uint32_t wl_n = 0;                                                 // entries on my wave's list (the same in every lane)
        // one visit: up to ZA_WL_STEPS more steps of the walk of position P; true = not done yet
        auto visit = [&](uint32_t P, uint32_t me0, uint32_t me1, uint32_t me2, uint32_t me3, int maxlen, int cap, int nice,
                         uint32_t &q, uint32_t &qb, uint32_t &d, int &best_len, int &best_dist, int &depth) -> bool {
            bool alive = depth > 0;
            int steps = ZA_WL_STEPS;
#pragma unroll
            while (alive && steps-- > 0) {
                depth--;
                q -= d;
                qb -= 2u * d;
                qb += qb < RING_B0 ? RING_BYTES : 0u;
                const int dist = (int)(P - q);
                if (dist > L.max_dist) { alive = false; break; }            // (also the end of the chain: a link of 0xFFFF)
                d = *(const uint16_t *)(lds + qb);
                const uint32_t sh = q & 3u;
                const uint32_t *cw = (const uint32_t *)(lds + (q & (uint32_t)(ZA_BYTES - 4)));
                const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], c3 = cw[3], c4 = cw[4];
                uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, sh) ^ me0;
                uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, sh) ^ me1;
                uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, sh) ^ me2;
                uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, sh) ^ me3;
                uint32_t fbit, f1, f2, f3;
                asm("v_ffbl_b32 %0, %4\n\tv_ffbl_b32 %1, %5\n\tv_ffbl_b32 %2, %6\n\tv_ffbl_b32 %3, %7\n\t"
                    "v_add_u32_e64 %1, %1, 32 clamp\n\tv_add_u32_e64 %2, %2, 64 clamp\n\tv_add_u32_e64 %3, %3, %8 clamp\n\t"
                    "v_min3_u32 %0, %0, %1, %2\n\tv_min_u32_e32 %0, %0, %3"
                    : "=&v"(fbit), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(96u));
                int len = (int)min(fbit >> 3, 16u);
                if (len == 16 && cap > 16 && best_len < cap &&
                    (best_len < 16 || (uint8_t)za_lds_ld32(win32, q + (uint32_t)best_len) == (uint8_t)za_lds_ld32(win32, P + (uint32_t)best_len)))
                    len = za_search_extend(win32, q, P, len, maxlen);
                len = len < cap ? len : cap;
                const bool better = len > best_len;
                best_len = better ? len : best_len;
                best_dist = better ? dist : best_dist;
                alive = best_len < nice && depth > 0;
            }
            return alive;
        };
        auto my16 = [&](uint32_t P, uint32_t &me0, uint32_t &me1, uint32_t &me2, uint32_t &me3) {
            const uint32_t idx = P & (ZA_BYTES - 1), w = idx >> 2, sh = idx & 3u;
            const uint32_t m0 = win32[w], m1 = win32[w + 1], m2 = win32[w + 2], m3 = win32[w + 3], m4 = win32[w + 4];
            me0 = __builtin_amdgcn_alignbyte(m1, m0, sh); me1 = __builtin_amdgcn_alignbyte(m2, m1, sh);
            me2 = __builtin_amdgcn_alignbyte(m3, m2, sh); me3 = __builtin_amdgcn_alignbyte(m4, m3, sh);
        };
        // a walk is over: the position's own links in tables B and C (one candidate each, compared in full; skipped behind a nice
        // match), the drop rules, the entry
        auto finish = [&](int p, uint32_t P, uint32_t me0, uint32_t me1, uint32_t me2, uint32_t me3, int maxlen, int cap, int nice,
                          int best_len, int best_dist, uint32_t own_b, uint32_t own_c) {
#pragma unroll
            for (int t = 0; t < (USEC ? 2 : 1); t++) {
                const int dl = maxlen >= ZA_MIN_MATCH ? (int)(t == 0 ? own_b : own_c) : 0;
                if (dl != 0 && dl <= L.max_dist && best_len < nice) {
                    const uint32_t q = P - (uint32_t)dl;
                    const uint32_t sh = q & 3u;
                    const uint32_t *cw = (const uint32_t *)(lds + (q & (uint32_t)(ZA_BYTES - 4)));
                    const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], c3 = cw[3], c4 = cw[4];
                    const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, sh) ^ me0, x1 = __builtin_amdgcn_alignbyte(c2, c1, sh) ^ me1;
                    const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, sh) ^ me2, x3 = __builtin_amdgcn_alignbyte(c4, c3, sh) ^ me3;
                    uint32_t fbit, f1, f2, f3;
                    asm("v_ffbl_b32 %0, %4\n\tv_ffbl_b32 %1, %5\n\tv_ffbl_b32 %2, %6\n\tv_ffbl_b32 %3, %7\n\t"
                        "v_add_u32_e64 %1, %1, 32 clamp\n\tv_add_u32_e64 %2, %2, 64 clamp\n\tv_add_u32_e64 %3, %3, %8 clamp\n\t"
                        "v_min3_u32 %0, %0, %1, %2\n\tv_min_u32_e32 %0, %0, %3"
                        : "=&v"(fbit), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(96u));
                    int len = (int)min(fbit >> 3, 16u);
                    if (len == 16 && cap > 16) len = za_search_extend(win32, q, P, len, maxlen);     // (no quick reject: a tie with a nearer candidate counts)
                    len = len < cap ? len : cap;
                    const bool better = len > best_len || (len == best_len && dl < best_dist);
                    best_len = better ? len : best_len;
                    best_dist = better ? dl : best_dist;
                }
            }
            uint32_t result = 0;
            if (best_len >= ZA_MIN_MATCH && !(best_len == 3 && best_dist > L.too_far3) && !(best_len == 4 && best_dist > L.too_far4))
                result = ((uint32_t)best_len << 15) | (uint32_t)(best_dist - 1);
            longm = longm || best_len > 64;
            best[p] = __builtin_amdgcn_perm(me0, result, 0x04020100u);
        };
        // the unfinished walks of this visit go to the list: position in the tile | best length << 12 | steps left << 21, and
        // how far back the walk stands | best distance << 16
        auto push = [&](bool alive, int p, uint32_t P, uint32_t q, int best_len, int best_dist, int depth) {
            const unsigned long long m = __ballot(alive);
            if (alive) {
                const uint32_t at = wl_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                wl[2 * at] = (uint32_t)(p - base) | ((uint32_t)best_len << 12) | ((uint32_t)depth << 21);
                wl[2 * at + 1] = (P - q) | ((uint32_t)best_dist << 16);
            }
            wl_n += (uint32_t)__builtin_popcountll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // other lanes read these entries (one wave: only the compiler must keep the order)
            __builtin_amdgcn_wave_barrier();
        };
        // the last nb entries of the list (nb <= 64) get their next visit
        auto batch = [&](uint32_t nb) {
            wl_n -= nb;
            const bool has = (uint32_t)lane < nb;
            uint32_t e0 = 0, e1 = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (has) { e0 = wl[2 * (wl_n + lane)]; e1 = wl[2 * (wl_n + lane) + 1]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (the pushes below may land on the slots just read)
            __builtin_amdgcn_wave_barrier();
            bool alive = false;
            int p = 0, best_len = 0, best_dist = 0, depth = 0;
            uint32_t P = 0, q = 0;
            if (has) {
                p = base + (int)(e0 & 0xFFFu);
                best_len = (int)((e0 >> 12) & 0x1FFu); depth = (int)(e0 >> 21); best_dist = (int)(e1 >> 16);
                P = goff + (uint32_t)(ZA_WIN + p);
                q = P - (e1 & 0xFFFFu);
                int seg_end = ((p >> sshift) + 1) << sshift;
                if (seg_end > n) seg_end = n;
                int maxlen = seg_end - p;
                if (maxlen > ZA_MAX_MATCH) maxlen = ZA_MAX_MATCH;
                const int cap = L.cap < maxlen ? L.cap : maxlen, nice = L.nice < cap ? L.nice : cap;
                uint32_t me0, me1, me2, me3;
                my16(P, me0, me1, me2, me3);
                const uint32_t sq = q % ZA_RING;
                uint32_t qb = RING_B0 + 2u * sq;
                uint32_t d = *(const uint16_t *)(lds + qb);
                alive = visit(P, me0, me1, me2, me3, maxlen, cap, nice, q, qb, d, best_len, best_dist, depth);
                // (a position that comes off the list fetches its own links here: they travelled ahead only for the lane that started it)
                if (!alive) finish(p, P, me0, me1, me2, me3, maxlen, cap, nice, best_len, best_dist, (uint32_t)linkb[p], USEC ? (uint32_t)linkc[p] : 0u);
            }
            push(alive, p, P, q, best_len, best_dist, depth);
        };
        uint32_t slot = (goff + (uint32_t)(ZA_WIN + base + tid)) % ZA_RING;
        load_own_links(base + ZA_SEARCH_TILE);                                                        // (lkb / lkc: my positions' own links, fetched a tile ago)
#pragma unroll
        for (int k = 0; k < ZA_SEARCH_TILE / ZA_SEARCH_THREADS; k++, slot = slot + ZA_SEARCH_THREADS >= ZA_RING ? slot + ZA_SEARCH_THREADS - ZA_RING : slot + ZA_SEARCH_THREADS) {
            const int p = base + k * ZA_SEARCH_THREADS + tid;
            bool alive = false;
            const uint32_t P = goff + (uint32_t)(ZA_WIN + p);
            uint32_t q = P;
            int best_len = ZA_MIN_MATCH - 1, best_dist = 0, depth = L.chain;
            if (p < n) {
                int seg_end = ((p >> sshift) + 1) << sshift;
                if (seg_end > n) seg_end = n;
                int maxlen = seg_end - p;
                if (maxlen > ZA_MAX_MATCH) maxlen = ZA_MAX_MATCH;
                uint32_t me0, me1, me2, me3;
                my16(P, me0, me1, me2, me3);
                const int cap = L.cap < maxlen ? L.cap : maxlen, nice = L.nice < cap ? L.nice : cap;
                if (maxlen >= ZA_MIN_MATCH) {
                    uint32_t qb = RING_B0 + 2u * slot;
                    uint32_t d = *(const uint16_t *)(lds + qb);
                    alive = visit(P, me0, me1, me2, me3, maxlen, cap, nice, q, qb, d, best_len, best_dist, depth);
                }
                if (!alive) finish(p, P, me0, me1, me2, me3, maxlen, cap, nice, best_len, best_dist, lkb[k], lkc[k]);
            }
            push(alive, p, P, q, best_len, best_dist, depth);
            while (__builtin_amdgcn_readfirstlane((int)wl_n) >= 64) batch(64u);
        }
        for (;;) {                                                         // what is left of the tile's walks
            const uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)wl_n);
            if (left == 0) break;
            batch(left < 64u ? left : 64u);
        }
        } else {
        uint32_t slot = (goff + (uint32_t)(ZA_WIN + base + (CONSEC ? 4 * tid : tid))) % ZA_RING;      // ring slot of my (first) position, moved on by 1 024 per round (CONSEC: a multiple of four, like the ring's size -- my four slots do not wrap)
        // my four positions' own links in tables B and C are what was fetched a tile ago (lkb / lkc); the next tile's are asked for now
        load_own_links(base + ZA_SEARCH_TILE);
        // (the statistics' loads go out BEHIND the point where last tile's loads are used: the compiler's wait there, across the
        // loop's back edge, is for every load in flight -- issued at the top of the tile, these two made all sixteen waves wait a
        // trip to memory per tile: search 8.18 -> 8.60 ms per GiB, more than the separate kernel costs)
        if (stats && base > 0) stats_issue(base - ZA_SEARCH_TILE);
#ifndef ZA_SEARCH_KUNROLL
#define ZA_SEARCH_KUNROLL 4              // the four positions of a thread per tile as straight code (18.9 against 19.4 ms per 4 GiB at level 6)
#endif
        // (r06, ZA_SEARCH_CONSEC) A thread's four positions of a tile are CONSECUTIVE ones, 4 tid .. 4 tid + 3 (until now tid, tid +
        // 1 024, ...): what a position needs of its own -- its 16 bytes, its link in the ring, its links in tables B and C, its
        // segment's end -- comes in once for the four: five aligned dwords instead of twenty (a run's positions are multiples of
        // four where the tile's are: the four byte offsets are constants, the first needs no alignment at all), one 8-byte LDS read
        // of the ring instead of four 2-byte ones, one 8-byte load per table instead of four 2-byte ones, one 16-byte store of the
        // entries instead of four -- vector-memory instructions are what this kernel's waves queue for.
        const int p0c = base + 4 * tid;
        const uint32_t P0c = goff + (uint32_t)(ZA_WIN + p0c);
        uint32_t cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0, cm4 = 0;
        uint2 cdq = make_uint2(0u, 0u);
        int cseg_end = 0;
        if (CONSEC && p0c < n) {
            const uint32_t w = (P0c & (ZA_BYTES - 1)) >> 2;
            cm0 = win32[w]; cm1 = win32[w + 1]; cm2 = win32[w + 2]; cm3 = win32[w + 3]; cm4 = win32[w + 4];
            cdq = *(const uint2 *)(lds + RING_B0 + 2u * slot);
            cseg_end = ((p0c >> sshift) + 1) << sshift;                 // (segments are 32 bytes at least: the four share one)
            if (cseg_end > n) cseg_end = n;
        }
#pragma unroll
        for (int k = 0; k < ZA_SEARCH_TILE / ZA_SEARCH_THREADS; k++, slot = CONSEC ? slot : slot + ZA_SEARCH_THREADS >= ZA_RING ? slot + ZA_SEARCH_THREADS - ZA_RING : slot + ZA_SEARCH_THREADS) {
            const int p = CONSEC ? p0c + k : base + k * ZA_SEARCH_THREADS + tid;
            res[k] = 0u;
            if (p >= n) continue;
            int seg_end = cseg_end;
            if (!CONSEC) {
                seg_end = ((p >> sshift) + 1) << sshift;
                if (seg_end > n) seg_end = n;
            }
            int maxlen = seg_end - p;
            if (maxlen > ZA_MAX_MATCH) maxlen = ZA_MAX_MATCH;
            // the result: distance - 1 (bits 0..14) | length (bits 15..23); the position's own byte travels in the top byte
            uint32_t result = 0;
            const uint32_t P = goff + (uint32_t)(ZA_WIN + p);
            // my first 16 bytes stay in registers; every candidate's first 16 bytes are compared
            // against them without branches (this also plays the role of zlib's quick-reject byte)
            uint32_t me0, me1, me2, me3;
            if (CONSEC) {
                if (k == 0) { me0 = cm0; me1 = cm1; me2 = cm2; me3 = cm3; }
                else {
                    me0 = __builtin_amdgcn_alignbyte(cm1, cm0, (uint32_t)k); me1 = __builtin_amdgcn_alignbyte(cm2, cm1, (uint32_t)k);
                    me2 = __builtin_amdgcn_alignbyte(cm3, cm2, (uint32_t)k); me3 = __builtin_amdgcn_alignbyte(cm4, cm3, (uint32_t)k);
                }
            } else {   // five aligned dwords and one shift amount (the ring has mirrored pad dwords behind its end)
                const uint32_t idx = P & (ZA_BYTES - 1), w = idx >> 2, sh = idx & 3u;
                const uint32_t m0 = win32[w], m1 = win32[w + 1], m2 = win32[w + 2], m3 = win32[w + 3], m4 = win32[w + 4];
                me0 = __builtin_amdgcn_alignbyte(m1, m0, sh); me1 = __builtin_amdgcn_alignbyte(m2, m1, sh);
                me2 = __builtin_amdgcn_alignbyte(m3, m2, sh); me3 = __builtin_amdgcn_alignbyte(m4, m3, sh);
            }
            if (maxlen >= ZA_MIN_MATCH) {
                const int cap = L.cap < maxlen ? L.cap : maxlen;             // bytes compared per candidate
                const int nice = L.nice < cap ? L.nice : cap;
                int best_len = ZA_MIN_MATCH - 1, best_dist = 0;
                uint32_t q = P;
                uint32_t qb = RING_B0 + 2u * (CONSEC ? slot + (uint32_t)k : slot);   // LDS byte address of q's link, kept incrementally
                uint32_t d = CONSEC ? ((k < 2 ? cdq.x : cdq.y) >> (16 * (k & 1))) & 0xFFFFu : (uint32_t)*(const uint16_t *)(lds + qb);
                int depth = STEPS > 0 ? STEPS : L.chain;
#pragma unroll
                while (depth-- > 0) {
                    q -= d;
                    qb -= 2u * d;
                    qb += qb < RING_B0 ? RING_BYTES : 0u;
                    const int dist = (int)(P - q);
                    if (dist > L.max_dist) break;                            // (also the end of the chain: a link of 0xFFFF)
                    // one LDS round trip per chain step: next link + 5 aligned dwords of the candidate
                    d = *(const uint16_t *)(lds + qb);
                    const uint32_t sh = q & 3u;
                    const uint32_t *cw = (const uint32_t *)(lds + (q & (uint32_t)(ZA_BYTES - 4)));
                    const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], c3 = cw[3], c4 = cw[4];
                    uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, sh) ^ me0;
                    uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, sh) ^ me1;
                    uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, sh) ^ me2;
                    uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, sh) ^ me3;
                    // first differing bit of the 128: v_ffbl gives 0xFFFFFFFF for "none", which the saturating adds keep as
                    // "none"; 4 ffbl + 3 add + min3 + min.  ONE asm statement: hipcc pads every statement with an s_nop, and the
                    // statement also keeps all four compares unconditional (the compiler otherwise sinks the loads into nested
                    // branches, one LDS round trip per level, and turns each "none" into a compare + select)
                    uint32_t fbit, f1, f2, f3;
                    asm("v_ffbl_b32 %0, %4\n\tv_ffbl_b32 %1, %5\n\tv_ffbl_b32 %2, %6\n\tv_ffbl_b32 %3, %7\n\t"
                        "v_add_u32_e64 %1, %1, 32 clamp\n\tv_add_u32_e64 %2, %2, 64 clamp\n\tv_add_u32_e64 %3, %3, %8 clamp\n\t"
                        "v_min3_u32 %0, %0, %1, %2\n\tv_min_u32_e32 %0, %0, %3"
                        : "=&v"(fbit), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(96u));     // 96 is not an inline constant
                    int len = (int)min(fbit >> 3, FULL ? 16u : (uint32_t)cap);     // (cap <= 16 on the levels that compare 16 bytes)
                    if (FULL && len == 16 && cap > 16 && best_len < cap &&
                        (best_len < 16 || (uint8_t)za_lds_ld32(win32, q + (uint32_t)best_len) == (uint8_t)za_lds_ld32(win32, P + (uint32_t)best_len))) {
                        // levels that compare in full: at least 16 equal bytes -- finish the compare the long way (a candidate that
                        // differs at the byte behind the best length so far cannot beat it: zlib's quick reject, one LDS read)
                        len = za_search_extend(win32, q, P, len, maxlen);       // (8 bytes per round; 4 per round until round 4)
                    }
                    if (FULL) len = len < cap ? len : cap;
                    const bool better = len > best_len;
                    best_len = better ? len : best_len;
                    best_dist = better ? dist : best_dist;
                    if (best_len >= nice) break;
                }
                // tables B and C (r06): a link counts only if the candidate really shares the table's context -- its first 3 (B) or
                // 12 (C) bytes, an equality test on one / three dwords instead of the 16-byte compare with its first-difference
                // arithmetic -- and then stands for a match of exactly that length (longer wins, nearer wins ties); a winner that
                // came from B or C is extended to its true length below, like a winner at `cap`.  (No link: the position against
                // itself, masked out.)  C is left out where fewer than 12 bytes remain.
                bool from_bc = false;
#ifndef ZA_ABL_SEARCH_NO_B
                {
                    const uint32_t dl = lkb[k];
#ifdef ZA_ABL_BC_SELF
                    const uint32_t q2 = P - (dl & 3u);             // (timing only: the candidates' LDS reads at the positions' own, consecutive addresses)
#else
                    const uint32_t q2 = P - dl;
#endif
                    const uint32_t *cw = (const uint32_t *)(lds + (q2 & (uint32_t)(ZA_BYTES - 4)));
                    const uint32_t c0 = cw[0], c1 = cw[1];
                    const uint32_t x = (__builtin_amdgcn_alignbyte(c1, c0, q2 & 3u) ^ me0) << 8;        // 0: the first three bytes are equal
                    const bool better = x == 0u && dl != 0u && (int)dl <= L.max_dist &&
                                        (ZA_HASH_BYTES_B > best_len || (ZA_HASH_BYTES_B == best_len && (int)dl < best_dist));
                    best_len = better ? ZA_HASH_BYTES_B : best_len;
                    best_dist = better ? (int)dl : best_dist;
                    from_bc = better;
                }
#endif
#ifndef ZA_ABL_SEARCH_NO_C
                if (USEC) {
                    const uint32_t dl = lkc[k];
#ifdef ZA_ABL_BC_SELF
                    const uint32_t q2 = P - (dl & 3u);
#else
                    const uint32_t q2 = P - dl;
#endif
                    const uint32_t sh = q2 & 3u;
                    const uint32_t *cw = (const uint32_t *)(lds + (q2 & (uint32_t)(ZA_BYTES - 4)));
                    const uint32_t c0 = cw[0], c1 = cw[1], c2 = cw[2], c3 = cw[3];
                    const uint32_t x = (__builtin_amdgcn_alignbyte(c1, c0, sh) ^ me0) | (__builtin_amdgcn_alignbyte(c2, c1, sh) ^ me1) |
                                       (__builtin_amdgcn_alignbyte(c3, c2, sh) ^ me2);                 // 0: the first twelve bytes are equal
                    const bool better = x == 0u && dl != 0u && (int)dl <= L.max_dist && cap >= ZA_HASH_BYTES_C &&
                                        (ZA_HASH_BYTES_C > best_len || (ZA_HASH_BYTES_C == best_len && (int)dl < best_dist));
                    best_len = better ? ZA_HASH_BYTES_C : best_len;
                    best_dist = better ? (int)dl : best_dist;
                    from_bc = from_bc || better;
                }
#endif
#ifndef ZA_ABL_NO_EXTEND
                if (!FULL && (best_len == cap || from_bc) && best_len < maxlen) {
                    // the winner of a 16-byte comparison, or of a context test: its true length (once per position, not per candidate)
                    best_len = za_search_extend(win32, P - (uint32_t)best_dist, P, best_len, maxlen);
                }
#endif
                if (best_len >= ZA_MIN_MATCH && !(best_len == 3 && best_dist > L.too_far3) && !(best_len == 4 && best_dist > L.too_far4))
                    result = ((uint32_t)best_len << 15) | (uint32_t)(best_dist - 1);
                longm = longm || best_len > 64;
            }
#ifdef ZA_ABL_SEARCH_NOLIT
            res[k] = result;                                                  // (timing only: the parse then sees zero bytes)
#else
            res[k] = __builtin_amdgcn_perm(me0, result, 0x04020100u);        // byte 3 = my byte (byte 0 of me0), bytes 0..2 = result
#endif
        }
        res_base = base;
        }
        // ---- put the next tile into the rings.  No barrier is needed in front of these stores: they land at least
        // 65536-4096-272-32768 byte slots / 40960-4096-32768 link slots behind any walk of this tile that is still
        // running in another wave.  The barrier behind them makes the next tile visible.
        if (pq < need_links) store_quad(pq, need_links, nq);
        {
            const int p = bytes_loaded + 8 * tid;
            if (p < need_bytes) {
                const uint64_t v = nbv >> nsh;
                store_bytes(p, (uint32_t)v);
                if (p + 4 < need_bytes) store_bytes(p + 4, (uint32_t)(v >> 32));
            }
        }
        if (ZA_OWN_STAGE) own_store(base + ZA_SEARCH_TILE);
        links_loaded = need_links; bytes_loaded = need_bytes;
        // (r06) the tile's barrier orders the rings -- LDS -- and nothing else: the plain __syncthreads() also waited for this wave's
        // result stores, the last of them issued a moment ago (the read-back of the folded statistics is the one thing that needs it)
        if (stats) { flush_results(); stats_use(); __syncthreads(); }      // (the read-back needs the results in memory behind this barrier)
        else za_lds_barrier();
    }
    if constexpr (!FULL) flush_results();
    if (stats && n > 0) {
        // the unit's last tile, then the cost table: 256 literal costs and the match base cost (what za_k_dpstats did)
        __syncthreads();
        stats_issue((n - 1) & ~(ZA_SEARCH_TILE - 1));
        stats_use();
        for (int d = 32; d >= 1; d >>= 1) { st_U += (uint32_t)__shfl_xor((int)st_U, d, 64); st_NM += (uint32_t)__shfl_xor((int)st_NM, d, 64); }
        if ((tid & 63) == 0) { atomicAdd(&st_cnt[0], st_U); atomicAdd(&st_cnt[1], st_NM); }
        __syncthreads();
        const uint32_t U = st_cnt[0], NM = st_cnt[1];
        uint32_t hh = 0u;
        if (tid < 256) {
            hh = 16u * st_hist[tid] + 1u + (U >> 6);
            st_hist[tid] = 0u;                                        // (ready for the run's next unit)
            uint32_t T = hh;
            for (int d = 32; d >= 1; d >>= 1) T += (uint32_t)__shfl_xor((int)T, d, 64);
            if ((tid & 63) == 0) atomicAdd(&st_cnt[2], T);
        }
        __syncthreads();
        if (tid < 256) {
            const uint32_t T = st_cnt[2];
            uint32_t *cost = cost_ws + (size_t)ui * ZA_DP_COSTS;
            int lbias = za_ilog4(U + NM, U ? U : 1u), mbias = za_ilog4(U + NM, NM ? NM : 1u);
            lbias = lbias > 24 ? 24 : lbias;
            mbias = mbias > 24 ? 24 : mbias;
            const int cl = za_ilog4(T, hh) + lbias;
            cost[tid] = (uint32_t)(cl < 12 ? 12 : cl > 52 ? 52 : cl);
            if (tid == 0) { cost[256] = (uint32_t)(12 + mbias + 20); cost[257] = 0u; }
        }
        __syncthreads();
        if (tid < 4) st_cnt[tid] = 0u;                                // (the next unit's staging barrier orders this)
        st_U = 0u; st_NM = 0u;
    }
    if (L.dp && __ballot(longm) != 0ull && (tid & 63) == 0) cost_ws[(size_t)ui * ZA_DP_COSTS + 258] = 1u;     // (every writer writes the same)
    carry_bytes = bytes_loaded; n_prev = n;
    }
}

// ------------------------------------------------------------------------------------------------
// k_dpstats + k_optparse : stage 3a, the dynamic programme of levels 4-9
// ------------------------------------------------------------------------------------------------
// Going BACKWARDS through a 2 KiB segment the programme works out, for every position p, the cheapest way to code everything
// from p to the segment's end: acc[p] = min(literal + acc[p + 1], match of l bytes + acc[p + l]) over the position's match and
// its ZA_DP_SUB next shorter lengths, in estimated quarter bits.  The choice is written back INTO the `best` entry (no match /
// the chosen length), so that the greedy walk of the parse kernel is the programme's parse.
//
// za_k_dpstats (one 256-thread workgroup per unit): the unit's cost table (oracle dp_costs) from its entries
// alone -- a coalesced pass over a QUARTER of the entries (every fourth block of 256 positions: the same parse within 0.03 % of
// size, a quarter of the 4 N bytes): bytes of the literal-like positions (LDS atomics), their count U, the count NM of positions
// where a new match starts; then 256 literal costs and one match base cost, integer logarithms in quarter bits.
//
// za_k_optparse (one wave per unit, one lane per segment, like the parse kernel it feeds):
//   acc[]      a step reads acc[p + l] for l up to 258, but lengths above 64 are rare: the last ZA_DP_NEAR = 64 values live in
//              an LDS ring (two 16-bit slots per dword, slot-major -- dword row (slot >> 1), column lane: a lane only ever touches
//              its own bank; costs are kept modulo 2^16, two entries at most 258 positions apart differ by less than 2^15, and
//              are compared through signed differences; four rows mirrored behind the end so that a step's five values never
//              wrap), and EVERY value also goes to a per-segment array in global memory (the token workspace, not in use yet) at
//              the end of its chunk, from where the rare long match fetches its five (behind a device-scope fence: the only
//              traffic between lanes through memory; it is at least 61 positions = 3 chunks old).  9.5 KiB of LDS instead of
//              the 35 KiB a full ring takes: ten waves per CU instead of three, and this kernel is all latency.
//   entries    staged through LDS rows in chunks of 16 positions, loaded and stored TRANSPOSED (four lanes move the four 16-byte
//              pieces of one segment's 64-byte row), last chunk first.
#define ZA_PCH 32                     // positions of a chunk of entries in the parse kernel
#define ZA_PROW (ZA_PCH + 1)          // dwords of a lane's LDS row there: the carried entry + the chunk; an odd stride
#ifndef ZA_DP_NEAR
#define ZA_DP_NEAR   64               // acc[p + 1 .. p + 64] come from the LDS ring
#endif
#define ZA_DP_ROWS   (ZA_DP_NEAR / 2 + 4)
#ifndef ZA_DCH
#define ZA_DCH       16               // positions per chunk (16 or 32)
#endif
#define ZA_DROW      (ZA_DCH + 1)
#define ZA_DPIECES   (ZA_DCH / 4)     // 16-byte pieces of a segment's row of one chunk = lanes per segment in the transposed moves
#define ZA_DP_SEGSLOTS 2064           // u16 slots per segment in the global array: 2 049 used, 16-byte multiples

__global__ __launch_bounds__(256) void za_k_dpstats(const ZaUnit *__restrict__ units, const uint32_t *__restrict__ best_ws,
                                                    uint32_t *__restrict__ cost_ws /* ZA_DP_COSTS per unit */)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t cnt[3], tsum;
    const int n = (int)units[blockIdx.x].in_len;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const uint32_t *best = best_ws + (size_t)blockIdx.x * ZA_BEST_STRIDE;
    hist[tid] = 0;
    if (tid < 3) cnt[tid] = 0;
    if (tid == 0) tsum = 0;
    __syncthreads();
    uint32_t U = 0, NM = 0;
    // a SAMPLE of the unit: every fourth block of 256 positions (the positions p with (p >> 8) & 3 == 0), one block per wave and round
    for (int base = 1024 * (tid >> 6); base < n; base += 4096) {
        const int i0 = base + 4 * lane;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        uint32_t pv = 0;
        if (i0 < n) v = *(const uint4 *)(best + i0);               // (rows are 16-byte aligned; what lies behind n is not looked at)
        if (lane == 0 && i0 > 0 && i0 < n) pv = best[i0 - 1];      // the entry in front of the block
        const uint32_t e[4] = {v.x, v.y, v.z, v.w};
        uint32_t lp = (uint32_t)__shfl_up((int)ZA_ELEN(v.w), 1, 64);
        if (lane == 0) lp = ZA_ELEN(pv);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t len = ZA_ELEN(e[k]);
            if (i0 + k < n) {
                if (len == 0u || (len == 3u && ZA_EDIST(e[k]) > (uint32_t)ZA_DP_WEAK_DIST)) { atomicAdd(&hist[e[k] >> 24], 1u); U++; }
                else if (len + 1u != lp) NM++;
            }
            lp = len;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { U += (uint32_t)__shfl_xor((int)U, d, 64); NM += (uint32_t)__shfl_xor((int)NM, d, 64); }
    if (lane == 0) { atomicAdd(&cnt[0], U); atomicAdd(&cnt[1], NM); }
    __syncthreads();
    U = cnt[0]; NM = cnt[1];
    const uint32_t hh = 16u * hist[tid] + 1u + (U >> 6);
    uint32_t T = hh;
    for (int d = 32; d >= 1; d >>= 1) T += (uint32_t)__shfl_xor((int)T, d, 64);
    if (lane == 0) atomicAdd(&tsum, T);
    __syncthreads();
    T = tsum;
    uint32_t *cost = cost_ws + (size_t)blockIdx.x * ZA_DP_COSTS;
    if (n == 0) return;
    int lbias = za_ilog4(U + NM, U ? U : 1u), mbias = za_ilog4(U + NM, NM ? NM : 1u);
    lbias = lbias > 24 ? 24 : lbias;
    mbias = mbias > 24 ? 24 : mbias;
    const int c = za_ilog4(T, hh) + lbias;
    cost[tid] = (uint32_t)(c < 12 ? 12 : c > 52 ? 52 : c);
    if (tid == 0) { cost[256] = (uint32_t)(12 + mbias + 20); cost[257] = 0u; }       // ([258]: the search's "has a long match" word)
}

__device__ __forceinline__ uint32_t za_pk_add_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The step is written for a wave that is alone on its SIMD most of the time (ten waves per CU): no branch in the common path
// (a position without a match reads a table row of "impossible" candidates), the five candidates of a match as packed 16-bit
// sums -- ring values (two slots per dword as they lie), the length table's row (extra bits of the five lengths, 0x3FFF where
// the length is not tried) and the base (distance cost - acc[p + 1]) -- and ONE minimum over (cost << 3 | 4 - k): the cheapest,
// among equals the longest.  The next position's entry, literal cost, table row and ring dwords are fetched before this
// position's sums (they do not depend on it: the nearest slot a step reads was written two steps ago).
__global__ __launch_bounds__(64) void za_k_optparse(const ZaUnit *__restrict__ units, uint32_t *__restrict__ best_ws,
                                                    const uint32_t *__restrict__ cost_ws, uint32_t *__restrict__ tok_ws /* scratch: acc[] */,
                                                    ZaLevel L)
{
    __shared__ uint32_t costt[ZA_DP_COSTS + 4];
    __shared__ uint32_t xt[3 * (ZA_MAX_MATCH + 1)];               // row `len`: 4 x extra bits of the lengths l0 .. l0 + 4 (u16 each; 0x3FFF: not tried), l0 in the last half
    __shared__ uint32_t ring[ZA_DP_ROWS * 64];                    // row r, column lane: slots 2 r (low half) and 2 r + 1
    __shared__ uint32_t rowb[64 * ZA_DROW];
    // (r06) The kernel's 17.3 KiB let a CU hold NINE of these one-wave workgroups: three on one SIMD, two on each of the others,
    // and the three share an issue port that two already fill -- eight (two per SIMD) run 6.5 % faster than nine (occupancy sweep:
    // 4 / 5 / 7 / 8 / 9 per CU: 5.33 / 3.76 / 3.34 / 2.89 / 3.09 ms per GiB; twelve, with a ring of 32 slots: within 1 % of eight).
    // So the workgroup asks for 20 KiB: ZA_DP_PAD bytes that nobody uses.
#ifndef ZA_DP_PAD
#define ZA_DP_PAD 2304
#endif
#if ZA_DP_PAD > 0
    __shared__ uint32_t dp_pad[ZA_DP_PAD / 4];
    if (units[blockIdx.x].in_len == 0xFFFFFFFFu) {               // (never: but the compiler cannot know, and must keep the array)
        dp_pad[threadIdx.x] = blockIdx.x; __syncthreads(); best_ws[threadIdx.x] = dp_pad[(threadIdx.x * 7u + best_ws[0]) % (ZA_DP_PAD / 4)];
    }
#endif
    const ZaUnit u = units[blockIdx.x];
    const int n = (int)u.in_len;
    const int lane = za_lane();
    const int sshift = ZA_UNIT_SEG_SHIFT(u.flags), seg = 1 << sshift;      // the unit's segment size: 32 .. 2 048
    const int nseg = (n + seg - 1) >> sshift;
    uint32_t *best = best_ws + (size_t)blockIdx.x * ZA_BEST_STRIDE;
    if (n == 0) return;
    for (int i = lane; i < ZA_DP_COSTS; i += 64) costt[i] = cost_ws[(size_t)blockIdx.x * ZA_DP_COSTS + i];
    for (int len = lane; len <= ZA_MAX_MATCH; len += 64) {
        const int l0 = len > 7 ? len - ZA_DP_SUB : 3;
        uint32_t x[ZA_DP_SUB + 1];
#pragma unroll
        for (int k = 0; k <= ZA_DP_SUB; k++) {
            const int l = l0 + k;
            int lc, ln = 0, le;
            if (l <= ZA_MAX_MATCH) za_len_sym(l, lc, ln, le);
            x[k] = (len >= 3 && l <= len) ? (uint32_t)(4 * ln) : 0x3FFFu;
        }
        xt[3 * len] = x[0] | (x[1] << 16); xt[3 * len + 1] = x[2] | (x[3] << 16); xt[3 * len + 2] = x[4] | ((uint32_t)l0 << 16);
    }
    // the ring starts as zeros: a step reads five slots whatever its match is, and the ones it does not try (0x3FFF in the table)
    // must hold something near acc[p + 1] for the packed sums to stay inside 16 signed bits -- real values, or, behind the segment's
    // end, acc[s1] = 0
    for (int i = lane; i < ZA_DP_ROWS * 64; i += 64) ring[i] = 0;
    __syncthreads();

    const int s0 = lane << sshift;
    int s1 = s0 + seg;
    if (s1 > n) s1 = n;
    const bool active = lane < nseg;
    uint32_t *myb = rowb + lane * ZA_DROW;
    uint8_t *myring = (uint8_t *)(ring + lane);                    // byte address of slot s: (s >> 1) * 256 + (s & 1) * 2
    uint16_t *accg = (uint16_t *)(tok_ws + (size_t)blockIdx.x * ZA_TOK_STRIDE) + (size_t)lane * ZA_DP_SEGSLOTS;     // my segment's acc[], index p - s0
    auto ring_put = [&](int slot, uint32_t v) {
        *(uint16_t *)(myring + (slot >> 1) * 256 + (slot & 1) * 2) = (uint16_t)v;
        if (slot < 8) *(uint16_t *)(myring + ((slot >> 1) + ZA_DP_NEAR / 2) * 256 + (slot & 1) * 2) = (uint16_t)v;
    };
    const int mbase = (int)costt[256];
#ifdef ZA_ABL_DP_NOLONG
    const bool has_long = false;                                   // (timing only: the ring's size against the waves a CU holds, wrong for matches longer than the ring)
#else
    const bool has_long = costt[258] != 0u;                        // (uniform, from the search) some match of the unit is longer than the ring: acc[] goes to memory too (a quarter of this kernel's time where it must)
#endif
    if (active) { ring_put((s1 - s0) & (ZA_DP_NEAR - 1), 0u); accg[s1 - s0] = 0; }     // acc[s1] = 0
    int acc_next = 0;                                              // acc[p + 1], the whole number (at most 2 048 x 52)
    // what the last long match fetched from memory: a window of sixteen values of acc[] (eight dwords), and where it starts
    int far_base = -100;
    uint32_t fw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint4 pb[ZA_DPIECES];
    auto prefetch = [&](int c) {
#pragma unroll
        for (int j = 0; j < ZA_DPIECES; j++) {
            const int sg = (64 / ZA_DPIECES) * j + lane / ZA_DPIECES, off = (sg << sshift) + c * ZA_DCH + 4 * (lane % ZA_DPIECES);
            pb[j] = make_uint4(0, 0, 0, 0);
            if (c >= 0 && sg < nseg && off < n) pb[j] = *(const uint4 *)(best + off);
        }
    };
    // stage B of a step: what needs the position's entry and nothing of the steps in front of it -- the literal's cost, the
    // length table's row, the ring's three dwords from acc[p + l0] on.  (idx = p - s0 is the same number in every lane.)
    struct Fetch { uint32_t e, clit, x0, x1, x2, w0, w1, w2; };
    auto stage_b = [&](uint32_t e, int idx_p) -> Fetch {
        Fetch f;
        f.e = e;
        f.clit = costt[e >> 24];
        const uint32_t len = ZA_ELEN(e);
        const uint32_t *xr = xt + 3u * len;
        f.x0 = xr[0]; f.x1 = xr[1]; f.x2 = xr[2];
        const uint32_t l0 = len > 7u ? len - (uint32_t)ZA_DP_SUB : 3u;
        const uint32_t st = ((uint32_t)idx_p + l0) & (uint32_t)(ZA_DP_NEAR - 1);      // slot of acc[p + l0]
        const uint8_t *rp = myring + (st >> 1) * 256;
        f.w0 = *(const uint32_t *)rp; f.w1 = *(const uint32_t *)(rp + 256); f.w2 = *(const uint32_t *)(rp + 512);
        return f;
    };
    const int nch = ((n < seg ? n : seg) + ZA_DCH - 1) / ZA_DCH;            // chunks of the longest segment
    uint4 po[ZA_DPIECES];                                           // the last chunk's entries on their way out
    uint32_t wacc[ZA_DCH / 2];                                      // ... and its acc values
    int pend = -1;
    auto flush_pending = [&](auto longs_tag) {
        if (pend < 0) return;
        const int pcb = s0 + pend * ZA_DCH;
        if (decltype(longs_tag)::value && active && pcb < s1) {
            uint4 *g = (uint4 *)(accg + pend * ZA_DCH);             // (segment arrays are 16-byte multiples apart)
#pragma unroll
            for (int k = 0; k < ZA_DCH / 8; k++) g[k] = make_uint4(wacc[4 * k], wacc[4 * k + 1], wacc[4 * k + 2], wacc[4 * k + 3]);
        }
#pragma unroll
        for (int j = 0; j < ZA_DPIECES; j++) {
            const int sg = (64 / ZA_DPIECES) * j + lane / ZA_DPIECES, off = (sg << sshift) + pend * ZA_DCH + 4 * (lane % ZA_DPIECES);
            if (sg < nseg && off < n) *(uint4 *)(best + off) = po[j];
        }
    };
    prefetch(nch - 1);
    // The walk comes in two instantiations, chosen per unit (uniform): a unit whose search saw no match longer than the ring -- all
    // of the bench's text -- never looks for one (no ballot per step, no window registers, no acc[] on its way to memory); the
    // other form is the general one.
    auto walk = [&](auto longs_tag) {
    constexpr bool LONGS = decltype(longs_tag)::value;
#pragma unroll 1
    for (int c = nch - 1; c >= 0; c--) {
        const int cb = s0 + c * ZA_DCH;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < ZA_DPIECES; j++) {
            uint32_t *r = rowb + ((64 / ZA_DPIECES) * j + lane / ZA_DPIECES) * ZA_DROW + 4 * (lane % ZA_DPIECES);
            r[0] = pb[j].x; r[1] = pb[j].y; r[2] = pb[j].z; r[3] = pb[j].w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        flush_pending(longs_tag);
        prefetch(c - 1);
        if (__ballot(active && cb < s1) != 0ull) {
            // Every lane walks the chunk's 16 positions in step (a segment's last chunk may be short: its positions at and behind s1
            // are walked without effect), as a pipeline of two stages: while position j is decided, stage B of position j - 1
            // and the entry of position j - 2 are on their way from the LDS.
            const int ib = c * ZA_DCH;                                // p - s0 of the chunk's first position, in every lane
            uint32_t e_a = myb[ZA_DCH - 2];                            // entry of position j - 1 (stage A: read one step ahead of stage B)
            Fetch f = stage_b(myb[ZA_DCH - 1], ib + ZA_DCH - 1);
#pragma unroll 4
            for (int j = ZA_DCH - 1; j >= 0; j--) {
                const int idx = ib + j;                                // = p - s0
                const bool live = active && cb + j < s1;
                const Fetch g = f;
                if (j >= 1) f = stage_b(e_a, idx - 1);
                if (j >= 2) e_a = myb[j - 2];
                const uint32_t e = g.e, len = ZA_ELEN(e), dm1 = e & 0x7FFFu, l0 = g.x2 >> 16;
                const uint32_t st1 = ((uint32_t)idx + l0) & 1u;
                uint32_t v0 = __builtin_amdgcn_alignbyte(g.w1, g.w0, 2u * st1), v1 = __builtin_amdgcn_alignbyte(g.w2, g.w1, 2u * st1);
                uint32_t v2 = st1 ? g.w2 >> 16 : g.w2;
                if (LONGS && __builtin_expect(__ballot(live && len > (uint32_t)ZA_DP_NEAR) != 0ull, 0)) {
                    // a long match: its five values acc[e - 4 .. e], e = p + len, from memory -- written at least three chunks ago
                    // by this wave (other lanes' stores among them: a fence, and loads that go to the device's L2).  A WINDOW of 16
                    // values is fetched, [fbase, fbase + 16) with e near its top: the positions inside one long match all end at the
                    // same place, and in a run longer than 258 (zeros) the end moves down by one per step, so a fetch serves the
                    // next eleven steps at least (one fetch per step made such units nine times slower)
                    if (live && len > (uint32_t)ZA_DP_NEAR) {
                        const int e_lo = idx + (int)len - ZA_DP_SUB;               // index of the first of the five
                        if (e_lo < far_base || e_lo + ZA_DP_SUB > far_base + 15) {
                            far_base = (e_lo + ZA_DP_SUB - 14) & ~1;               // even: the window is eight aligned dwords
                            // this wave's own stores of rounds ago must have reached the L2 (a wait for its vector-memory
                            // counter: a workgroup-scope fence; a device-scope one would write the whole L2 back first), and the
                            // loads must come from there, not from a line the L1 fetched before those stores (device-scope loads)
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                            const uint32_t *g32 = (const uint32_t *)(accg + far_base);
#pragma unroll
                            for (int k = 0; k < 8; k++) fw[k] = __hip_atomic_load(g32 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        const uint32_t o = (uint32_t)(e_lo - far_base), q = o >> 1;   // 0 .. 11; dwords q, q + 1, q + 2 of the window
                        uint32_t w0 = fw[0], w1 = fw[1], w2 = fw[2];
#pragma unroll
                        for (uint32_t k = 1; k <= 5; k++) { w0 = q == k ? fw[k] : w0; w1 = q == k ? fw[k + 1] : w1; w2 = q == k ? fw[k + 2] : w2; }
                        const uint32_t sh = (o & 1u) * 2u;
                        v0 = __builtin_amdgcn_alignbyte(w1, w0, sh); v1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
                        v2 = (o & 1u) ? w2 >> 16 : w2;
                    }
                }
                int de = 30 - (int)__builtin_clz(dm1 | 1u);        // extra bits of the distance: 0 for distances 1 .. 4
                de = de < 0 ? 0 : de;
                const uint32_t b16 = (uint32_t)(mbase + 4 * de - acc_next) & 0xFFFFu;
                const uint32_t bb = b16 | (b16 << 16);
                uint32_t sA = za_pk_add_u16(za_pk_add_u16(v0, g.x0), bb);
                const uint32_t sB = za_pk_add_u16(za_pk_add_u16(v1, g.x1), bb);
                const uint32_t sC = za_pk_add_u16(za_pk_add_u16(v2, g.x2), bb);
                if (l0 == 3u && dm1 >= (uint32_t)L.too_far3) sA = (sA & 0xFFFF0000u) | 0x3FFFu;      // a 3-byte match that far back is no candidate
                // cost << 3 | (4 - k): the minimum is the cheapest, and the longest among equals
#ifdef ZA_DP_TAGS_SHIFT
                const int t0 = (__builtin_amdgcn_sbfe((int)sA, 0, 16) << 3) | 4, t1 = (((int)sA >> 16) << 3) | 3;
                const int t2 = (__builtin_amdgcn_sbfe((int)sB, 0, 16) << 3) | 2, t3 = (((int)sB >> 16) << 3) | 1;
                const int t4 = __builtin_amdgcn_sbfe((int)sC, 0, 16) << 3;
#else
                // (r06) ONE instruction per candidate: the signed 16-bit half times 8 plus the tag (v_mad_i32_i16, the upper half by op_sel)
                // instead of a sign extension and a shift-or each
                int t0, t1, t2, t3, t4;
                asm("v_mad_i32_i16 %0, %5, 8, 4\n\tv_mad_i32_i16 %1, %5, 8, 3 op_sel:[1,0,0,0]\n\t"
                    "v_mad_i32_i16 %2, %6, 8, 2\n\tv_mad_i32_i16 %3, %6, 8, 1 op_sel:[1,0,0,0]\n\tv_mad_i32_i16 %4, %7, 8, 0"
                    : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4) : "v"(sA), "v"(sB), "v"(sC));
#endif
                int t = t0 < t1 ? t0 : t1;
                t = t < t2 ? t : t2;
                int tb = t3 < t4 ? t3 : t4;
                t = t < tb ? t : tb;
                const int mc = t >> 3;
                const bool ok = mc < (int)g.clit;
                const int c_best = ok ? mc : (int)g.clit;
                const uint32_t choice = ok ? l0 + 4u - ((uint32_t)t & 7u) : 0u;
                if (live) {
                    acc_next += c_best;
                    ring_put(idx & (ZA_DP_NEAR - 1), (uint32_t)acc_next);
                    myb[j] = (e & (ok ? 0xFF007FFFu : 0xFF000000u)) | (choice << 15);     // (choice is 0 where the literal won)
                }
            }
        }
        // ---- the chunk's entries (and, where the unit has long matches, its acc values: the ring's slots of this chunk, 16 slots =
        // 8 rows) are taken into registers now and STORED at the start of the next round, in front of that round's loads: the wait
        // for those loads at the top of the loop is then a wait for the youngest operations in flight, and the stores -- a whole
        // round old by then -- cost nothing (stored here, behind the loads, every round waited for its stores' round trip).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (LONGS && active && cb < s1) {
            const uint8_t *rp = myring + ((c * ZA_DCH) & (ZA_DP_NEAR - 1)) / 2 * 256;
#pragma unroll
            for (int k = 0; k < ZA_DCH / 2; k++) wacc[k] = *(const uint32_t *)(rp + 256 * k);
        }
#pragma unroll
        for (int j = 0; j < ZA_DPIECES; j++) {
            const uint32_t *r = rowb + ((64 / ZA_DPIECES) * j + lane / ZA_DPIECES) * ZA_DROW + 4 * (lane % ZA_DPIECES);
            po[j] = make_uint4(r[0], r[1], r[2], r[3]);
        }
        pend = c;
    }
    flush_pending(longs_tag);
    };
    if (has_long) walk(std::true_type{}); else walk(std::false_type{});
}

// ------------------------------------------------------------------------------------------------
// k_parse  (+ histogram + CRC-32)
// ------------------------------------------------------------------------------------------------
// One wave per unit, one lane per 2 KiB segment.  Each lane walks its own stream of `best` entries -- which carry the
// positions' own bytes (top byte of an entry), so the input is not read here at all; a dependent global load per token would cost
// microseconds, so the stream is staged through LDS in chunks of 32 positions per lane (rows with an odd dword stride:
// conflict free), with the next chunk's global loads in flight while the current one is parsed.
// The loads are TRANSPOSED: a lane does not fetch its own row (64 lanes x 16 bytes in 64 different cache lines per
// instruction = 64 memory requests, and the kernel was bound by the request rate of L2, not by bytes or instructions) --
// eight lanes fetch the eight 16-byte pieces of one segment's 128-byte row, so an instruction touches eight whole lines,
// and the pieces are written to the owning lane's LDS row.  A row keeps the previous chunk's last entry in front of the
// chunk (slot 0): the token at a chunk's last position looks at best[p + 1] (literals share token words) and is decided one chunk later.


__global__ __launch_bounds__(64) void za_k_parse(const ZaUnit *__restrict__ units,
                                                 const uint32_t *__restrict__ best_ws, uint32_t *__restrict__ tok_ws,
                                                 uint32_t *__restrict__ segtok_ws, uint32_t *__restrict__ hist_ws,
                                                 uint32_t *__restrict__ crc_out,
                                                 const uint32_t *__restrict__ crc_table,   // [256]
                                                 const uint32_t *__restrict__ x8k_table,   // [64] x^(8*2048*k)
                                                 ZaLevel L)
{
    __shared__ uint32_t hist[ZA_HIST_STRIDE];
    __shared__ uint32_t crct[256];
    __shared__ uint32_t rowb[64 * ZA_PROW + 3];      // (+3: the look-ahead reads at a segment's last positions)
    const ZaUnit u = units[blockIdx.x];
    const int n = (int)u.in_len;
    const int lane = za_lane();
    const int sshift = ZA_UNIT_SEG_SHIFT(u.flags), seg = 1 << sshift;      // the unit's segment size: 32 .. 2 048
    const int nseg = (n + seg - 1) >> sshift;
    for (int i = lane; i < ZA_HIST_STRIDE; i += 64) hist[i] = 0;
    for (int i = lane; i < 256; i += 64) crct[i] = crc_table[i];
    __syncthreads();

    const int s0 = lane << sshift;
    int s1 = s0 + seg;
    if (s1 > n) s1 = n;
    const bool active = lane < nseg;
    const uint32_t *best = best_ws + (size_t)blockIdx.x * ZA_BEST_STRIDE;
    uint32_t *myb = rowb + lane * ZA_PROW;

    uint4 pb[8];                   // piece lane & 7 of the `best` rows of segments 8 j + (lane >> 3)
    auto prefetch = [&](int c) {
        const int rel = c * ZA_PCH;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int sg = 8 * j + (lane >> 3), off = (sg << sshift) + rel;
            pb[j] = make_uint4(0, 0, 0, 0);
            // (streamed once: a non-temporal load)
            if (c < seg / ZA_PCH && sg < nseg && off < n) {
                typedef uint32_t za_v4u __attribute__((ext_vector_type(4)));
                const za_v4u v = __builtin_nontemporal_load((const za_v4u *)(best + off + 4 * (lane & 7)));   // rows are 128-byte aligned
                pb[j] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };

    uint32_t ntok = 0;
    uint32_t crc_r = 0xFFFFFFFFu;
    int p = s0;
    uint32_t carry_b = 0;
    // Tokens are collected in the lane's own LDS row -- token j of a chunk takes slot j, which holds an entry the lane has
    // already read (a token consumes at least one position) -- and leave at the end of the chunk TRANSPOSED, like the loads:
    // eight lanes store the 16-byte pieces of one segment's tokens, so a store instruction touches a few whole lines.
    uint32_t nchunk = 0;           // tokens of the current chunk in my row
    prefetch(0);
#pragma unroll 1
    for (int c = 0; c < seg / ZA_PCH; c++) {
        const int cb = s0 + c * ZA_PCH;
        // wave-uniform early exit: every active lane is past its segment end
        if (__ballot(active && cb < s1) == 0ull) break;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t *r = rowb + (8 * j + (lane >> 3)) * ZA_PROW + 1 + 4 * (lane & 7);
            r[0] = pb[j].x; r[1] = pb[j].y; r[2] = pb[j].z; r[3] = pb[j].w;
        }
        myb[0] = carry_b;
        __builtin_amdgcn_wave_barrier();
        prefetch(c + 1);
        int ce = cb + ZA_PCH;
        if (ce > s1) ce = s1;
        if (active && cb < s1) {
            // CRC-32 over this chunk's bytes (zng_crc32_z at zlib_ngmodule.c:1741): the top bytes of four entries make a dword
#ifndef ZA_ABL_PARSE_NOCRC
            {
                int k = cb;
                for (; k + 4 <= ce; k += 4) {
                    const uint32_t *e = myb + 1 + (k - cb);
                    const uint32_t w4 = __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0703u) | __builtin_amdgcn_perm(e[3], e[2], 0x07030c0cu);
                    crc_r = crct[(crc_r ^ w4) & 0xFF] ^ (crc_r >> 8);
                    crc_r = crct[(crc_r ^ (w4 >> 8)) & 0xFF] ^ (crc_r >> 8);
                    crc_r = crct[(crc_r ^ (w4 >> 16)) & 0xFF] ^ (crc_r >> 8);
                    crc_r = crct[(crc_r ^ (w4 >> 24)) & 0xFF] ^ (crc_r >> 8);
                }
                for (; k < ce; k++) crc_r = crct[(crc_r ^ (myb[1 + k - cb] >> 24)) & 0xFF] ^ (crc_r >> 8);
            }
#endif
            {
                // one token per lane and round, literal and match on one predicated path (no divergent if/else); the chunk's
                // last position waits for the next chunk (its successor's entry is not here yet) unless the segment ends
                const int lim = ce == s1 ? ce : ce - 1;
                auto push = [&](uint32_t t) {
                    myb[nchunk] = t;
                    nchunk++;
                };
                while (p < lim) {
                    const uint32_t b = myb[p - cb + 1], bn = myb[p - cb + 2], e2 = myb[p - cb + 3];
                    // (greedy over the entries: on levels 4-9 the dynamic programme has rewritten them so that this IS its parse)
                    const uint32_t lf = ZA_ELEN(b), nlf = ZA_ELEN(bn);
                    const bool is_match = lf != 0u;
                    const int len = (int)lf;
                    const uint32_t lit = b >> 24;
                    const int dist = is_match ? (int)ZA_EDIST(b) : 1;
                    int lc, ln, le, dc, dn, de;
                    za_len_sym(is_match ? len : 3, lc, ln, le);
                    za_dist_sym(dist, dc, dn, de);
                    // a match token carries its symbols (length code << 26, extra << 21, distance code << 16, extra): the
                    // packer, which is VALU-bound, needs no symbol arithmetic.  A literal (a position without a match: length
                    // field 0) takes up to two more such positions into its token word --
                    // bytes in bits 0..23, count - 1 in bits 24..25 -- as far as this chunk's entries reach: most tokens of
                    // text are literals, 3.8 in a row, and fewer token words are fewer stores here and fewer loads in the packer.
                    // (Their entries were read together with this position's: one LDS round trip per round.)
                    const bool c1 = lf == 0u && p + 1 < ce && nlf == 0u;
                    const bool c2 = c1 && p + 2 < ce && ZA_ELEN(e2) == 0u;
                    const uint32_t l1 = bn >> 24, l2 = e2 >> 24;
                    const uint32_t t = is_match ? (0x80000000u | ((uint32_t)lc << 26) | ((uint32_t)le << 21) | ((uint32_t)dc << 16) | (uint32_t)de)
                                                : lit | (c1 ? l1 << 8 : 0u) | (c2 ? l2 << 16 : 0u) | (((c1 ? 1u : 0u) + (c2 ? 1u : 0u)) << 24);
                    push(t);
#ifndef ZA_ABL_PARSE_NOHIST
                    atomicAdd(&hist[is_match ? 257u + (uint32_t)lc : lit], 1u);
                    if (is_match) atomicAdd(&hist[288 + dc], 1u);
                    if (c1) atomicAdd(&hist[l1], 1u);
                    if (c2) atomicAdd(&hist[l2], 1u);
#endif
                    p += is_match ? len : 1 + (c1 ? 1 : 0) + (c2 ? 1 : 0);
                }
            }
            carry_b = myb[ZA_PCH];
        }
        // ---- the chunk's tokens leave: lane (8 j + g, piece) stores tokens 4 piece .. 4 piece + 3 of segment 8 j + g
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int sg = 8 * j + (lane >> 3);
            const uint32_t cnt = (uint32_t)__shfl((int)nchunk, sg, 64), at = (uint32_t)__shfl((int)ntok, sg, 64);
            const uint32_t first = 4u * (uint32_t)(lane & 7);
            // (a chunk holds at most 33 tokens: the ninth piece, one token, is stored by piece 7's lane as well)
            const uint32_t *r = rowb + sg * ZA_PROW;
            uint32_t *dst = tok_ws + (size_t)blockIdx.x * ZA_TOK_STRIDE + ((size_t)sg << sshift) + at;
            // whole 16-byte pieces, the last one with up to three slots of no meaning behind the chunk's tokens: the next chunk's
            // tokens land on them.  Only where that would leave the segment's 2 048 slots (a segment of nothing but literals)
            // the last piece goes token by token.
            if (first < cnt) {
                if (at + first + 4u <= (uint32_t)seg) { ZaU4u v; v.x = r[first]; v.y = r[first + 1]; v.z = r[first + 2]; v.w = r[first + 3]; *(ZaU4u *)(dst + first) = v; }
                else for (uint32_t i = first; i < cnt && i < first + 4u; i++) dst[i] = r[i];
            }
            if ((lane & 7) == 7 && cnt > 32u) dst[32] = r[32];
        }
        ntok += nchunk; nchunk = 0;
        __builtin_amdgcn_wave_barrier();
    }
    // ---- fold the per-segment CRCs: crc(A||B) = crc(A) * x^(8|B|) ^ crc(B)
    uint32_t cseg = 0;
    if (active) {
        cseg = crc_r ^ 0xFFFFFFFFu;
        if (lane < nseg - 1) {
            const int tail = n - ((nseg - 1) << sshift);
            uint32_t xt = 0x80000000u, sq = 0x00800000u;
            for (int m = tail; m; m >>= 1) { if (m & 1) xt = za_multmodp(sq, xt); sq = za_multmodp(sq, sq); }
            // x^(8 * seg * k), k = the whole segments between mine and the last: from the table for 2 KiB segments, worked out for
            // the smaller segments of small units (x^8 squared sshift times, raised to the k)
            uint32_t xk = 0x80000000u;
            if (sshift == ZA_SEG_SHIFT) xk = x8k_table[nseg - 2 - lane];
            else {
                uint32_t xs = 0x00800000u;
                for (int i = 0; i < sshift; i++) xs = za_multmodp(xs, xs);
                for (int m = nseg - 2 - lane; m; m >>= 1) { if (m & 1) xk = za_multmodp(xs, xk); xs = za_multmodp(xs, xs); }
            }
            cseg = za_multmodp(za_multmodp(xk, xt), cseg);
        }
    }
    cseg = za_wave_xor_reduce(cseg);
    if (lane == 0) crc_out[blockIdx.x] = cseg;
    segtok_ws[(size_t)blockIdx.x * ZA_MAX_SEGS + lane] = ntok;
    __syncthreads();
    if (lane == 0) hist[256] = 1;
    __syncthreads();
    for (int i = lane; i < ZA_HIST_STRIDE; i += 64) hist_ws[(size_t)blockIdx.x * ZA_HIST_STRIDE + i] = hist[i];
}

// level 0 (stored blocks): nothing to parse, only the unit's CRC-32
__global__ __launch_bounds__(64) void za_k_unit_crc(const uint8_t *__restrict__ in, const ZaUnit *__restrict__ units,
                                                    uint32_t *__restrict__ crc_out, const uint32_t *__restrict__ crc_table,
                                                    const uint32_t *__restrict__ x8k_table)
{
    __shared__ uint32_t crct[256];
    const ZaUnit u = units[blockIdx.x];
    for (int i = za_lane(); i < 256; i += 64) crct[i] = crc_table[i];
    __syncthreads();
    const uint32_t c = za_wave_crc32(in + u.in_off, (int)u.in_len, crct, x8k_table);
    if (za_lane() == 0) crc_out[blockIdx.x] = c;
}

// ------------------------------------------------------------------------------------------------
// k_plan : Huffman code lengths, block type, header bits
// ------------------------------------------------------------------------------------------------
struct ZaPlanLds {
    uint32_t key[288];
    __attribute__((aligned(16))) uint32_t A[288];
    uint32_t freq[320];
    uint8_t lens[320];
    uint16_t codes[320];
    uint8_t seq[320];
    uint16_t cltok[320];
    uint32_t clf[19];
    uint8_t cl_lens[19];
    uint16_t cl_codes[19];
    int m;
    int btype;
};

// rank-sort the non-zero symbols by (freq, index); all lanes take part.  Result in S.key[0..m).  The symbols in use are packed
// to the front first (ballots), so that a rank costs m comparisons instead of nsym (r06: a call of 1 KiB uses 50 of the 286
// literal/length symbols, a unit of text 120; the comparisons were 19 of the kernel's 105 us), four keys per LDS read.
__device__ void za_sort_syms(ZaPlanLds &S, const uint32_t *freq, int nsym)
{
    const int lane = za_lane();
    __syncthreads();
    int m = 0;
    for (int base = 0; base < nsym; base += 64) {
        const int i = base + lane;
        const uint32_t k = (i < nsym && freq[i]) ? ((freq[i] << 9) | (uint32_t)i) : 0u;
        const unsigned long long nz = __ballot(k != 0u);
        if (k) S.A[m + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u))] = k;
        m += (int)__builtin_popcountll(nz);
    }
    __syncthreads();
    for (int idx = lane; idx < m; idx += 64) {
        const uint32_t k = S.A[idx];
        int rank = 0;
        int j = 0;
        for (; j + 4 <= m; j += 4) {
            const uint4 o = *(const uint4 *)&S.A[j];
            rank += (o.x < k) + (o.y < k) + (o.z < k) + (o.w < k);
        }
        for (; j < m; j++) rank += S.A[j] < k;
        S.key[rank] = k;
    }
    __syncthreads();
    if (lane == 0) S.m = m;
    __syncthreads();
}

// lane 0 only: Moffat-Katajainen in-place lengths on S.key[0..m) + count-based limiting.
__device__ void za_lengths_serial(ZaPlanLds &S, int nsym, int limit, uint8_t *lens)
{
    const int m = S.m;
    for (int i = 0; i < nsym; i++) lens[i] = 0;
    if (m == 0) return;
    if (m == 1) { lens[S.key[0] & 511u] = 1; return; }
    uint32_t *A = S.A;
    for (int i = 0; i < m; i++) A[i] = S.key[i] >> 9;
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
    while (avbl > 0 && dpth < 320) {
        while (root >= 0 && (int)A[root] == dpth) { used++; root--; }
        while (avbl > used && next >= 0) { A[next--] = (uint32_t)dpth; avbl--; }
        if (next < 0) break;
        avbl = 2 * used; dpth++; used = 0;
    }
    int *cnt = (int *)(S.A + 256);                 // (LDS, behind the at most 256 symbols this lane-0 form is used for: a local array
                                                   // indexed by a variable would live in scratch memory)
    for (int i = 0; i < 17; i++) cnt[i] = 0;
    bool over = false;
    for (int i = 0; i < m; i++) {
        int d = (int)A[i];
        if (d > limit) { d = limit; over = true; }
        cnt[d]++;
    }
    if (over) {
        uint32_t total = 0;
        for (int i = 1; i <= limit; i++) total += (uint32_t)cnt[i] << (limit - i);
        // every step removes one unit of the Kraft sum; the guard only bounds a corrupted state
        for (int guard = 0; total != (1u << limit) && guard < (1 << 17); guard++) {
            cnt[limit]--;
            for (int i = limit - 1; i > 0; i--)
                if (cnt[i]) { cnt[i]--; cnt[i + 1] += 2; break; }
            total--;
        }
    }
    int idx = 0;
    for (int l = limit; l >= 1; l--)
        for (int k = 0; k < cnt[l] && idx < m; k++) lens[S.key[idx++] & 511u] = (uint8_t)l;
}

// The same lengths with the wave: only the two-queue merge that builds the tree (phase 1 of Moffat-Katajainen) is a chain of
// dependent steps, and it stays on lane 0 -- with the heads of its two queues kept in registers and the next two leaves fetched
// ahead, so that a step waits for one LDS round trip at most instead of four.  Everything behind it is done by all lanes: the
// depths of the internal nodes by pointer jumping (depth += depth of the parent, parent = the parent's parent: nine rounds at
// most for 285 nodes) instead of a walk down the array, the leaves per depth from the internal nodes per depth (a level holds
// twice the internal nodes of the level above it: leaves = 2 x inner[d - 1] - inner[d]) instead of the third pass, the lengths
// by rank.  S.cltok and S.clf are scratch here (the header is built later).  125 -> 50 us for the literal/length tree of a text
// unit: the latency of every batch too small to fill the device, and of every small call.
__device__ void za_lengths_wave(ZaPlanLds &S, int nsym, int limit, uint8_t *lens)
{
    const int lane = za_lane();
    __syncthreads();
    const int m = S.m;
    for (int i = lane; i < nsym; i += 64) lens[i] = 0;
    __syncthreads();
    if (m == 0) return;
    if (m == 1) { if (lane == 0) lens[S.key[0] & 511u] = 1; __syncthreads(); return; }
    uint32_t *A = S.A;
    uint16_t *dep = S.cltok;                                   // depth of internal node i
    for (int i = lane; i < m; i += 64) A[i] = S.key[i] >> 9;
    __syncthreads();
    if (lane == 0) {
        const uint32_t NONE = 0xFFFFFFFFu;                     // (no frequency and no sum of frequencies is that large)
        const uint32_t a0 = A[0] + A[1];
        A[0] = a0;
        int root = 0, leaf = 2;
        uint32_t wr = a0;                                      // weight of the internal node at the head of its queue (valid while root < next)
        uint32_t fl = leaf < m ? A[leaf] : NONE, fl1 = leaf + 1 < m ? A[leaf + 1] : NONE;      // the next two leaves
        for (int next = 1; next < m - 1; next++) {
            uint32_t w;
            // (the first pick always finds an internal node: the one the step in front made)
            if (fl == NONE || wr < fl) { w = wr; A[root] = (uint32_t)next; root++; wr = root < next ? A[root] : NONE; }
            else { w = fl; leaf++; fl = fl1; fl1 = leaf + 1 < m ? A[leaf + 1] : NONE; }
            if (fl == NONE || (root < next && wr < fl)) { w += wr; A[root] = (uint32_t)next; root++; wr = root < next ? A[root] : NONE; }
            else { w += fl; leaf++; fl = fl1; fl1 = leaf + 1 < m ? A[leaf + 1] : NONE; }
            A[next] = w;
            if (root == next) wr = w;                          // the queue was empty: the new node is its head
        }
    }
    __syncthreads();
    // internal nodes 0 .. m - 2, node m - 2 the root; A[i] = parent of node i
    const int ninner = m - 1, top = m - 2;
    int pi[5], di[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int i = lane + 64 * k;
        pi[k] = top; di[k] = 0;
        if (i < top) { pi[k] = (int)A[i]; di[k] = 1; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) { const int i = lane + 64 * k; if (i < ninner) { A[i] = (uint32_t)pi[k]; dep[i] = (uint16_t)di[k]; } }
    for (int round = 0; round < 10; round++) {
        __syncthreads();
        int np[5], nd[5];
        bool open = false;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = lane + 64 * k;
            np[k] = pi[k]; nd[k] = di[k];
            if (i < ninner) { np[k] = (int)A[pi[k]]; nd[k] = di[k] + (int)dep[pi[k]]; open = open || np[k] != top; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = lane + 64 * k;
            pi[k] = np[k]; di[k] = nd[k];
            if (i < ninner) { A[i] = (uint32_t)np[k]; dep[i] = (uint16_t)nd[k]; }
        }
        if (__ballot(open) == 0ull) break;
    }
    __syncthreads();
    // internal nodes per depth (in A, which is free now), then leaves per depth
    for (int i = lane; i < 288; i += 64) A[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) { const int i = lane + 64 * k; if (i < ninner) atomicAdd(&A[di[k]], 1u); }
    __syncthreads();
    uint32_t deep = 0, beyond = 0;                             // leaves at depth >= limit / > limit
    for (int d = lane + 1; d < 288; d += 64) {
        const uint32_t nl = 2u * A[d - 1] - A[d];
        if (d < limit) S.clf[d] = nl;
        deep += d >= limit ? nl : 0u;
        beyond += d > limit ? nl : 0u;
    }
    for (int x = 32; x >= 1; x >>= 1) { deep += __shfl_xor(deep, x, 64); beyond += __shfl_xor(beyond, x, 64); }
    __syncthreads();
    if (lane == 0) {
        uint32_t *cnt = S.clf;                                 // cnt[1 .. limit]
        cnt[limit] = deep;
        if (beyond) {
            uint32_t total = 0;
            for (int i = 1; i <= limit; i++) total += cnt[i] << (limit - i);
            // every step removes one unit of the Kraft sum; the guard only bounds a corrupted state
            for (int guard = 0; total != (1u << limit) && guard < (1 << 17); guard++) {
                cnt[limit]--;
                for (int i = limit - 1; i > 0; i--)
                    if (cnt[i]) { cnt[i]--; cnt[i + 1] += 2; break; }
                total--;
            }
        }
    }
    __syncthreads();
    // lengths by rank: the rarest symbols take the longest codes
    for (int r = lane; r < m; r += 64) {
        uint32_t acc = 0;
        int len = 1;
        for (int l = limit; l >= 1; l--) { acc += S.clf[l]; if ((uint32_t)r < acc) { len = l; break; } }
        lens[S.key[r] & 511u] = (uint8_t)len;
    }
    __syncthreads();
}

// The same lengths for an alphabet of at most 32 symbols (the code-length code's 19) without a single LDS round trip inside the
// algorithm: the array lives in ONE register across the lanes (lane i holds A[i]); every index is wave-uniform, so a read is a
// v_readlane, a write a compare and a select, and the control flow scalar.  (Lane 0 alone, through LDS: 15 us of the plan kernel's 105.)
__device__ void za_lengths_small(ZaPlanLds &S, int nsym, int limit, uint8_t *lens)
{
    const int lane = za_lane();
    __syncthreads();
    const int m = __builtin_amdgcn_readfirstlane(S.m);
    for (int i = lane; i < nsym; i += 64) lens[i] = 0;
    __syncthreads();
    if (m == 0) return;
    if (m == 1) { if (lane == 0) lens[S.key[0] & 511u] = 1; __syncthreads(); return; }
    const uint32_t mykey = lane < m ? S.key[lane] : 0u;
    uint32_t Av = mykey >> 9;
    auto RD = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)Av, i); };
    auto WR = [&](int i, uint32_t v) { Av = lane == i ? v : Av; };
    int root = 0, leaf = 2, next;
    WR(0, RD(0) + RD(1));
    for (next = 1; next < m - 1; next++) {
        uint32_t w;
        if (leaf >= m || RD(root) < RD(leaf)) { w = RD(root); WR(root, (uint32_t)next); root++; }
        else { w = RD(leaf); leaf++; }
        if (leaf >= m || (root < next && RD(root) < RD(leaf))) { w += RD(root); WR(root, (uint32_t)next); root++; }
        else { w += RD(leaf); leaf++; }
        WR(next, w);
    }
    WR(m - 2, 0u);
    for (next = m - 3; next >= 0; next--) WR(next, RD((int)RD(next)) + 1u);
    int avbl = 1, used = 0, dpth = 0;
    root = m - 2; next = m - 1;
    while (avbl > 0 && dpth < 64) {
        while (root >= 0 && (int)RD(root) == dpth) { used++; root--; }
        while (avbl > used && next >= 0) { WR(next, (uint32_t)dpth); next--; avbl--; }
        if (next < 0) break;
        avbl = 2 * used; dpth++; used = 0;
    }
    // Av: the depth of leaf `lane` (rank order).  Leaves per depth, cut at the limit: lane d holds cnt[d]
    const bool in = lane < m;
    const bool over = __ballot(in && (int)Av > limit) != 0ull;
    const int dcl = (int)Av > limit ? limit : (int)Av;
    uint32_t Cv = 0;
    for (int d = 1; d <= limit; d++) {
        const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(in && dcl == d));
        Cv = lane == d ? c : Cv;
    }
    auto RC = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)Cv, i); };
    auto WC = [&](int i, uint32_t v) { Cv = lane == i ? v : Cv; };
    if (over) {
        uint32_t total = 0;
        for (int i = 1; i <= limit; i++) total += RC(i) << (limit - i);
        // every step removes one unit of the Kraft sum; the guard only bounds a corrupted state
        for (int guard = 0; total != (1u << limit) && guard < (1 << 17); guard++) {
            WC(limit, RC(limit) - 1u);
            for (int i = limit - 1; i > 0; i--)
                if (RC(i)) { WC(i, RC(i) - 1u); WC(i + 1, RC(i + 1) + 2u); break; }
            total--;
        }
    }
    if (in) {
        uint32_t acc = 0;
        int len = 1;
        for (int l = limit; l >= 1; l--) { acc += RC(l); if ((uint32_t)lane < acc) { len = l; break; } }
        lens[mykey & 511u] = (uint8_t)len;
    }
    __syncthreads();
}

// lane 0 only: canonical codes, bit-reversed for LSB-first emission
__device__ void za_canon_serial(const uint8_t *lens, int n, uint16_t *codes, uint32_t *scratch32 /* 32 dwords of LDS */)
{
    uint32_t *bl = scratch32, *nc = scratch32 + 16;
    for (int i = 0; i < 16; i++) bl[i] = 0;
    for (int i = 0; i < n; i++) bl[lens[i]]++;
    bl[0] = 0;
    uint32_t code = 0;
    for (int b = 1; b <= 15; b++) { code = (code + bl[b - 1]) << 1; nc[b] = code; }
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        uint32_t r = 0;
        if (l) { const uint32_t c = nc[l]++; r = __brev(c) >> (32 - l); }
        codes[i] = (uint16_t)r;
    }
}

// all lanes: the same canonical codes, without lane 0 walking every symbol -- lengths counted with LDS atomics, the first code of
// every length by lane (15 steps each), a symbol's place among the symbols of its length from ballots (symbols in index order)
__device__ void za_canon_wave(const uint8_t *lens, int n, uint16_t *codes, uint32_t *scratch32 /* 32 dwords of LDS */)
{
    const int lane = za_lane();
    uint32_t *cnt = scratch32, *first = scratch32 + 16;
    __syncthreads();
    if (lane < 16) cnt[lane] = 0;
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const uint32_t l = lens[i]; if (l) atomicAdd(&cnt[l], 1u); }
    __syncthreads();
    if (lane >= 1 && lane < 16) {
        uint32_t code = 0;
        for (int b = 1; b <= lane; b++) code = (code + (b > 1 ? cnt[b - 1] : 0u)) << 1;
        first[lane] = code;
    }
    __syncthreads();
    uint32_t run[16];
#pragma unroll
    for (int b = 0; b < 16; b++) run[b] = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const uint32_t l = i < n ? lens[i] : 0u;
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t b = 1; b <= 15; b++) {
            const unsigned long long m = __ballot(l == b);
            if (l == b) rank = run[b] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            run[b] += (uint32_t)__builtin_popcountll(m);
        }
        if (i < n) codes[i] = l ? (uint16_t)(__brev(first[l] + rank) >> (32u - l)) : (uint16_t)0;
    }
    __syncthreads();
}

// serial LSB-first bit writer into a zero-initialised, 4-byte aligned slot (plain stores: the plan
// kernel is the first writer of its slot)
struct ZaBitW {
    uint32_t *out; uint32_t cap_words; uint32_t w; uint64_t acc; int nb; bool ovf;
    __device__ void put(uint32_t v, int n)
    {
        acc |= (uint64_t)v << nb; nb += n;
        if (nb >= 32) {
            if (w < cap_words) out[w] = (uint32_t)acc; else ovf = true;
            w++; acc >>= 32; nb -= 32;
        }
    }
    __device__ void finish() { if (nb > 0) { if (w < cap_words) out[w] = (uint32_t)acc; else ovf = true; } }
    __device__ uint32_t bits() const { return w * 32u + (uint32_t)nb; }
};

__constant__ uint8_t za_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Packed mode (hdr_ws != nullptr): the unit's final place is not known yet -- the header goes to a buffer of its own (ZA_HDR_STRIDE
// bytes per unit) and the unit's EXACT compressed size to unit_len (header + histogram x (code length + extra bits) + end of
// block + marker: everything that decides it is here), so that one prefix sum gives every unit its offset in the stream and the
// packer writes it there -- no slots, no gather pass.
#ifdef ZA_PLAN_STATS
__device__ unsigned long long za_plan_stat[16];   // profiling build only: clocks of the plan kernel's phases (lane 0 of unit 0), [15] = launches
#define ZA_PLAN_T(i) do { if (lane == 0 && blockIdx.x == 0) { const unsigned long long t_ = clock64(); za_plan_stat[i] += t_ - pt_; pt_ = t_; } } while (0)
#else
#define ZA_PLAN_T(i) do { } while (0)
#endif
#define ZA_HDR_STRIDE 640u      // bytes of header buffer per unit: the longest dynamic header is 3 + 14 + 57 + 316 x 14 bits = 563 bytes
__global__ __launch_bounds__(64) void za_k_plan(const ZaUnit *__restrict__ units, const uint32_t *__restrict__ hist_ws,
                                                uint32_t *__restrict__ code_ws, ZaPlan *__restrict__ plan_ws,
                                                uint8_t *__restrict__ out, uint32_t out_stride, int level,
                                                uint8_t *__restrict__ hdr_ws, uint32_t *__restrict__ unit_len)
{
    __shared__ ZaPlanLds S;
    const ZaUnit u = units[blockIdx.x];
    const int n = (int)u.in_len;
    const int lane = za_lane();
    const bool final = (u.flags & ZA_FLAG_FINAL) != 0;
    const bool flat = (u.flags & ZA_FLAG_FLATHDR) != 0;      // header form of indexed members
    ZaPlan plan; plan.btype = 0; plan.header_bits = 0; plan.pad0 = plan.pad1 = 0;
    uint32_t *code_out = code_ws + (size_t)blockIdx.x * ZA_CODE_STRIDE;
    if (n == 0 || level == 0) {
        if (lane == 0) {
            plan_ws[blockIdx.x] = plan;
            if (unit_len) {
                const uint32_t nchunks = ((uint32_t)n + 65534u) / 65535u;
                unit_len[blockIdx.x] = n == 0 ? (final ? 2u : 5u) : (uint32_t)n + 5u * nchunks + (final ? 0u : 5u);
            }
        }
        return;
    }
#ifdef ZA_PLAN_STATS
    unsigned long long pt_ = clock64();
    if (lane == 0 && blockIdx.x == 0) za_plan_stat[15] += 1;
#endif
    const uint32_t *hist = hist_ws + (size_t)blockIdx.x * ZA_HIST_STRIDE;
    for (int i = lane; i < 320; i += 64) S.freq[i] = hist[i];
    __syncthreads();
    if (lane == 0) {   // at least two distance codes
        int cntd = 0;
        for (int i = 0; i < 30; i++) cntd += S.freq[288 + i] != 0;
        if (cntd < 2 && S.freq[288] == 0) { S.freq[288] = 1; cntd++; }
        if (cntd < 2) S.freq[289] = 1;
    }
    __syncthreads();
    // lit/len tree
    ZA_PLAN_T(0);
    za_sort_syms(S, S.freq, 286);
    ZA_PLAN_T(1);
    za_lengths_wave(S, 286, ZA_LIMIT_L, S.lens);
    ZA_PLAN_T(2);
    if (lane < 2) S.lens[286 + lane] = 0;
    // distance tree
    za_sort_syms(S, S.freq + 288, 30);
    za_lengths_small(S, 30, ZA_LIMIT_D, S.lens + 288);           // (30 symbols: one register across the lanes)
    if (lane < 2) S.lens[318 + lane] = 0;
    __syncthreads();
    ZA_PLAN_T(3);

    // canonical codes and the exact data costs: by all lanes (lane 0 alone walked 600 symbols through LDS, one round trip each)
    za_canon_wave(S.lens, 286, S.codes, S.A);
    za_canon_wave(S.lens + 288, 30, S.codes + 288, S.A);
    uint32_t cost_dd = 0, cost_df = 0;
    for (int s = lane; s < 286; s += 64) {
        const uint32_t f = hist[s];
        const int ex = s >= 257 ? za_len_extra_bits(s - 257) : 0;
        const int fx = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        cost_dd += f * (uint32_t)(S.lens[s] + ex);
        cost_df += f * (uint32_t)(fx + ex);
    }
    if (lane < 30) {
        const uint32_t f = hist[288 + lane];
        const int ex = za_dist_extra_bits(lane);
        cost_dd += f * (uint32_t)(S.lens[288 + lane] + ex);
        cost_df += f * (uint32_t)(5 + ex);
    }
    for (int d = 32; d >= 1; d >>= 1) { cost_dd += __shfl_xor(cost_dd, d, 64); cost_df += __shfl_xor(cost_df, d, 64); }
    ZA_PLAN_T(4);
    // the code-length sequence of the header: hlit literal/length lengths, then hdist distance lengths (copied by all lanes)
    int hlit, hdist;
    {
        unsigned long long nz[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { const int i = lane + 64 * k; nz[k] = __ballot(i < 286 && S.lens[i] != 0); }
        hlit = 257;
#pragma unroll
        for (int k = 0; k < 5; k++) if (nz[k]) { const int t = 64 * k + 64 - __builtin_clzll(nz[k]); hlit = t > hlit ? t : hlit; }
        const unsigned long long nzd = __ballot(lane < 30 && S.lens[288 + lane] != 0);
        hdist = nzd ? 64 - __builtin_clzll(nzd) : 1;
        for (int i = lane; i < hlit; i += 64) S.seq[i] = S.lens[i];
        if (lane < hdist) S.seq[hlit + lane] = S.lens[288 + lane];
        __syncthreads();
    }
    // Run-length encoding of the code lengths (tokens: sym | extra << 8), by all lanes (r06; lane 0 alone took 31 us for it, every
    // entry an LDS round trip): the first entry of every run works out its run's tokens in closed form -- a run of zeros is
    // 138s, then one 18 / 17 for a rest of 3 or more, else the rest as literals; any other value is itself once, then 6s, then
    // one 16 for a rest of 3 or more, else literals -- and a prefix sum over the runs' token counts gives them their places.
    const int nseq = hlit + hdist;
    int nt = 0;
    {
        if (lane < 19) S.clf[lane] = 0;
        uint32_t vv[5];
        unsigned long long st[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = lane + 64 * k;
            vv[k] = i < nseq ? S.seq[i] : 0xFFu;
            const uint32_t pv = (i > 0 && i < nseq) ? S.seq[i - 1] : 0xFEu;
            st[k] = __ballot(i < nseq && vv[k] != pv);
        }
        __syncthreads();                                            // (the counters are clear)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = lane + 64 * k;
            const bool start = ((st[k] >> lane) & 1ull) != 0ull;
            // the next run's first entry: behind me in my own group of 64, or the first of a later group, or the end
            int later = nseq;
#pragma unroll
            for (int k2 = 4; k2 > k; k2--) if (st[k2]) later = 64 * k2 + (int)__builtin_ctzll(st[k2]);
            const unsigned long long behind = lane == 63 ? 0ull : st[k] >> (lane + 1);
            const int nxt = behind ? i + 1 + (int)__builtin_ctzll(behind) : later;
            const int run = nxt - i;
            const uint32_t v = vv[k];
            int q, rem, cnt;
            if (v == 0u) { q = run / 138; rem = run - 138 * q; cnt = q + (rem >= 3 ? 1 : rem); }
            else { q = (run - 1) / 6; rem = run - 1 - 6 * q; cnt = 1 + q + (rem >= 3 ? 1 : rem); }
            const uint32_t c = start ? (uint32_t)cnt : 0u;
            const uint32_t incl = za_wave_incl_scan(c);
            if (start) {
                int o = nt + (int)(incl - c);
                if (v == 0u) {
                    for (int j = 0; j < q; j++) S.cltok[o++] = (uint16_t)(18u | (127u << 8));
                    if (rem >= 11) S.cltok[o++] = (uint16_t)(18u | ((uint32_t)(rem - 11) << 8));
                    else if (rem >= 3) S.cltok[o++] = (uint16_t)(17u | ((uint32_t)(rem - 3) << 8));
                    else for (int j = 0; j < rem; j++) S.cltok[o++] = 0;
                    if (q + (rem >= 11)) atomicAdd(&S.clf[18], (uint32_t)(q + (rem >= 11)));
                    if (rem >= 3 && rem < 11) atomicAdd(&S.clf[17], 1u);
                    if (rem < 3 && rem) atomicAdd(&S.clf[0], (uint32_t)rem);
                } else {
                    S.cltok[o++] = (uint16_t)v;
                    for (int j = 0; j < q; j++) S.cltok[o++] = (uint16_t)(16u | (3u << 8));
                    if (rem >= 3) S.cltok[o++] = (uint16_t)(16u | ((uint32_t)(rem - 3) << 8));
                    else for (int j = 0; j < rem; j++) S.cltok[o++] = (uint16_t)v;
                    atomicAdd(&S.clf[v], 1u + (uint32_t)(rem < 3 ? rem : 0));
                    if (q + (rem >= 3)) atomicAdd(&S.clf[16], (uint32_t)(q + (rem >= 3)));
                }
            }
            nt += (int)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __syncthreads();
    }
    ZA_PLAN_T(5);
    // the code-length alphabet: 19 symbols, ranked by (count, index) with the keys in registers (lane = symbol)
    {
        const uint32_t key = (lane < 19 && S.clf[lane]) ? ((S.clf[lane] << 9) | (uint32_t)lane) : 0u;
        const int m = (int)__builtin_popcountll(__ballot(key != 0u));
        int rank = 0;
        for (int j = 0; j < 19; j++) { const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, j); rank += (kj != 0u && kj < key); }
        if (key) S.key[rank] = key;
        if (lane == 0) S.m = m;
    }
    za_lengths_small(S, 19, 7, S.cl_lens);
    za_canon_wave(S.cl_lens, 19, S.cl_codes, S.A + 192);
    int hclen;
    {
        const unsigned long long nzc = __ballot(lane < 19 && S.cl_lens[za_cl_order[lane < 19 ? lane : 0]] != 0);
        hclen = nzc ? 64 - (int)__builtin_clzll(nzc) : 0;
        hclen = hclen < 4 ? 4 : hclen;
    }
    ZA_PLAN_T(6);
    // exact costs (every lane ends up with the sums)
    uint32_t tokbits = 0;
    for (int k = lane; k < nt; k += 64) {
        const int sy = S.cltok[k] & 0xFF;
        tokbits += (uint32_t)S.cl_lens[sy] + (sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u);
    }
    for (int d = 32; d >= 1; d >>= 1) tokbits += __shfl_xor(tokbits, d, 64);
    const unsigned long long data_dyn = cost_dd, data_fix = cost_df;        // (at most 131 072 tokens of at most 48 bits: 32 bits hold the sums)
    unsigned long long hdr_dyn = 3 + 5 + 5 + 4 + 3ull * (unsigned)hclen + tokbits;
    if (flat) hdr_dyn = 3 + 5 + 5 + 4 + 3 * 19 + 4ull * (unsigned)(hlit + hdist);
    const unsigned long long cost_dyn = hdr_dyn + data_dyn, cost_fix = 3 + data_fix;
    const unsigned long long nchunks = ((unsigned long long)n + 65534ull) / 65535ull;
    const unsigned long long cost_sto = 8ull * ((unsigned long long)n + 5ull * nchunks);
    unsigned long long bestc = cost_dyn; int btype = 2;
    if (cost_fix <= bestc) { bestc = cost_fix; btype = 1; }
    if (cost_sto <= bestc) { bestc = cost_sto; btype = 0; }
    ZA_PLAN_T(7);
    plan.btype = (uint32_t)btype;
    if (unit_len && lane == 0) {
        // the unit's size, exactly: what the packer will write (it checks)
        const unsigned long long bits = btype == 2 ? cost_dyn : cost_fix;       // header + tokens + end of block
        const unsigned long long endbit = bits + (final ? 0ull : 3ull);
        unit_len[blockIdx.x] = btype == 0 ? (uint32_t)n + 5u * (uint32_t)nchunks + (final ? 0u : 5u)
                                          : (uint32_t)((endbit + 7ull) >> 3) + (final ? 0u : 4u);
    }
    if (btype != 0) {
        // The header's bits, by all lanes: every field knows its place (the code-length tokens' from a prefix sum over their
        // sizes) and is ORed into a zeroed image in LDS, whose dwords then leave together (lane 0 alone: 13 us).
        uint32_t *img = S.A;                                       // 160 dwords (ZA_HDR_STRIDE bytes); S.A + 192 .. 224 was the canonical codes' scratch
        uint32_t *dst = hdr_ws ? (uint32_t *)(hdr_ws + (size_t)blockIdx.x * ZA_HDR_STRIDE) : (uint32_t *)(out + (size_t)blockIdx.x * out_stride);
        const uint32_t cap_words = hdr_ws ? ZA_HDR_STRIDE / 4 : out_stride / 4;
        __syncthreads();
        for (int i = lane; i < (int)(ZA_HDR_STRIDE / 4); i += 64) img[i] = 0;
        __syncthreads();
        auto put = [&](uint32_t v, uint32_t pos, uint32_t len) {     // len <= 17 bits at bit `pos`
            const uint32_t wd = pos >> 5, sh = pos & 31u;
            atomicOr(&img[wd], v << sh);
            if (sh + len > 32u) atomicOr(&img[wd + 1], v >> (32u - sh));
        };
        uint32_t hbits = 3;
        if (btype == 2) {
            const uint32_t hc = flat ? 19u : (uint32_t)hclen;
            if (lane == 0) put((uint32_t)final | (2u << 1) | ((uint32_t)(hlit - 257) << 3) | ((uint32_t)(hdist - 1) << 8) | ((hc - 4u) << 13), 0u, 17u);
            if (lane < (int)hc) put(flat ? (za_cl_order[lane] < 16 ? 4u : 0u) : (uint32_t)S.cl_lens[za_cl_order[lane]], 17u + 3u * (uint32_t)lane, 3u);
            hbits = 17u + 3u * hc;
            if (flat) {
                // flat form: the code-length code is the 4-bit code of the symbols 0..15 (code(s) = s), no run lengths
                for (int k = lane; k < nseq; k += 64) {
                    const uint32_t v = S.seq[k];
                    put(((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3), hbits + 4u * (uint32_t)k, 4u);
                }
                hbits += 4u * (uint32_t)nseq;
            } else {
                for (int base = 0; base < nt; base += 64) {
                    const int k = base + lane;
                    uint32_t val = 0, len = 0;
                    if (k < nt) {
                        const uint32_t sy = S.cltok[k] & 0xFFu, ex = S.cltok[k] >> 8;
                        const uint32_t cl = S.cl_lens[sy];
                        val = (uint32_t)S.cl_codes[sy] | (ex << cl);
                        len = cl + (sy == 16u ? 2u : sy == 17u ? 3u : sy == 18u ? 7u : 0u);
                    }
                    const uint32_t incl = za_wave_incl_scan(len);
                    if (k < nt) put(val, hbits + incl - len, len);
                    hbits += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
        } else if (lane == 0) put((uint32_t)final | (1u << 1), 0u, 3u);
        __syncthreads();
        for (uint32_t i = (uint32_t)lane; i < (hbits + 31u) / 32u && i < cap_words; i += 64u) dst[i] = img[i];
        plan.header_bits = hbits;
    }
    if (lane == 0) { S.btype = btype; plan_ws[blockIdx.x] = plan; }
    ZA_PLAN_T(8);
    __syncthreads();
    if (S.btype == 1) {
        // the fixed code (RFC 1951 3.2.6), bit-reversed for LSB-first emission: lane 0 alone walked 600 LDS round trips for it,
        // and the smallest inputs -- whose latency is this kernel's -- are the ones that take fixed blocks
        for (int i = lane; i < 320; i += 64) {
            uint32_t len = 0, code = 0;
            if (i < 144) { len = 8; code = 0x30u + (uint32_t)i; }
            else if (i < 256) { len = 9; code = 0x190u + (uint32_t)(i - 144); }
            else if (i < 280) { len = 7; code = (uint32_t)(i - 256); }
            else if (i < 288) { len = 8; code = 0xC0u + (uint32_t)(i - 280); }
            else { len = 5; code = (uint32_t)(i - 288); }            // 288 .. 319: distance codes 0 .. 31
            S.lens[i] = (uint8_t)len;
            S.codes[i] = i < 318 ? (uint16_t)(__brev(code) >> (32u - len)) : (uint16_t)0;
        }
        __syncthreads();
    }
    for (int i = lane; i < 320; i += 64) code_out[i] = (uint32_t)S.codes[i] | ((uint32_t)S.lens[i] << 16);
    ZA_PLAN_T(9);
}

// ------------------------------------------------------------------------------------------------
// k_pack
// ------------------------------------------------------------------------------------------------
// One wave per unit.  Stored blocks are copied; fixed and dynamic blocks are packed token-parallel (see below): 64 token words
// per step, places from a wave prefix sum over their bit lengths, bits ORed into an LDS ring, complete dwords stored 64 at a time.
#define ZA_PK_RING 256  // dwords of the packer's output ring (a power of two; a group of 64 token words fills at most 96)

// Packed mode (dst_off != nullptr): the unit is written at out + dst_off[unit], any byte address, its size as the plan kernel
// worked it out (out_len on entry; a different size is ZA_ST_SIZE); the stream's first and last dword are shared with the units
// beside it and written byte by byte.  out_stride is then the room behind the unit's place: the end of the destination.
#define ZA_ST_SIZE 2u           // (packed mode) the packer's size differs from the planned one
__global__ __launch_bounds__(64) void za_k_pack(const uint8_t *__restrict__ in, const ZaUnit *__restrict__ units,
                                                const uint32_t *__restrict__ tok_ws, const uint32_t *__restrict__ segtok_ws,
                                                const uint32_t *__restrict__ code_ws, const ZaPlan *__restrict__ plan_ws,
                                                uint32_t *__restrict__ segbits_ws, uint32_t *__restrict__ cidx_ws,
                                                uint8_t *__restrict__ out,
                                                uint32_t out_stride, uint32_t *__restrict__ out_len,
                                                uint32_t *__restrict__ status,
                                                const uint64_t *__restrict__ dst_off, uint64_t dst_cap, const uint8_t *__restrict__ hdr_ws)
{
    __shared__ uint32_t codes[ZA_CODE_STRIDE];
    __shared__ uint32_t ring[ZA_PK_RING];            // the open dwords of the unit's bit stream
    const ZaUnit u = units[blockIdx.x];
    const uint8_t *data = in + u.in_off;
    const int n = (int)u.in_len;
    const int lane = za_lane();
    const bool final = (u.flags & ZA_FLAG_FINAL) != 0;
    const ZaPlan plan = plan_ws[blockIdx.x];
    const bool packed = dst_off != nullptr;
    const uint64_t my_off = packed ? dst_off[blockIdx.x] : 0ull;
    const uint32_t planned = packed ? out_len[blockIdx.x] : 0u;
    if (packed) {
        // the planned size must fit the destination: nothing is written otherwise
        out_stride = my_off + (uint64_t)planned <= dst_cap ? planned : 0u;
        if (planned && !out_stride) { if (lane == 0) { out_len[blockIdx.x] = 0u; status[blockIdx.x] = ZA_ST_OVERFLOW; } return; }
    }
    uint8_t *slot = packed ? out + my_off : out + (size_t)blockIdx.x * out_stride;
    // the bit stream is built in dwords of an ALIGNED base: a unit at an odd byte address starts at bit 8 * (address & 3) of dword 0
    const uint32_t b0 = packed ? (uint32_t)((uintptr_t)slot & 3u) : 0u, bit0 = 8u * b0;
    uint32_t *slot32 = (uint32_t *)(slot - b0);
    const uint32_t cap_words = packed ? (b0 + out_stride + 3u) / 4u : out_stride / 4;
    uint32_t *segbits = segbits_ws + (size_t)blockIdx.x * ZA_SEGB_STRIDE;
    uint32_t *cidx = cidx_ws + (size_t)blockIdx.x * ZA_CIDX_STRIDE;       // chunk index of indexed members (oracle: chunk_idx)
    const int sshift = ZA_UNIT_SEG_SHIFT(units[blockIdx.x].flags);
    const int nseg = (n + (1 << sshift) - 1) >> sshift;
    bool ovf = false;
    uint32_t total_bytes = 0;

    if (n == 0) {
        if (lane == 0) {
            if (final) { slot[0] = 0x03; slot[1] = 0x00; total_bytes = 2; }
            else { slot[0] = 0; slot[1] = 0; slot[2] = 0; slot[3] = 0xFF; slot[4] = 0xFF; total_bytes = 5; }
            out_len[blockIdx.x] = total_bytes; status[blockIdx.x] = 0;
        }
        for (int i = lane; i < ZA_SEGB_STRIDE; i += 64) segbits[i] = 0;
        for (int i = lane; i < ZA_CIDX_STRIDE; i += 64) cidx[i] = 0;
        return;
    }
    if (plan.btype == 0) {
        // stored blocks, 65535 bytes at most each; unit starts byte aligned
        const uint32_t nchunks = ((uint32_t)n + 65534u) / 65535u;
        const uint32_t need = (uint32_t)n + 5u * nchunks + (final ? 0u : 5u);
        if (need > out_stride) ovf = true;
        else {
            for (uint32_t c = 0; c < nchunks; c++) {
                const uint32_t off = c * 65535u;
                const uint32_t len = (uint32_t)n - off > 65535u ? 65535u : (uint32_t)n - off;
                uint8_t *dst = slot + off + 5u * c;
                if (lane == 0) {
                    dst[0] = (uint8_t)((final && c == nchunks - 1) ? 1 : 0);
                    dst[1] = (uint8_t)(len & 0xFF); dst[2] = (uint8_t)(len >> 8);
                    dst[3] = (uint8_t)(~len & 0xFF); dst[4] = (uint8_t)((~len >> 8) & 0xFF);
                }
                // the payload: 16 bytes per lane and round (both sides at any alignment), the tail bytewise
                const uint32_t full = len & ~15u;
                for (uint32_t i = 16u * (uint32_t)lane; i < full; i += 16u * 64u) *(ZaU4u *)(dst + 5 + i) = *(const ZaU4u *)(data + off + i);
                for (uint32_t i = full + (uint32_t)lane; i < len; i += 64) dst[5 + i] = data[off + i];
            }
            total_bytes = (uint32_t)n + 5u * nchunks;
            if (!final && lane == 0) {
                uint8_t *t = slot + total_bytes;
                t[0] = 0; t[1] = 0; t[2] = 0; t[3] = 0xFF; t[4] = 0xFF;
            }
            if (!final) total_bytes += 5;
        }
        for (int i = lane; i < ZA_SEGB_STRIDE; i += 64) segbits[i] = 0;
        for (int i = lane; i < ZA_CIDX_STRIDE; i += 64) cidx[i] = 0;
        if (lane == 0) { out_len[blockIdx.x] = ovf ? 0u : total_bytes; status[blockIdx.x] = ovf ? ZA_ST_OVERFLOW : (packed && total_bytes != planned) ? ZA_ST_SIZE : 0u; }
        return;
    }

    for (int i = lane; i < ZA_CODE_STRIDE; i += 64) codes[i] = code_ws[(size_t)blockIdx.x * ZA_CODE_STRIDE + i];
    for (int i = lane; i < ZA_PK_RING; i += 64) ring[i] = 0;
    __syncthreads();
    // ---- token-parallel packing.  The unit's bit stream is produced in order, one segment after the other, 64 token words at a
    // time: lane j takes word j of the group (one coalesced 256-byte load, the next group's already in flight), looks its codes
    // up, a wave prefix sum over the bit lengths gives every word its place, and the lanes OR their bits into an LDS ring of the
    // stream's open dwords; dwords that are complete leave at once, 64 per store.  No second walk over the tokens to size the
    // segments first (their starts fall out of the running offset), no lane waiting for the segment with the most tokens, no
    // lane writing single dwords into lines of its own (the packer's 45 KB of output cost 365 KB of memory writes that way).
    const uint32_t *tok_unit = tok_ws + (size_t)blockIdx.x * ZA_TOK_STRIDE;
    const uint32_t mycnt = lane < nseg ? segtok_ws[(size_t)blockIdx.x * ZA_MAX_SEGS + lane] : 0u;
    uint32_t bitpos = packed ? bit0 : plan.header_bits;      // (wave-uniform) bits of the stream so far, counted from bit 0 of slot32
    uint32_t wbase = bitpos >> 5;                  // the first dword that is still open; ring slot = dword index mod ZA_PK_RING
    if (!packed && (bitpos & 31u) && lane == 0) ring[wbase & (ZA_PK_RING - 1)] = wbase < cap_words ? slot32[wbase] : 0u;     // the header's last, partial dword (plan kernel)
    __syncthreads();
    const uint32_t end_byte = b0 + planned;        // (packed) first byte behind the unit, counted from slot32
    // `v` (at most 37 bits... 48 with a fixed block's longer codes) of `nb` bits from every lane, in lane order, behind `bitpos`
    auto emit = [&](uint64_t v, uint32_t nb) {
        const uint32_t incl = za_wave_incl_scan(nb);
        const uint32_t p = bitpos + incl - nb;
        const uint32_t w = p >> 5, sh = p & 31u;
        if (nb) {
            const uint64_t x = v << sh;
            const uint32_t top = sh ? (uint32_t)(v >> (64u - sh)) : 0u;
            atomicOr(&ring[w & (ZA_PK_RING - 1)], (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(&ring[(w + 1u) & (ZA_PK_RING - 1)], (uint32_t)(x >> 32));
            if (top) atomicOr(&ring[(w + 2u) & (ZA_PK_RING - 1)], top);
        }
        bitpos += (uint32_t)__shfl((int)incl, 63, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the complete dwords leave (at most 64 x 48 bits = 96 of them per call)
        const uint32_t nw = (bitpos >> 5) - wbase;
        for (uint32_t i = (uint32_t)lane; i < nw; i += 64) {
            const uint32_t d = wbase + i;
            const uint32_t val = ring[d & (ZA_PK_RING - 1)];
            ring[d & (ZA_PK_RING - 1)] = 0;
            if (d >= cap_words) ovf = true;
            else if (packed && (d == 0u ? b0 != 0u : 4u * d + 4u > end_byte)) {
                // a dword shared with the unit in front (its low bytes) or behind (its high bytes): my bytes only
                for (uint32_t k = d == 0u ? b0 : 0u; k < 4u && 4u * d + k < end_byte; k++) ((uint8_t *)slot32)[4u * d + k] = (uint8_t)(val >> (8u * k));
            } else slot32[d] = val;
        }
        wbase += nw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (packed) {
        // the header, from its buffer: 64 dwords per step through the same ring (the last one cut to what is left)
        const uint32_t *hw = (const uint32_t *)(hdr_ws + (size_t)blockIdx.x * ZA_HDR_STRIDE);
        for (uint32_t base = 0; base < plan.header_bits; base += 64u * 32u) {
            const uint32_t at = base + 32u * (uint32_t)lane;
            const uint32_t nb = at < plan.header_bits ? (plan.header_bits - at < 32u ? plan.header_bits - at : 32u) : 0u;
            uint32_t v = nb ? hw[at >> 5] : 0u;
            if (nb < 32u) v &= (1u << nb) - 1u;
            emit((uint64_t)v, nb);
        }
    }
    for (int sg = 0; sg < nseg; sg++) {
        const uint32_t cnt = (uint32_t)__shfl((int)mycnt, sg, 64);
        if (lane == 0) { segbits[sg] = bitpos - bit0; cidx[sg] = bitpos - bit0; }      // (index entries: the codec forces a token boundary at every segment start)
        const uint32_t *tk = tok_unit + ((size_t)sg << sshift);
        uint32_t tnext = (uint32_t)lane < cnt ? tk[lane] : 0u;
        for (uint32_t g = 0; g < cnt; g += 64) {
            const uint32_t t = tnext;
            tnext = g + 64u + (uint32_t)lane < cnt ? tk[g + 64u + lane] : 0u;
            const bool has = g + (uint32_t)lane < cnt;
            const bool m = (t & 0x80000000u) != 0u;
            const int lc = (int)((t >> 26) & 31u), dc = m ? (int)((t >> 16) & 31u) : 0;
            const int ln = m ? za_len_extra_bits(lc) : 0, dn = za_dist_extra_bits(dc);
            const uint32_t le = m ? (t >> 21) & 31u : 0u, de = t & 0x1FFFu;
            const uint32_t cl = codes[m ? 257u + (uint32_t)lc : (t & 0xFFu)], cd = codes[288 + dc];
            // literal / length code + extra (at most 20 bits), then the distance part of a match (at most 28) -- or the second and
            // third literal of a literal word
            const uint32_t nl = m ? 0u : (t >> 24) & 3u;
            const uint32_t c1 = codes[(t >> 8) & 0xFFu], c2 = codes[(t >> 16) & 0xFFu];
            const uint32_t a = (cl & 0xFFFFu) | (le << (cl >> 16)), an = (cl >> 16) + (uint32_t)ln;
            const uint32_t lits = (c1 & 0xFFFFu) | (nl > 1u ? (c2 & 0xFFFFu) << (c1 >> 16) : 0u);
            const uint32_t nlits = (c1 >> 16) + (nl > 1u ? (c2 >> 16) : 0u);
            const uint32_t b = m ? (cd & 0xFFFFu) | (de << (cd >> 16)) : nl ? lits : 0u;
            const uint32_t bn = m ? (cd >> 16) + (uint32_t)dn : nl ? nlits : 0u;
            emit(has ? (uint64_t)a | ((uint64_t)b << an) : 0ull, has ? an + bn : 0u);
        }
    }
    const uint32_t end_all = bitpos - bit0;                       // bit offset of the end-of-block code
    for (int i = lane; i <= ZA_MAX_SEGS; i += 64) if (i >= nseg) segbits[i] = end_all;
    if (lane == 0) { segbits[ZA_MAX_SEGS] = end_all; cidx[nseg] = end_all; }
    // tail: EOB, then final padding or the sync-flush marker (empty stored block): one more group with a single word
    {
        const uint32_t ce = codes[256];
        uint64_t v = ce & 0xFFFFu;
        uint32_t nb = ce >> 16;
        if (!final) nb += 3;                                      // 000: a stored block, not final
        const uint32_t endbit = end_all + nb;
        const uint32_t padded = (endbit + 7u) & ~7u;
        nb += padded - endbit;
        total_bytes = padded >> 3;
        if (!final) { v |= 0xFFFF0000ull << nb; nb += 32; total_bytes += 4; }      // LEN = 0, NLEN = 0xFFFF
        emit(lane == 0 ? v : 0ull, lane == 0 ? nb : 0u);
        // the last, partial dword (packed: my bytes of it)
        if ((bitpos & 31u) && lane == 0) {
            if (wbase >= cap_words) ovf = true;
            else if (!packed) slot32[wbase] = ring[wbase & (ZA_PK_RING - 1)];
            else {
                const uint32_t val = ring[wbase & (ZA_PK_RING - 1)];
                for (uint32_t k = wbase == 0u ? b0 : 0u; k < 4u && 4u * wbase + k < end_byte; k++) ((uint8_t *)slot32)[4u * wbase + k] = (uint8_t)(val >> (8u * k));
            }
        }
        if (total_bytes > out_stride) ovf = true;
    }
    const unsigned long long anyovf = __ballot(ovf);
    if (lane == 0) {
        const bool wrong = packed && !anyovf && total_bytes != planned;
        out_len[blockIdx.x] = anyovf ? 0u : total_bytes;
        status[blockIdx.x] = anyovf ? ZA_ST_OVERFLOW : wrong ? ZA_ST_SIZE : 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// compaction: gather the per-unit slots of each reference block / of the whole batch into a
// contiguous stream at byte offsets computed by a prefix sum (host or device side).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void za_k_gather(const uint8_t *__restrict__ slots, uint32_t out_stride,
                                                   const uint32_t *__restrict__ out_len,
                                                   const uint64_t *__restrict__ dst_off, uint8_t *__restrict__ dst)
{
    const uint8_t *src = slots + (size_t)blockIdx.x * out_stride;
    uint8_t *d = dst + dst_off[blockIdx.x];
    const uint32_t len = out_len[blockIdx.x];
    // head bytes until the destination is 4-byte aligned, then dwords, then the tail
    uint32_t headb = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u);
    if (headb > len) headb = len;
    if (threadIdx.x < headb) d[threadIdx.x] = src[threadIdx.x];
    const uint32_t nw = (len - headb) >> 2;
    uint32_t *d32 = (uint32_t *)(d + headb);
    for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) d32[i] = za_ld32(src + headb + 4u * i);
    const uint32_t done = headb + 4u * nw;
    if (threadIdx.x < len - done) d[done + threadIdx.x] = src[done + threadIdx.x];
}

// exclusive prefix sum of unit sizes -> byte offsets (single workgroup; batches are <= a few 10^5)
// units != nullptr: indexed members -- every unit also takes 4 bytes of index per 256 bytes of its input
__global__ __launch_bounds__(1024) void za_k_offsets(const uint32_t *__restrict__ out_len, uint32_t n, uint32_t extra,
                                                     uint64_t base, uint64_t *__restrict__ dst_off,
                                                     uint64_t *total, const ZaUnit *__restrict__ units,
                                                     const uint64_t *d_base = nullptr,      // (+ *d_base: the total of the launches in front; may BE `total`: no __restrict__ on either)
                                                     int running = 0)                       // the total continues d_base's (a first launch passes no d_base and starts it)
{
    if (d_base) base += *d_base;
    auto ext = [&](uint32_t i) -> unsigned long long {
        return (unsigned long long)extra + (units ? 4ull * ((units[i].in_len + (1u << ZA_CHUNK_SHIFT) - 1u) >> ZA_CHUNK_SHIFT) : 0ull);
    };
    __shared__ unsigned long long part[1024];
    const uint32_t tid = threadIdx.x, nth = blockDim.x;          // (64 threads for the few units of a small call: thread 0 walks `nth` partial sums)
    const uint32_t per = (n + nth - 1u) / nth;
    const uint32_t b = tid * per, e = (b + per < n) ? b + per : n;
    unsigned long long s = 0;
    for (uint32_t i = b; i < e; i++) s += (unsigned long long)out_len[i] + ext(i);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = base;
        const uint32_t used = per ? (n + per - 1u) / per : 0u;    // threads that hold anything
        for (uint32_t i = 0; i < used; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        *total = (d_base || running) ? run : run - base;
    }
    __syncthreads();
    if (b >= n) return;
    unsigned long long run = part[tid];
    for (uint32_t i = b; i < e; i++) { dst_off[i] = run; run += (unsigned long long)out_len[i] + ext(i); }
}
