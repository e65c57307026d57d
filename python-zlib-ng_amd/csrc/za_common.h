// Shared definitions for the MI355X (gfx950) DEFLATE / inflate engine.
//
// Product code.  Nothing here includes, links or calls anything under oracle/.
// The codec ("ZA codec") is specified in DESIGN.md section 3; the kernels implement it with
// 64-lane wavefronts, LDS-resident hash-chain windows and lane-per-segment parsing/packing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZA_SEG        2048
#define ZA_SEG_SHIFT  11
#define ZA_MAX_UNIT   131072
#define ZA_MAX_SEGS   64
#define ZA_WIN        32768
#ifndef ZA_HASH_BITS
#define ZA_HASH_BITS  14
#endif
#define ZA_MIN_MATCH  3
// three link tables: A = chains over 5-byte contexts, B / C = the nearest earlier 3- / 12-byte context
#define ZA_TABLE_A 0
#define ZA_TABLE_B 1
#define ZA_TABLE_C 2
#define ZA_HASH_BYTES_A 5
#define ZA_HASH_BYTES_B 3
#define ZA_HASH_BYTES_C 12
#define ZA_MAX_MATCH  258
#define ZA_DP_SUB     4         // the dynamic programme also tries the 4 next shorter lengths of a position's match
#define ZA_DP_COSTS  260         // u32 per unit: [0..255] literal costs, [256] match base, [257] 0, [258] 1 if a match of the unit is longer than 64, [259] 0
#define ZA_DP_WEAK_DIST 256     // cost statistics: a 3-byte match farther back than this counts as literals
// a `best` entry: distance - 1 (bits 0..14) | length (bits 15..23: 0 = no match, else 3..258) | the position's own byte << 24
#define ZA_ELEN(e)  (((e) >> 15) & 0x1FFu)
#define ZA_EDIST(e) (((e) & 0x7FFFu) + 1u)

#define ZA_FLAG_FINAL 1u
#define ZA_FLAG_FLATHDR 2u      // dynamic header in its flat form (4-bit code lengths at fixed offsets): indexed members
#define ZA_FLAG_CARRY   4u      // (set by the host) the unit's 32 KiB dictionary is the tail of the unit in front of it in the batch:
                                // inside a run the chain tables are carried over instead of inserting the dictionary again
#define ZA_FLAG_SEG2K   16u     // segments of 2 KiB whatever the unit's size (what the segment index of a dict-chained stream counts in: the threaded writer)
#define ZA_FLAG_UNITS16K 32u    // (a block's flag, host only) the block is cut into units of 16 KiB instead of 128: one-shot calls of up to 128 KiB (r06), whose
                                // latency is that of ONE wavefront walking its 64 segments -- eight wavefronts side by side, segments of 256 bytes
#define ZA_SMALL_UNIT  16384u
#define ZA_FLAG_RUNHEAD 8u      // (set by the host) first unit of a chain-kernel run: its dictionary IS inserted, its links are all in its own row
// bits 8..11 of a unit's flags (set by the host, za_seg_shift_for): log2 of the unit's SEGMENT size.  A unit is always cut into at
// most 64 segments (token boundaries are forced there: parse, dynamic programme and packer give a lane to each); a full unit's are
// 2 KiB, the units of small calls get smaller ones -- 32 bytes at least -- so that a call of a few KiB is not ONE lane walking
// 2 048 positions (zlib_ng.compress of 16 KiB: 850 -> 625 us in r05, 377 us with r06's host path; profiles/time_small_calls.py).  Indexed members always use 2 KiB (their index's grain).
#define ZA_UNIT_SEG_SHIFT(flags) ((int)(((flags) >> 8) & 15u))
#define ZA_LIMIT_L     10       // longest literal/length code the encoder emits: one 2^10-entry table decodes every symbol
#define ZA_LIMIT_D     9        // longest distance code
#define ZA_CHUNK_SHIFT 11       // index granularity of indexed members: one entry per 2 KiB segment
#define ZA_MAX_CHUNKS  (ZA_MAX_UNIT >> ZA_CHUNK_SHIFT)

// per-unit workspace strides (elements)
#define ZA_PREV_STRIDE   (ZA_WIN + ZA_MAX_UNIT)   // u16 chain links, index = p + dict_len
#define ZA_BEST_STRIDE   ZA_MAX_UNIT              // u32 len<<16|dist
#define ZA_TOK_STRIDE    ZA_MAX_UNIT              // u32 tokens at segment slots
#define ZA_HIST_STRIDE   320                      // u32: 0..285 lit/len, 288..317 dist
#define ZA_CODE_STRIDE   320                      // u32: code | len<<16, same layout
#define ZA_SEGB_STRIDE   (ZA_MAX_SEGS + 1)        // u32 bit offsets
#define ZA_CIDX_STRIDE   (ZA_MAX_CHUNKS + 4)      // u32 chunk index entries: bit offset | overshoot << 23; [nchunk] = EOB

// unit status bits
#define ZA_ST_OVERFLOW   1u     // compressed output did not fit the slot

__host__ __device__ inline int za_seg_shift_for(uint32_t n, uint32_t flags)
{
    if ((flags & (ZA_FLAG_FLATHDR | ZA_FLAG_SEG2K)) != 0u || n > 65536u) return ZA_SEG_SHIFT;
    int s = 5;
    while ((64u << s) < n) s++;
    return s;
}

struct ZaUnit {
    uint64_t in_off;     // byte offset of the unit's first byte in the input buffer
    uint32_t in_len;     // <= ZA_MAX_UNIT
    uint32_t dict_len;   // <= ZA_WIN bytes readable before in_off
    uint32_t flags;      // ZA_FLAG_FINAL
    uint32_t block;      // index of the reference-level block this unit belongs to
};

// per-unit entropy plan written by the plan kernel, read by the pack kernel
struct ZaPlan {
    uint32_t btype;        // 0 stored, 1 fixed, 2 dynamic
    uint32_t header_bits;  // bits already written to the slot by the plan kernel
    uint32_t pad0, pad1;
};

// chain: steps of the walk over table A; cap: bytes compared per candidate (16 or 258; the winner is extended afterwards);
// use_c: table C's candidate too; dp: the dynamic programme (stage 3a) between search and parse;
// too_far3 / too_far4: a match of 3 / 4 bytes farther back than this is dropped
struct ZaLevel { int chain, nice, max_dist, cap, use_c, dp, too_far3, too_far4; };

typedef uint32_t __attribute__((aligned(1))) za_u32u;
typedef uint64_t __attribute__((aligned(1))) za_u64u;

__device__ __forceinline__ uint32_t za_ld32(const uint8_t *p) { return *(const za_u32u *)p; }
__device__ __forceinline__ uint64_t za_ld64(const uint8_t *p) { return *(const za_u64u *)p; }
typedef uint16_t __attribute__((aligned(1))) za_u16u;
struct __attribute__((packed, aligned(1))) ZaU4u { uint32_t x, y, z, w; };      // 16 bytes at any address
struct __attribute__((packed, aligned(1))) ZaU2u { uint32_t x, y; };
static_assert(alignof(ZaU4u) == 1 && alignof(ZaU2u) == 1, "under-aligned vector types");
__device__ __forceinline__ uint32_t za_ld16(const uint8_t *p) { return *(const za_u16u *)p; }
// the full 32-bit hash of a table's context out of its little-endian dwords w0 = bytes 0..3, w1 = bytes 4..7, w2 = bytes 8..11
// (the bucket is its top ZA_HASH_BITS bits; bytes behind the context do not matter: A masks byte 4 out of w1, B's product has
// its multiplier shifted so that byte 3 cannot reach the top bits)
#define ZA_K1 2654435761u
#define ZA_K2 2246822519u
#define ZA_K3 3266489917u
template <int TABLE> __device__ __forceinline__ uint32_t za_hash_x(uint32_t w0, uint32_t w1, uint32_t w2)
{
    if (TABLE == ZA_TABLE_A) return (w0 * ZA_K1) ^ ((w1 & 0xFFu) * ZA_K2);
    if (TABLE == ZA_TABLE_B) return w0 * (ZA_K1 << 8);
    uint32_t x = (w0 * ZA_K1) ^ (w1 * ZA_K2);
    return ((x ^ (x >> 15)) * ZA_K2) ^ (w2 * ZA_K3);
}
template <int TABLE> struct ZaTableBytes { static constexpr int value = TABLE == ZA_TABLE_A ? ZA_HASH_BYTES_A : TABLE == ZA_TABLE_B ? ZA_HASH_BYTES_B : ZA_HASH_BYTES_C; };

// volatile accesses to LDS with the address space spelled out (ds_read / ds_write instead of flat_load / flat_store)
typedef __attribute__((address_space(3))) volatile uint16_t za_lds_vu16;
typedef __attribute__((address_space(3))) volatile uint32_t za_lds_vu32;
typedef __attribute__((address_space(3))) volatile unsigned long long za_lds_vu64;

__device__ __forceinline__ int za_lane() { return (int)(threadIdx.x & 63); }

// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's global stores (its release fence
// covers every address space: s_waitcnt vmcnt(0) -- on gfx9 stores count there too), which a kernel that hands nothing through
// global memory inside the workgroup does not need
__device__ __forceinline__ void za_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// wave-wide inclusive scan (64 lanes): DPP row shifts inside the rows of 16 lanes, then the two row broadcasts of gfx9 --
// six VALU instructions with a DPP operand instead of six LDS-pipe shuffles
__device__ __forceinline__ uint32_t za_wave_incl_scan(uint32_t v)
{
#ifdef ZA_SCAN_SHFL
    int lane = za_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
#else
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);       // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);       // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);       // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);       // row_shr:8  -> inclusive scan inside every row
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
#endif
}
__device__ __forceinline__ uint32_t za_wave_xor_reduce(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, 64);
    return v;
}

// length symbol (0..28) / extra-bit count / extra value from match length 3..258
__device__ __forceinline__ void za_len_sym(int len, int &code, int &nextra, int &extra)
{
    int l = len - 3;
    if (l < 8) { code = l; nextra = 0; extra = 0; }
    else if (len == 258) { code = 28; nextra = 0; extra = 0; }
    else {
        int nb = 31 - __builtin_clz((unsigned)l);          // floor(log2 l), l >= 8 -> nb >= 3
        code = 4 * (nb - 1) + ((l >> (nb - 2)) & 3);
        nextra = nb - 2;
        extra = l & ((1 << (nb - 2)) - 1);
    }
}
// distance symbol (0..29) from distance 1..32768
__device__ __forceinline__ void za_dist_sym(int dist, int &code, int &nextra, int &extra)
{
    int x = dist - 1;
    if (x < 2) { code = x; nextra = 0; extra = 0; }
    else {
        int nb = 31 - __builtin_clz((unsigned)x);
        code = 2 * nb + ((x >> (nb - 1)) & 1);
        nextra = nb - 1;
        extra = x & ((1 << (nb - 1)) - 1);
    }
}
__device__ __forceinline__ int za_len_extra_bits(int code)   // code 0..28
{
    return (code < 8 || code == 28) ? 0 : ((code - 4) >> 2);
}
__device__ __forceinline__ int za_dist_extra_bits(int code)  // code 0..29
{
    return code < 4 ? 0 : ((code - 2) >> 1);
}

// GF(2) polynomial product modulo the reflected CRC-32 polynomial
__host__ __device__ __forceinline__ uint32_t za_multmodp(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1) ? ((b >> 1) ^ 0xEDB88320u) : (b >> 1);
    }
    return p;
}
