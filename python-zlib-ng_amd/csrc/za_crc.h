// Wave-cooperative CRC-32 and Adler-32 over one <=128 KiB span: one lane per 2 KiB segment, partial
// results folded with GF(2) products (CRC) or prefix sums (Adler).  Product code.
// Replaces zng_crc32_z / zng_adler32 (reference call sites zlib_ngmodule.c:1741, :2556, :1487).
#pragma once
#include "za_common.h"

// All 64 lanes call with wave-uniform arguments.  crct: 256-entry table in LDS; x8k[k] = x^(8*2048*k).
__device__ __forceinline__ uint32_t za_wave_crc32(const uint8_t *data, int n, const uint32_t *crct,
                                                  const uint32_t *__restrict__ x8k)
{
    const int lane = za_lane();
    const int nseg = (n + ZA_SEG - 1) >> ZA_SEG_SHIFT;
    const int s0 = lane << ZA_SEG_SHIFT;
    int s1 = s0 + ZA_SEG;
    if (s1 > n) s1 = n;
    uint32_t c = 0;
    if (lane < nseg) {
        uint32_t r = 0xFFFFFFFFu;
        int p = s0;
        // byte steps until 4-byte aligned, then a dword per load
        for (; p < s1 && (((uintptr_t)(data + p)) & 3u); p++) r = crct[(r ^ data[p]) & 0xFF] ^ (r >> 8);
        for (; p + 4 <= s1; p += 4) {
            r ^= *(const uint32_t *)(data + p);
            r = crct[r & 0xFF] ^ (r >> 8);
            r = crct[r & 0xFF] ^ (r >> 8);
            r = crct[r & 0xFF] ^ (r >> 8);
            r = crct[r & 0xFF] ^ (r >> 8);
        }
        for (; p < s1; p++) r = crct[(r ^ data[p]) & 0xFF] ^ (r >> 8);
        c = r ^ 0xFFFFFFFFu;
        if (lane < nseg - 1) {
            // crc(A||B) = crc(A) * x^(8|B|) ^ crc(B);  |B| = (nseg-2-lane) full segments + the tail
            const int tail = n - ((nseg - 1) << ZA_SEG_SHIFT);
            uint32_t xt = 0x80000000u, sq = 0x00800000u;       // x^0, x^8
            for (int m = tail; m; m >>= 1) { if (m & 1) xt = za_multmodp(sq, xt); sq = za_multmodp(sq, sq); }
            c = za_multmodp(za_multmodp(x8k[nseg - 2 - lane], xt), c);
        }
    }
    return za_wave_xor_reduce(c);
}

// The same with a slice-by-4 table (crct4: 4 x 256 entries in LDS, table t = table t-1 advanced by one byte) and 16-byte loads: four
// look-ups per dword that do not wait for one another, a quarter of the load instructions (the lanes' segments lie 2 KiB apart: every
// load instruction of the wave touches 64 lines, and with 4 bytes per lane each line was touched 32 times).  za_k_checksum.
__device__ __forceinline__ uint32_t za_wave_crc32_s4(const uint8_t *data, int n, const uint32_t *crct4,
                                                     const uint32_t *__restrict__ x8k)
{
    const int lane = za_lane();
    const int nseg = (n + ZA_SEG - 1) >> ZA_SEG_SHIFT;
    const int s0 = lane << ZA_SEG_SHIFT;
    int s1 = s0 + ZA_SEG;
    if (s1 > n) s1 = n;
    uint32_t c = 0;
    if (lane < nseg) {
        uint32_t r = 0xFFFFFFFFu;
        int p = s0;
        auto dword = [&](uint32_t v) {
            r ^= v;
            r = crct4[768 + (r & 0xFFu)] ^ crct4[512 + ((r >> 8) & 0xFFu)] ^ crct4[256 + ((r >> 16) & 0xFFu)] ^ crct4[r >> 24];
        };
        for (; p < s1 && (((uintptr_t)(data + p)) & 15u); p++) r = crct4[(r ^ data[p]) & 0xFF] ^ (r >> 8);
        for (; p + 16 <= s1; p += 16) {
            const uint4 v = *(const uint4 *)(data + p);
            dword(v.x); dword(v.y); dword(v.z); dword(v.w);
        }
        for (; p < s1; p++) r = crct4[(r ^ data[p]) & 0xFF] ^ (r >> 8);
        c = r ^ 0xFFFFFFFFu;
        if (lane < nseg - 1) {
            const int tail = n - ((nseg - 1) << ZA_SEG_SHIFT);
            uint32_t xt = 0x80000000u, sq = 0x00800000u;       // x^0, x^8
            for (int m = tail; m; m >>= 1) { if (m & 1) xt = za_multmodp(sq, xt); sq = za_multmodp(sq, sq); }
            c = za_multmodp(za_multmodp(x8k[nseg - 2 - lane], xt), c);
        }
    }
    return za_wave_xor_reduce(c);
}

// Adler-32 partial of data[0..n) from a zero state: returns (a, b) sums mod 65521 in lane-uniform
// registers; fold with  B' = B + n*A + b,  A' = A + a.
__device__ __forceinline__ void za_wave_adler(const uint8_t *data, int n, uint32_t &a_out, uint32_t &b_out)
{
    const int lane = za_lane();
    const int s0 = lane << ZA_SEG_SHIFT;
    int s1 = s0 + ZA_SEG;
    if (s1 > n) s1 = n;
    uint32_t a = 0, b = 0;
    const int len = s1 > s0 ? s1 - s0 : 0;
    for (int i = 0; i < len; i++) { const uint32_t d = data[s0 + i]; a += d; b += (uint32_t)(len - i) * d; }
    // exclusive prefix of a over lanes
    const uint32_t incl = za_wave_incl_scan(a);
    const unsigned long long contrib = (unsigned long long)b + (unsigned long long)len * (unsigned long long)(incl - a);
    unsigned long long bs = contrib % 65521ull;
    for (int d = 32; d >= 1; d >>= 1) bs += __shfl_xor(bs, d, 64);
    a_out = __shfl(incl, 63, 64) % 65521u;
    b_out = (uint32_t)(bs % 65521ull);
}
