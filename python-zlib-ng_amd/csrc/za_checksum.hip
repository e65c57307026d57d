// Checksum kernels over arbitrary device buffers (product code).
// Replaces zng_crc32 / zng_adler32 behind zlib_ng.crc32 / adler32 (reference
// src/zlib_ng/zlib_ngmodule.c:1455-1562) and the containers' trailers of the one-shot calls: one 256-thread workgroup per
// 128 KiB span, a thread per S = 64 .. 512 consecutive bytes (the smallest of these with 256 S >= the span: a call of 1 KiB keeps
// 16 threads busy for 64 bytes each, not one lane for all of it -- until r06 a lane took 2 KiB, byte by byte for the Adler sums, and
// both checksums were worked out whichever was asked for: 56 us for 1 KiB, 135 us for 64 KiB, a quarter of a small call).
// Per-span partials are folded on the host with crc32_combine / the Adler recurrence.
#include "za_common.h"
#include "za_crc.h"

struct ZaCkPart { uint32_t crc, a, b, len; };

// tabs (zngamd_ctx::d_crc_slice4): [0, 1024) CRC-32 slice-by-4; [1280 + 256 j + k] = x^(8 * (64 << j) * k) mod P for j < 4, k < 256;
// [2304 + t] = x^(8 t) mod P for t <= 512
#define ZA_CK_XS 1280
#define ZA_CK_XT 2304
#define ZA_CK_TABS 2820
__global__ __launch_bounds__(256) void za_k_checksum(const uint8_t *__restrict__ buf, uint64_t n, const uint32_t *__restrict__ tabs,
                                                     ZaCkPart *__restrict__ parts, int want_crc, int want_adler,
                                                     const uint64_t *__restrict__ n_dev = nullptr)     // (optional) the buffer's length where a kernel in front left it: `n` is then its bound
{
    __shared__ uint32_t crct[1024];
    __shared__ uint32_t red_c[4], red_a[4];
    __shared__ unsigned long long red_b[4];
    const int tid = (int)threadIdx.x, lane = za_lane(), wave = tid >> 6;
    const uint64_t off = (uint64_t)blockIdx.x * ZA_MAX_UNIT;
    if (n_dev) n = *n_dev < n ? *n_dev : n;
    if (off >= n) {                                                  // (uniform) a span behind the data's end: nothing to add
        if (tid == 0) { ZaCkPart pt; pt.crc = 0; pt.a = 0; pt.b = 0; pt.len = 0; parts[blockIdx.x] = pt; }
        return;
    }
    const int len = (int)((n - off) > ZA_MAX_UNIT ? ZA_MAX_UNIT : (n - off));
    int sh = 6;
    while ((256 << sh) < len) sh++;                                  // 6 .. 9
    const int s0 = tid << sh;
    const int s1 = s0 + (1 << sh) < len ? s0 + (1 << sh) : len;
    if (want_crc) {
        for (int i = tid; i < 1024; i += 256) crct[i] = tabs[i];
        __syncthreads();
    }
    uint32_t c = 0, a = 0;
    unsigned long long bc = 0;
    if (s0 < len) {
        const uint8_t *p = buf + off + s0, *e = buf + off + s1;
        uint32_t r = 0xFFFFFFFFu, b = 0;
        // Adler: a = the bytes' sum, b = the sum of the running a (the first byte counts once per byte of the segment)
        auto one = [&](uint32_t d) { if (want_crc) r = crct[(r ^ d) & 0xFFu] ^ (r >> 8); a += d; b += a; };
        auto dword = [&](uint32_t v) {
            if (want_crc) {
                r ^= v;
                r = crct[768 + (r & 0xFFu)] ^ crct[512 + ((r >> 8) & 0xFFu)] ^ crct[256 + ((r >> 16) & 0xFFu)] ^ crct[r >> 24];
            }
            if (want_adler) {
                b += 4u * a + __builtin_amdgcn_udot4(v, 0x01020304u, 0u, false);      // the dword's first byte is its lowest
                a = __builtin_amdgcn_udot4(v, 0x01010101u, a, false);
            }
        };
        for (; p < e && (((uintptr_t)p) & 15u); p++) one(*p);
        for (; p + 16 <= e; p += 16) {
            const uint4 v = *(const uint4 *)p;
            dword(v.x); dword(v.y); dword(v.z); dword(v.w);
        }
        for (; p < e; p++) one(*p);
        const uint32_t after = (uint32_t)(len - s1);                 // bytes of the span behind my segment
        if (want_crc) {
            c = r ^ 0xFFFFFFFFu;
            // crc(A || B) = crc(A) * x^(8 |B|) + crc(B):  |B| = q whole segments and t bytes
            if (after) c = za_multmodp(za_multmodp(tabs[ZA_CK_XS + 256 * (sh - 6) + (int)(after >> sh)], tabs[ZA_CK_XT + (int)(after & ((1u << sh) - 1u))]), c);
        }
        bc = ((unsigned long long)b + (unsigned long long)after * a) % 65521ull;
    }
    c = za_wave_xor_reduce(c);
    for (int d = 32; d >= 1; d >>= 1) { bc += __shfl_xor(bc, d, 64); a += __shfl_xor(a, d, 64); }
    if (lane == 0) { red_c[wave] = c; red_a[wave] = a; red_b[wave] = bc; }
    __syncthreads();
    if (tid == 0) {
        ZaCkPart pt;
        pt.crc = red_c[0] ^ red_c[1] ^ red_c[2] ^ red_c[3];
        pt.a = (uint32_t)(((unsigned long long)red_a[0] + red_a[1] + red_a[2] + red_a[3]) % 65521ull);
        pt.b = (uint32_t)((red_b[0] + red_b[1] + red_b[2] + red_b[3]) % 65521ull);
        pt.len = (uint32_t)len;
        parts[blockIdx.x] = pt;
    }
}
