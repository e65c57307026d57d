// Checksum kernels over arbitrary device buffers (product code).
// Replaces zng_crc32 / zng_adler32 behind zlib_ng.crc32 / adler32 (reference
// src/zlib_ng/zlib_ngmodule.c:1455-1562): one wavefront per 128 KiB span, one lane per 2 KiB
// segment, per-span partials folded on the host with crc32_combine / the Adler recurrence.
#include "za_common.h"
#include "za_crc.h"

struct ZaCkPart { uint32_t crc, a, b, len; };

__global__ __launch_bounds__(64) void za_k_checksum(const uint8_t *__restrict__ buf, uint64_t n,
                                                    const uint32_t *__restrict__ crc_table,
                                                    const uint32_t *__restrict__ x8k_table,
                                                    ZaCkPart *__restrict__ parts, int want_adler)
{
    __shared__ uint32_t crct[1024];               // slice-by-4 (crc_table: the context's 4 x 256 table)
    const int lane = za_lane();
    for (int i = lane; i < 1024; i += 64) crct[i] = crc_table[i];
    __syncthreads();
    const uint64_t off = (uint64_t)blockIdx.x * ZA_MAX_UNIT;
    const int len = (int)((n - off) > ZA_MAX_UNIT ? ZA_MAX_UNIT : (n - off));
    const uint32_t c = za_wave_crc32_s4(buf + off, len, crct, x8k_table);
    uint32_t a = 0, b = 0;
    if (want_adler) za_wave_adler(buf + off, len, a, b);
    if (lane == 0) { ZaCkPart p; p.crc = c; p.a = a; p.b = b; p.len = (uint32_t)len; parts[blockIdx.x] = p; }
}
