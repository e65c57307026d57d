// Micro-benchmark (not product code): issue rate of the integer VALU / LDS instructions the deflate and inflate kernels are
// made of, on gfx950, by resident waves per SIMD.  Answers: is a wave64 integer VALU op 2 or 4 cycles of its SIMD, which ops are
// slower (v_mul_lo_u32, 64-bit shifts, v_perm, DPP), what do random ds_read_b32 / ds_read2 cost.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_issue profiles/ubench_issue.hip && build/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define UNROLL16(OPSTR) \
    asm volatile(OPSTR : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a8) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a9) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a10) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a11) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a12) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a13) : "v"(b), "v"(c) : "vcc", "s20"); \
    asm volatile(OPSTR : "+v"(a14) : "v"(b), "v"(c) : "vcc", "s20"); asm volatile(OPSTR : "+v"(a15) : "v"(b), "v"(c) : "vcc", "s20");

#define KERNEL(NAME, OPSTR) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters) { \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t a8 = a0 + 8, a9 = a0 + 9, a10 = a0 + 10, a11 = a0 + 11, a12 = a0 + 12, a13 = a0 + 13, a14 = a0 + 14, a15 = a0 + 15; \
    uint32_t b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x & 3u; \
    for (int i = 0; i < iters; i++) { UNROLL16(OPSTR) UNROLL16(OPSTR) } \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ a8 ^ a9 ^ a10 ^ a11 ^ a12 ^ a13 ^ a14 ^ a15; \
}

KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_addclamp, "v_add_u32_e64 %0, %0, %1 clamp")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, %2")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_mul, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, %1, %2")
KERNEL(k_lshl, "v_lshlrev_b32 %0, %2, %0")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, %2, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL(k_snop, "s_nop 0")
KERNEL(k_sadd, "s_add_u32 s20, s20, 1")

// 64-bit shift (two-register operands)
__global__ __launch_bounds__(256) void k_lshl64(uint32_t *out, int iters)
{
    uint64_t a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x + j;
    uint32_t c = threadIdx.x & 3u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[j]) : "v"(c));
    }
    uint64_t s = 0;
    for (int j = 0; j < 8; j++) s ^= a[j];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// LDS reads: `mode` 0 = ds_read_b32 at random dword addresses, 1 = ds_read2_b32 (w, w+1) random, 2 = ds_read_b32 linear (conflict free),
// 3 = ds_read_b64 random (8-byte aligned), 4 = 5 dwords random base (b32 x5 as the compiler would: read2 + read2 + read)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(uint32_t *out, int iters)
{
    __shared__ uint32_t lds[4096];        // 16 KiB
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t w = MODE == 2 ? ((threadIdx.x + 64u * j + i) & 4095u) : (x >> 20);      // 12 bits
            if (MODE == 0 || MODE == 2) acc += lds[w];
            else if (MODE == 1) { const uint32_t w2 = w & 4094u; acc += lds[w2] ^ lds[w2 + 1]; }
            else if (MODE == 3) { const uint64_t v = *(const uint64_t *)&lds[w & 4094u]; acc += (uint32_t)v ^ (uint32_t)(v >> 32); }
            else { const uint32_t w2 = w & 4088u; acc += lds[w2] ^ lds[w2 + 1] ^ lds[w2 + 2] ^ lds[w2 + 3] ^ lds[w2 + 4]; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kern_t)(uint32_t *, int);

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    printf("start\n");
    uint32_t *out;
    const int maxblocks = 256 * 8;
    CHK(hipMalloc(&out, (size_t)maxblocks * 256 * 4));
    printf("allocated\n");
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct K { const char *name; kern_t f; double per_iter; };      // per_iter: wave-instructions of the measured kind per iteration
    std::vector<K> ks = {
        {"v_xor_b32", k_xor, 32}, {"v_add_u32", k_add, 32}, {"v_add_u32 clamp", k_addclamp, 32}, {"v_alignbyte", k_alignbyte, 32},
        {"v_alignbit", k_alignbit, 32}, {"v_ffbl", k_ffbl, 32}, {"v_min3_u32", k_min3, 32}, {"v_perm_b32", k_perm, 32},
        {"v_mul_lo_u32", k_mul, 32}, {"v_mul_u32_u24", k_mul24, 32}, {"v_mad_u32_u24", k_mad24, 32}, {"v_bfe_u32", k_bfe, 32},
        {"v_lshlrev_b32", k_lshl, 32}, {"v_lshl_add_u32", k_lshladd, 32}, {"v_and_or_b32", k_and_or, 32}, {"v_cndmask", k_cndmask, 32},
        {"v_mbcnt_lo", k_mbcnt, 32}, {"v_add dpp row_shr", k_dpp, 32}, {"v_bcnt", k_bcnt, 32}, {"v_cmp", k_cmp, 32},
        {"v_readlane", k_readlane, 32}, {"s_nop", k_snop, 32}, {"s_add_u32", k_sadd, 32}, {"v_lshlrev_b64", k_lshl64, 32},
        {"ds_read_b32 random", k_lds<0>, 8}, {"ds_read2_b32 random", k_lds<1>, 8}, {"ds_read_b32 linear", k_lds<2>, 8},
        {"ds_read_b64 random", k_lds<3>, 8}, {"5 dwords random base", k_lds<4>, 8},
    };
    printf("%-22s", "instruction");
    for (int w = 1; w <= 8; w *= 2) printf("  %d w/SIMD: cyc/instr/SIMD", w);
    printf("   (cycles at 2.4 GHz per wave-instruction of one SIMD; LDS rows: per CU)\n");
    for (auto &k : ks) {
        printf("%-22s", k.name);
        for (int w = 1; w <= 8; w *= 2) {
            const bool lds = std::string(k.name).find("ds_") == 0 || std::string(k.name).find("5 dw") == 0;
                        const int blocks = 256 * w;                                      // 256-thread blocks: one wave per SIMD each
            const int iters = lds ? 4096 : 16384;
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 64);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            // per SIMD: w waves x iters x per_iter instructions in ms
            const double instr_per_simd = (double)w * iters * k.per_iter * (lds ? 4.0 : 1.0);   // LDS: per CU (4 SIMDs share it)
            const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
            printf("  %24.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
