// Micro-benchmark (not product code): how many 64-thread workgroups with 32 KiB (or 16 / 8 KiB) of LDS run at once per CU,
// and what one such wave issues per cycle alone and with company.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int LDSW>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters)
{
    __shared__ uint32_t t[LDSW];
    t[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t a = threadIdx.x, b = blockIdx.x, c = 7, d = 9;
    for (int i = 0; i < iters; i++) {
        a = a * 2654435761u + b; b = (b ^ a) >> 3; c += a ^ b; d = (d << 1) ^ c;
        a += d; b ^= c; c = c * 3u + 1u; d ^= a >> 7;
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d ^ t[(a & 63)];
}
template <int LDSW> void run(uint32_t *out, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%s:", name);
    for (int per_cu = 1; per_cu <= 12; per_cu++) {
        const int blocks = 256 * per_cu;
        hipLaunchKernelGGL(k<LDSW>, dim3(blocks), dim3(64), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<LDSW>, dim3(blocks), dim3(64), 0, 0, out, 100000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %d/CU %.2f ms;", per_cu, ms);
    }
    printf("\n");
}
int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    uint32_t *out; hipMalloc(&out, 256 * 12 * 64 * 4);
    run<8192>(out, "32 KiB LDS");
    run<4096>(out, "16 KiB LDS");
    run<2048>(out, " 8 KiB LDS");
    run<64>(out, "256 B LDS");
    return 0;
}
