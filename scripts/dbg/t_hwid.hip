// Which SIMD does each wave of a workgroup land on?  (HW_ID: wave_id[3:0], simd_id[5:4], pipe[7:6], cu_id[11:8], sh, se, ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned *d, h[16 * 8];
    (void) hipMalloc(&d, sizeof h);
    for (int threads : {128, 256, 512, 1024}) {
        (void) hipMemset(d, 0, sizeof h);
        hipLaunchKernelGGL(k, dim3(8), dim3(threads), 0, 0, d);
        (void) hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int b = 0; b < 3; b++) {
            printf("%4d threads, workgroup %d: simd of waves 0..%d:", threads, b, threads / 64 - 1);
            for (int w = 0; w < threads / 64; w++) printf(" %u", (h[b * 16 + w] >> 4) & 3);
            printf("   (cu %u, se %u)\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
        }
    }
    return 0;
}
