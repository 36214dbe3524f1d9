// How fast does one wave issue dependent VALU instructions, as a function of how busy the chip is?
// (are the single-wave chains of the DP kernels running at the full engine clock?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void chain(float *out, int iters)
{
    float a = threadIdx.x * 1e-9f, b = 1.000001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = __fmaf_rn(a, b, 1e-7f);
    }
    if (a == 12345.f) out[0] = a;
}
__global__ void chain_s(int *out, int iters)
{
    int x = blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_add_i32 %0, %0, 3\n\ts_lshr_b32 %0, %0, 1" : "+s"(x));
    }
    if (x == 12345) out[0] = x;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *d; (void) hipMalloc(&d, 4096);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int iters = 4000;            // 256 K dependent FMAs
    int grids[] = {1, 30, 256, 1024, 4096};
    for (int rep = 0; rep < 2; rep++)
    for (int g : grids) {
        hipLaunchKernelGGL(chain, dim3(g), dim3(64), 0, 0, d, 100);
        (void) hipDeviceSynchronize();
        (void) hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain, dim3(g), dim3(64), 0, 0, d, iters);
        (void) hipEventRecord(e1, 0); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        printf("valu grid %5d waves: %.3f ms  -> %.2f ns per dependent v_fma\n", g, ms, ms * 1e6 / (iters * 64.0));
        (void) hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain_s, dim3(g), dim3(64), 0, 0, (int *) d, iters);
        (void) hipEventRecord(e1, 0); (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
        printf("salu grid %5d waves: %.3f ms  -> %.2f ns per dependent s_op\n", g, ms, ms * 1e6 / (iters * 128.0));
    }
    return 0;
}
