// How fast does one wave issue dependent VALU instructions, as a function of how many waves share
// its CU / how busy the chip is?  (context: the DP kernels are single-wave dependency chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void chain(float *out, int iters, int active_mask)
{
    const int wave = threadIdx.x >> 6;
    if (!((active_mask >> wave) & 1)) return;
    float a = threadIdx.x * 1e-9f, b = 1.000001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = __fmaf_rn(a, b, 1e-7f);
    }
    if (a == 12345.f) out[0] = a;
}
// a mix like the DP row: v_cmp -> sgpr -> s_or -> v_cndmask
__global__ void chain_mix(float *out, int iters, int active_mask)
{
    const int wave = threadIdx.x >> 6;
    if (!((active_mask >> wave) & 1)) return;
    float a = threadIdx.x * 1e-3f, b = 0.5f, c = 0.25f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float m = fminf(fminf(a, b), c);
            const bool p = (a == m), q = (fabsf(a - b) > 1e-5f);
            a = (p || q) ? a + 1.0f : b;
            b = b + c;
        }
    }
    if (a == 12345.f) out[0] = a + b;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *d; (void) hipMalloc(&d, 4096);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int iters = 4000;
    struct { int grid, threads, mask; const char *what; } cfg[] = {
        {1, 64, 1, "1 WG x 1 wave"}, {1, 256, 0xf, "1 WG x 4 waves"}, {1, 512, 0xff, "1 WG x 8 waves"}, {1, 512, 0x0f, "1 WG x 8 waves, 0-3 active"},
        {1, 512, 0x55, "1 WG x 8 waves, even active"}, {64, 512, 0x0f, "64 WG x 8 waves, 0-3 active"}, {30, 128, 1, "30 WG x 2 waves, 1 active"},
        {256, 64, 1, "256 WG x 1 wave"}, {1024, 64, 1, "1024 WG x 1 wave"}, {4096, 64, 1, "4096 WG x 1 wave"}};
    for (auto &c : cfg) {
        for (int mix = 0; mix < 2; mix++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                (void) hipEventRecord(e0, 0);
                if (mix) hipLaunchKernelGGL(chain_mix, dim3(c.grid), dim3(c.threads), 0, 0, d, iters, c.mask);
                else hipLaunchKernelGGL(chain, dim3(c.grid), dim3(c.threads), 0, 0, d, iters, c.mask);
                (void) hipEventRecord(e1, 0); (void) hipEventSynchronize(e1);
                (void) hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-32s %s: %.3f ms -> %.2f ns per %s\n", c.what, mix ? "mix" : "fma", ms, ms * 1e6 / (iters * (mix ? 16.0 : 64.0)), mix ? "mix step (~9 instr)" : "dependent v_fma");
        }
    }
    return 0;
}
