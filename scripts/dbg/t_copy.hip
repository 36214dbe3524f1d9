// micro-benchmark: what streaming-copy form gets closest to the HBM ceiling on this device (read + write GB/s).
// Variants: loads in flight per thread (U), workgroups, non-temporal vs plain.  Used to set lqrhip_copy_bandwidth's
// kernel (the "measured_copy_peak" of the bench line) and to judge how far k_carve is from the real ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *src, u32x4 *dst, size_t n16)
{
    const GLOBAL_AS u32x4 *s = (const GLOBAL_AS u32x4 *) src;
    GLOBAL_AS u32x4 *d = (GLOBAL_AS u32x4 *) dst;
    const size_t stride = (size_t) gridDim.x * 256;
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(s + i + k * stride) : s[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(v[k], d + i + k * stride); else d[i + k * stride] = v[k]; }
    }
    for (; i < n16; i += stride) d[i] = s[i];
}
template <int U, bool NT>
void run(const char *name, u32x4 *a, u32x4 *b, size_t n16, int grid)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, n16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-6s U=%d grid=%5d: %.3f ms  %.0f GB/s\n", name, U, grid, best, 2.0 * n16 * 16 / (best * 1e-3) / 1e9);
}
int main()
{
    const size_t bytes = (size_t) 2 << 30, n16 = bytes / 16;
    u32x4 *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<1, true>("nt", a, b, n16, grid);
        run<4, true>("nt", a, b, n16, grid);
        run<8, true>("nt", a, b, n16, grid);
        run<1, false>("plain", a, b, n16, grid);
        run<4, false>("plain", a, b, n16, grid);
    }
    // one shot: every thread exactly one 16-byte element
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_copy<1, true>), dim3((unsigned) (n16 / 256)), dim3(256), 0, 0, a, b, n16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("one element per thread (grid %zu): %.3f ms  %.0f GB/s\n", n16 / 256, best, 2.0 * n16 * 16 / (best * 1e-3) / 1e9);
    }
    return 0;
}
