// micro-benchmark: cost of vector accesses that are only element-aligned (gfx950).
//  (a) streaming copy with 16-B loads + 16-B stores at byte offset 4*k from a 16-B aligned row, k = 0..3
//  (b) same with 4-B loads/stores of a byte plane at byte offset k = 0..3 (unaligned dwords)
//  (c) correctness of both (compared with a host copy)
// This decides whether the carved planes may start at a per-image element offset.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))

__global__ __launch_bounds__(256) void k_copy16(const unsigned char *src, unsigned char *dst, int off_bytes, size_t n16)
{
    const GLOBAL_AS u32x4 *s = (const GLOBAL_AS u32x4 *) (src + off_bytes);
    GLOBAL_AS u32x4 *d = (GLOBAL_AS u32x4 *) (dst + off_bytes);
    const size_t stride = (size_t) gridDim.x * 256;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        u32x4 v = __builtin_nontemporal_load(s + i);
        __builtin_nontemporal_store(v, d + i);
    }
}
__global__ __launch_bounds__(256) void k_copy4(const unsigned char *src, unsigned char *dst, int off_bytes, size_t n4)
{
    const GLOBAL_AS unsigned int *s = (const GLOBAL_AS unsigned int *) (src + off_bytes);
    GLOBAL_AS unsigned int *d = (GLOBAL_AS unsigned int *) (dst + off_bytes);
    const size_t stride = (size_t) gridDim.x * 256;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d[i] = s[i];
}
// row-walk flavour: one wave per row, 16 B per lane, like the DP kernels' loads (latency-bound chain of rows)
__global__ __launch_bounds__(64) void k_rows(const float *src, float *sink, int stride, int h, int off_elems)
{
    const GLOBAL_AS float *p = (const GLOBAL_AS float *) src + off_elems + 4 * threadIdx.x + 1024 * blockIdx.x;
    float acc = 0;
    for (int y = 0; y < h; y++) {
        const GLOBAL_AS u32x4 *q = (const GLOBAL_AS u32x4 *) (p + (size_t) y * stride);
        u32x4 v = *q;
        acc += __uint_as_float(v.x) + __uint_as_float(v.w);
    }
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    const size_t bytes = (size_t) 1 << 30;
    unsigned char *a, *b;
    hipMalloc(&a, bytes + 64); hipMalloc(&b, bytes + 64);
    std::vector<unsigned char> h(1 << 20), out(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char) (i * 2654435761u >> 13);
    hipMemset(a, 0, bytes + 64);
    hipMemcpy(a, h.data(), h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int k = 0; k < 4; k++) {
            const int off = mode == 0 ? 4 * k : k;
            hipMemset(b, 0xee, bytes + 64);
            const size_t n = mode == 0 ? (bytes - 64) / 16 : (bytes / 4 - 64) / 4;      // the byte plane is a quarter of the size
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_copy16, dim3(8192), dim3(256), 0, 0, a, b, off, n);
                else hipLaunchKernelGGL(k_copy4, dim3(8192), dim3(256), 0, 0, a, b, off, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            hipMemcpy(out.data(), b, out.size(), hipMemcpyDeviceToHost);
            const size_t nb = mode == 0 ? 16 : 4;
            bool ok = memcmp(out.data() + off, h.data() + off, out.size() - 64) == 0;
            for (int i = 0; i < off; i++) ok &= out[i] == 0xee;
            printf("%s offset %d B: %.3f ms  %.1f GB/s (r+w)  %s\n", mode == 0 ? "dwordx4" : "dword  ", off, best,
                   2.0 * n * nb / (best * 1e-3) / 1e9, ok ? "data ok" : "DATA WRONG");
        }
    }
    // dependent-ish row walk (no data dependence, but one load per row per lane: issue + latency), warm L2
    const int stride = 3904, hh = 2160;
    float *sink; hipMalloc(&sink, 64 * 64 * 4);
    for (int k = 0; k < 4; k++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rows, dim3(3), dim3(64), 0, 0, (const float *) a, sink, stride, hh, k);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("row walk, 16-B loads at element offset %d: %.1f us for %d rows\n", k, best * 1e3, hh);
    }
    return 0;
}
