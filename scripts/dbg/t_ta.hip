// Cost of a vector-memory instruction on the CU's memory path as a function of the active lanes:
// one wave issues batches of 48 loads (16-byte, rows of a 4K image), all lanes or the first N lanes only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NW>
__global__ __launch_bounds__(64 * NW) void k(const float *img, float *out, unsigned long long *cyc, int iters, int stride, int nlanes, int do_store)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float *base = img + (size_t) blockIdx.x * stride * 2200 + wv * 1024;
    float *obase = out + (size_t) blockIdx.x * stride * 2200 + wv * 1024;
    f32x4 acc = {0, 0, 0, 0};
    const bool act = lane < nlanes;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        f32x4 q[48];
        if (act) {
#pragma unroll
            for (int r = 0; r < 48; r++) q[r] = *(const f32x4 *) (base + (size_t) ((it * 48 + r) % 2100) * stride + 4 * lane);
#pragma unroll
            for (int r = 0; r < 48; r++) acc += q[r];
            if (do_store) {
#pragma unroll
                for (int r = 0; r < 32; r++) *(f32x4 *) (obase + (size_t) ((it * 32 + r) % 2100) * stride + 4 * lane) = acc;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wv == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] == 12345.f) out[0] = acc[1] + acc[2] + acc[3];
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int stride = 3904, grid = 64;
    float *img, *out; unsigned long long *cyc;
    (void) hipMalloc(&img, (size_t) grid * stride * 2200 * 4); (void) hipMalloc(&out, (size_t) grid * stride * 2200 * 4); (void) hipMalloc(&cyc, grid * 8);
    (void) hipMemset(img, 0, (size_t) grid * stride * 2200 * 4);
    const int iters = 200;
    for (int nw : {1, 4}) for (int st = 0; st < 2; st++) for (int nl : {64, 48, 32, 16, 8}) {
        for (int rep = 0; rep < 2; rep++) {
            if (nw == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, img, out, cyc, iters, stride, nl, st);
            else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, img, out, cyc, iters, stride, nl, st);
            (void) hipDeviceSynchronize();
        }
        unsigned long long c[64];
        (void) hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; i++) s += c[i];
        const int n = st ? 80 : 48;
        printf("%d wave(s) per CU, %s, %2d lanes: %.0f cycles per batch = %.1f per memory instruction of one wave\n", nw, st ? "48 loads + 32 stores" : "48 loads", nl, s / grid / iters, s / grid / iters / n);
    }
    return 0;
}
