// LDS-DMA (global_load_lds_dwordx4) on gfx950: layout, unaligned global addresses, and what a wave pays to ISSUE
// vector-memory instructions on cold lines (registers vs LDS-DMA; 1, 2, 4 waves of a workgroup issuing at once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gv;
typedef __attribute__((address_space(3))) void lv;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float *src, float *dst, int off)
{
    __shared__ __attribute__((aligned(16))) float s[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) s[i] = -1.0f;
    __syncthreads();
    // rows of 32 lanes: lanes 0-31 read src[off + 4 * lane ..], lanes 32-63 read a second "row" 1000 floats further
    const float *g = src + off + (lane < 32 ? 4 * lane : 1000 + 4 * (lane - 32));
    __builtin_amdgcn_global_load_lds((gv *) g, (lv *) s, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gv *) (g + 2000), (lv *) (s + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) dst[i] = s[i];
}

// MODE 0: 48 dwordx4 loads into registers; 1: 48 LDS-DMA loads.  Every workgroup walks its own never-touched region
// (stride rows of a "4K image"), NW waves of the workgroup do the same on different columns.
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_issue(const float *img, float *out, unsigned long long *cyc, int iters, int stride)
{
    extern __shared__ __attribute__((aligned(16))) float s_buf[];        // NW * 48 KB? no: NW * 16 rows * 1 KB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float *base = img + (size_t) blockIdx.x * stride * 2200 + wv * 512 + 4 * lane;
    float *lds = s_buf + wv * 16 * 256;
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t_issue = 0, t_land = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) {
            f32x4 q[48];
#pragma unroll
            for (int r = 0; r < 48; r++) q[r] = *(const f32x4 *) (base + (size_t) ((it * 48 + r) % 2100) * stride);
            const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
            for (int r = 0; r < 48; r++) acc += q[r];
            asm volatile("" :: "v"(acc));
            const unsigned long long t2 = __builtin_readcyclecounter();
            t_issue += t1 - t0; t_land += t2 - t1;
        } else {
#pragma unroll
            for (int r = 0; r < 48; r++)
                __builtin_amdgcn_global_load_lds((gv *) (base + (size_t) ((it * 48 + r) % 2100) * stride), (lv *) (lds + (r & 15) * 256), 16, 0, 0);
            const unsigned long long t1 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t2 = __builtin_readcyclecounter();
            acc[0] += lds[lane];
            t_issue += t1 - t0; t_land += t2 - t1;
        }
    }
    if (lane == 0 && wv == 0) { cyc[2 * blockIdx.x] = t_issue; cyc[2 * blockIdx.x + 1] = t_land; }
    if (acc[0] == 12345.f) out[0] = acc[1] + acc[2] + acc[3];
}


// what the band kernels issue per batch: 16 rows x {16 B of plane A, 16 B of plane B, 4 B of plane C}, optionally off 16-B
// alignment (the origin shift), on the first `nl` lanes only, optionally behind 32 stores
__global__ __launch_bounds__(256) void k_bandlike(const float *img, float *out, unsigned long long *cyc, int iters, int stride, int mis, int nl, int st, int nwaves)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv >= nwaves) return;
    const size_t plane = (size_t) stride * 2200;
    const float *base = img + (size_t) blockIdx.x * plane * 3 + wv * 512 + 4 * lane + mis;
    float *obase = out + (size_t) blockIdx.x * plane * 3 + wv * 512 + 4 * lane + mis;
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t_issue = 0, t_land = 0, t_st = 0;
    for (int it = 0; it < iters; it++) {
        f32x4 a[16], b[16]; float c[16];
        unsigned long long t0 = __builtin_readcyclecounter();
        if (st && lane < nl) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const size_t ro = (size_t) ((it * 16 + r) % 2100) * stride;
                *(f32x4 *) (obase + ro) = acc;
                *(float *) (obase + plane * 2 + (ro >> 2)) = acc[0];
            }
        }
        const unsigned long long ts = __builtin_readcyclecounter();
        t_st += ts - t0; t0 = ts;
        if (lane < nl) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const size_t ro = (size_t) ((it * 16 + r + 32) % 2100) * stride;
                a[r] = *(const f32x4 *) (base + ro);
                b[r] = *(const f32x4 *) (base + plane + ro);
                c[r] = *(const float *) (base + plane * 2 + (ro >> 2));
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane < nl) {
#pragma unroll
            for (int r = 0; r < 16; r++) { acc += a[r]; acc += b[r]; acc[0] += c[r]; }
        }
        asm volatile("" :: "v"(acc));
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0; t_land += t2 - t1;
    }
    if (lane == 0 && wv == 0) { cyc[3 * blockIdx.x] = t_issue; cyc[3 * blockIdx.x + 1] = t_land; cyc[3 * blockIdx.x + 2] = t_st; }
    if (acc[0] == 12345.f) out[0] = acc[1] + acc[2] + acc[3];
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *dst;
    (void) hipMalloc(&src, 8192 * 4); (void) hipMalloc(&dst, 512 * 4);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (float) i;
    (void) hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int off : {0, 1, 3}) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, src, dst, off);
        float r[512];
        (void) hipMemcpy(r, dst, sizeof r, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 512; i++) {
            const int half = i / 256, j = i % 256;
            const float want = (float) (off + (j < 128 ? j : 1000 + (j - 128)) + 2000 * half);
            if (r[i] != want) { if (bad < 4) printf("  off %d: lds[%d] = %g, want %g\n", off, i, r[i], want); bad++; }
        }
        printf("LDS-DMA dwordx4, global address %d floats off 16-B alignment: %s (%d wrong)\n", off, bad ? "WRONG" : "layout = lane * 16 B, data right", bad);
    }
    const int stride = 3904, grid = 64;
    float *img, *out; unsigned long long *cyc;
    (void) hipMalloc(&img, (size_t) grid * stride * 2200 * 4); (void) hipMalloc(&out, 4096); (void) hipMalloc(&cyc, grid * 16);
    (void) hipMemset(img, 0, (size_t) grid * stride * 2200 * 4);
    const int iters = 40;
    for (int mode = 0; mode < 2; mode++) for (int nw : {1, 2, 4}) {
        for (int rep = 0; rep < 2; rep++) {
            const size_t lds = (size_t) nw * 16 * 1024;
#define L(M, N) hipLaunchKernelGGL((k_issue<M, N>), dim3(grid), dim3(64 * N), lds, 0, img, out, cyc, iters, stride)
            if (mode == 0) { if (nw == 1) L(0, 1); else if (nw == 2) L(0, 2); else L(0, 4); }
            else { if (nw == 1) L(1, 1); else if (nw == 2) L(1, 2); else L(1, 4); }
            (void) hipDeviceSynchronize();
        }
        unsigned long long c[128];
        (void) hipMemcpy(c, cyc, grid * 16, hipMemcpyDeviceToHost);
        double si = 0, sl = 0; for (int i = 0; i < grid; i++) { si += c[2 * i]; sl += c[2 * i + 1]; }
        printf("%s, %d wave(s) per workgroup: issue %.0f cycles per load instruction, then %.0f cycles until the batch of 48 landed\n",
               mode ? "LDS-DMA  " : "registers", nw, si / grid / iters / 48, sl / grid / iters);
    }

    {
        float *img3, *out3;
        (void) hipMalloc(&img3, (size_t) grid * stride * 2200 * 4 * 3); (void) hipMalloc(&out3, (size_t) grid * stride * 2200 * 4 * 3);
        (void) hipMemset(img3, 0, (size_t) grid * stride * 2200 * 4 * 3);
        unsigned long long *cyc3; (void) hipMalloc(&cyc3, grid * 24);
        for (int nwv : {1, 2}) for (int st : {0, 1}) for (int mis : {0, 1}) for (int nl : {64, 40}) {
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k_bandlike, dim3(grid), dim3(256), 0, 0, img3, out3, cyc3, 60, stride, mis, nl, st, nwv);
                (void) hipDeviceSynchronize();
            }
            unsigned long long c3[192];
            (void) hipMemcpy(c3, cyc3, grid * 24, hipMemcpyDeviceToHost);
            double si = 0, sl = 0, ss = 0; for (int i = 0; i < grid; i++) { si += c3[3 * i]; sl += c3[3 * i + 1]; ss += c3[3 * i + 2]; }
            printf("band-like batch, %d wave(s), %s, %s, %d lanes: 48 loads issue %.0f cycles (%.0f each), landed after %.0f more; 32 stores %.0f (%.0f each)\n", nwv,
                   st ? "behind 32 stores" : "no stores", mis ? "4-B aligned" : "16-B aligned", nl, si / grid / 60, si / grid / 60 / 48, sl / grid / 60, ss / grid / 60, ss / grid / 60 / 32);
        }
    }
    return 0;
}
