// What does one DP row (dp_row4, keep rule on) cost a lone wave when everything is in registers?
// Context: DESIGN.md 4.5 -- the band / tiled kernels measure 365-445 cycles per row for ~58 instructions.
#include "../../gimp-lqr-plugin_amd/csrc/lqr_hip.hip"
#include <cstdio>

template <int VARIANT>
__global__ __launch_bounds__(64) void k_rows(const float *src, float *out, unsigned long long *cyc, int iters, int store)
{
    const int lane = threadIdx.x;
    constexpr int R = 16;
    f32x4 q_e[R], q_mo[R];
    uint32_t q_lo[R];
    for (int r = 0; r < R; r++) {
        q_e[r] = *(const f32x4 *) (src + (r * 64 + lane) * 4);
        q_mo[r] = *(const f32x4 *) (src + 8192 + (r * 64 + lane) * 4);
        q_lo[r] = __float_as_uint(src[16384 + r * 64 + lane]) & 0x01ff01ffu;
    }
    float mp[4] = {src[lane], src[lane + 64], src[lane + 128], src[lane + 192]};
    const bool in[4] = {true, true, true, true};
    const bool own = lane >= 4 && lane < 60 && store;
    float *o = out + (size_t) blockIdx.x * (VARIANT >= 6 ? 4194304 + 65536 : 65536);
    __shared__ f32x4 s_rows[R][64];
    __shared__ uint32_t s_l[R][64];
    __shared__ int s_cnt;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        unsigned so = lane * 4;
#pragma unroll
        for (int r = 0; r < R; r++, so += 256) {
            asm volatile("" : "+v"(so));
            float mc[4];
            uint32_t lnew = 0;
            bool ch[4];
            const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[3]), DPP_WAVE_SHR1, 0xf, 0xf, true));
            const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
            dp_row4<false, false, true, false>(mp, left, right, q_e[r], q_mo[r], q_lo[r], in, 0.f, 0.f, mc, lnew, ch);
            if (VARIANT == 1) {
                if (own) {
                    *(f32x4 *) (o + so) = f32x4{mc[0], mc[1], mc[2], mc[3]};
                    *(uint32_t *) (o + 32768 + (so >> 2)) = lnew;
                }
            } else if (VARIANT == 2) {
                if (own) *(f32x4 *) (o + so) = f32x4{mc[0], mc[1], mc[2], mc[3]};
                acc ^= lnew;
            } else if (VARIANT == 3) {
                if (own) *(uint32_t *) (o + 32768 + (so >> 2)) = lnew;
            } else if (VARIANT == 4) {
                // hand the row to another wave through LDS instead
                s_rows[r][lane] = f32x4{mc[0], mc[1], mc[2], mc[3]};
                s_l[r][lane] = lnew;
                if (lane == 0) *(volatile int *) &s_cnt = r + 1;
            } else if (VARIANT == 6 || VARIANT == 7) {
                float *os = o + (VARIANT == 7 ? 1 : 0) + (size_t) (it & 255) * 16384;
                *(f32x4 *) (os + so) = f32x4{mc[0], mc[1], mc[2], mc[3]};
                *(uint32_t *) (os + 8192 + (so >> 2)) = lnew;
            } else if (VARIANT == 5) {
                // unconditional stores (halo lanes to a scratch line): no exec juggling
                *(f32x4 *) (o + so) = f32x4{mc[0], mc[1], mc[2], mc[3]};
                *(uint32_t *) (o + 32768 + (so >> 2)) = lnew;
            } else {
                acc ^= lnew;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) mp[k] = mc[k];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    o[60000 + lane] = mp[0] + mp[1] + mp[2] + mp[3] + acc;
}


// two waves as in the band kernels: wave 1 computes rows and stores them with a 4K image's row stride, wave 0 plays the
// partner: 0 exits, 1 spins on LDS with s_sleep, 2 prefetches 48 rows' worth of loads every ~4000 cycles
template <int PARTNER>
__global__ __launch_bounds__(128) void k_rows2(const float *src, float *big, unsigned long long *cyc, int iters, int stride)
{
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    constexpr int R = 16;
    __shared__ volatile int s_done;
    if (threadIdx.x == 0) s_done = 0;
    __syncthreads();
    float *img = big + (size_t) blockIdx.x * ((size_t) stride * 2800);
    if (q == 0) {
        if (PARTNER == 0) return;
        float acc = 0;
        int y = 0;
        while (!s_done) {
            if (PARTNER == 2) {
                f32x4 a[R], b[R]; float c[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const size_t ro = (size_t) ((y + r) % 2100) * stride + 1024 + 4 * lane;
                    a[r] = *(const f32x4 *) (img + ro); b[r] = *(const f32x4 *) (img + ro + 512); c[r] = img[ro / 4 + 2000];
                }
#pragma unroll
                for (int r = 0; r < R; r++) acc += a[r][0] + b[r][1] + c[r];
                y += R;
                for (int i = 0; i < 30; i++) __builtin_amdgcn_s_sleep(2);
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (acc == 12345.f) big[0] = acc;
        return;
    }
    f32x4 q_e[R], q_mo[R];
    uint32_t q_lo[R];
    for (int r = 0; r < R; r++) {
        q_e[r] = *(const f32x4 *) (src + (r * 64 + lane) * 4);
        q_mo[r] = *(const f32x4 *) (src + 8192 + (r * 64 + lane) * 4);
        q_lo[r] = __float_as_uint(src[16384 + r * 64 + lane]) & 0x01ff01ffu;
    }
    float mp[4] = {src[lane], src[lane + 64], src[lane + 128], src[lane + 192]};
    const bool in[4] = {true, true, true, true};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        unsigned so = (unsigned) ((it * R) % 2100) * (unsigned) stride + 1 + 4 * lane, so4 = so * 4u;
#pragma unroll
        for (int r = 0; r < R; r++, so += (unsigned) stride, so4 += 4u * (unsigned) stride) {
            asm volatile("" : "+v"(so), "+v"(so4));
            float mc[4];
            uint32_t lnew = 0;
            bool ch[4];
            const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[3]), DPP_WAVE_SHR1, 0xf, 0xf, true));
            const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
            dp_row4<false, false, true, false>(mp, left, right, q_e[r], q_mo[r], q_lo[r], in, 0.f, 0.f, mc, lnew, ch);
            *(f32x4 *) ((char *) img + so4) = f32x4{mc[0], mc[1], mc[2], mc[3]};
            *(uint32_t *) ((char *) img + (size_t) stride * 2200 * 4 + so) = lnew;
#pragma unroll
            for (int k = 0; k < 4; k++) mp[k] = mc[k];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[blockIdx.x] = t1 - t0; s_done = 1; }
    img[60000 + lane] = mp[0] + mp[1] + mp[2] + mp[3];
}

// several waves of one workgroup (one CU, different SIMDs) walking rows at the same time, as two active slots of the band
// kernel do: do they slow each other down?
template <int NC>
__global__ __launch_bounds__(64 * NC) void k_rowsN(const float *src, float *big, unsigned long long *cyc, int iters, int stride)
{
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    constexpr int R = 16;
    float *img = big + (size_t) (blockIdx.x * NC + q) * ((size_t) stride * 2800);
    f32x4 q_e[R], q_mo[R];
    uint32_t q_lo[R];
    for (int r = 0; r < R; r++) {
        q_e[r] = *(const f32x4 *) (src + (r * 64 + lane) * 4);
        q_mo[r] = *(const f32x4 *) (src + 8192 + (r * 64 + lane) * 4);
        q_lo[r] = __float_as_uint(src[16384 + r * 64 + lane]) & 0x01ff01ffu;
    }
    float mp[4] = {src[lane], src[lane + 64], src[lane + 128], src[lane + 192]};
    const bool in[4] = {true, true, true, true};
    const bool own = lane >= 4 && lane < 60;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        unsigned so = (unsigned) ((it * R) % 2100) * (unsigned) stride + 1 + 4 * lane, so4 = so * 4u;
#pragma unroll
        for (int r = 0; r < R; r++, so += (unsigned) stride, so4 += 4u * (unsigned) stride) {
            asm volatile("" : "+v"(so), "+v"(so4));
            float mc[4];
            uint32_t lnew = 0;
            bool ch[4];
            const float left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[3]), DPP_WAVE_SHR1, 0xf, 0xf, true));
            const float right = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mp[0]), DPP_WAVE_SHL1, 0xf, 0xf, true));
            dp_row4<false, false, true, false>(mp, left, right, q_e[r], q_mo[r], q_lo[r], in, 0.f, 0.f, mc, lnew, ch);
            if (own) {
                *(f32x4 *) ((char *) img + so4) = f32x4{mc[0], mc[1], mc[2], mc[3]};
                *(uint32_t *) ((char *) img + (size_t) stride * 2200 * 4 + so) = lnew;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) mp[k] = mc[k];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && q == 0) cyc[blockIdx.x] = t1 - t0;
    img[60000 + lane] = mp[0] + mp[1] + mp[2] + mp[3];
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *out; unsigned long long *cyc;
    (void) hipMalloc(&src, 32768 * 4); (void) hipMalloc(&out, (size_t) 256 * (4194304 + 65536) * 4); (void) hipMalloc(&cyc, 256 * 8);
    float h[32768];
    for (int i = 0; i < 32768; i++) h[i] = (float) ((i * 2654435761u) >> 8) * 1e-6f;
    (void) hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int grid : {1, 64, 256}) {
        for (int var = 0; var < 8; var++) {
            for (int rep = 0; rep < 2; rep++) {
                switch (var) {
                case 0: hipLaunchKernelGGL(k_rows<0>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 1: hipLaunchKernelGGL(k_rows<1>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 2: hipLaunchKernelGGL(k_rows<2>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 3: hipLaunchKernelGGL(k_rows<3>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 4: hipLaunchKernelGGL(k_rows<4>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 6: hipLaunchKernelGGL(k_rows<6>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 7: hipLaunchKernelGGL(k_rows<7>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                default: hipLaunchKernelGGL(k_rows<5>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                }
                (void) hipDeviceSynchronize();
            }
            unsigned long long c[256];
            (void) hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < grid; i++) s += c[i];
            printf("grid %3d %-22s: %.1f cycles per row\n", grid, (const char *[]){"registers only", "both stores", "dwordx4 store only", "dword store only", "LDS hand-off", "unconditional stores", "uncond., streaming", "uncond., streaming, +4 B"}[var], s / grid / (iters * 16.0));
        }
    }
    {
        const int stride = 3904, grid = 64;
        float *big; (void) hipMalloc(&big, (size_t) grid * stride * 2800 * 4);
        (void) hipMemset(big, 0, (size_t) grid * stride * 2800 * 4);
        for (int var = 0; var < 3; var++) {
            for (int rep = 0; rep < 2; rep++) {
                if (var == 0) hipLaunchKernelGGL(k_rows2<0>, dim3(grid), dim3(128), 0, 0, src, big, cyc, 500, stride);
                else if (var == 1) hipLaunchKernelGGL(k_rows2<1>, dim3(grid), dim3(128), 0, 0, src, big, cyc, 500, stride);
                else hipLaunchKernelGGL(k_rows2<2>, dim3(grid), dim3(128), 0, 0, src, big, cyc, 500, stride);
                (void) hipDeviceSynchronize();
            }
            unsigned long long c[256];
            (void) hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
            double sum = 0; for (int i = 0; i < grid; i++) sum += c[i];
            printf("2 waves, 4K stride, partner %s: %.1f cycles per row\n", (const char *[]){"exits", "spins", "prefetches"}[var], sum / grid / (500 * 16.0));
        }
    }
    {
        const int stride = 3904, grid = 32;
        float *big3; (void) hipMalloc(&big3, (size_t) grid * 4 * stride * 2800 * 4);
        for (int nc : {1, 2, 4}) {
            for (int rep = 0; rep < 2; rep++) {
                if (nc == 1) hipLaunchKernelGGL(k_rowsN<1>, dim3(grid), dim3(64), 0, 0, src, big3, cyc, 500, stride);
                else if (nc == 2) hipLaunchKernelGGL(k_rowsN<2>, dim3(grid), dim3(128), 0, 0, src, big3, cyc, 500, stride);
                else hipLaunchKernelGGL(k_rowsN<4>, dim3(grid), dim3(256), 0, 0, src, big3, cyc, 500, stride);
                (void) hipDeviceSynchronize();
            }
            unsigned long long c[256];
            (void) hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
            double sum = 0; for (int i = 0; i < grid; i++) sum += c[i];
            printf("%d wave(s) of one workgroup walking rows at once (conditional stores): %.1f cycles per row\n", nc, sum / grid / (500 * 16.0));
        }
    }
    return 0;
}
