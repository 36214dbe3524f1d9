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
    float *o = out + (size_t) blockIdx.x * 65536;
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
    out[(size_t) blockIdx.x * 65536 + 60000 + lane] = mp[0] + mp[1] + mp[2] + mp[3] + acc;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *out; unsigned long long *cyc;
    (void) hipMalloc(&src, 32768 * 4); (void) hipMalloc(&out, (size_t) 256 * 65536 * 4); (void) hipMalloc(&cyc, 256 * 8);
    float h[32768];
    for (int i = 0; i < 32768; i++) h[i] = (float) ((i * 2654435761u) >> 8) * 1e-6f;
    (void) hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int grid : {1, 256}) {
        for (int var = 0; var < 6; var++) {
            for (int rep = 0; rep < 2; rep++) {
                switch (var) {
                case 0: hipLaunchKernelGGL(k_rows<0>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 1: hipLaunchKernelGGL(k_rows<1>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 2: hipLaunchKernelGGL(k_rows<2>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 3: hipLaunchKernelGGL(k_rows<3>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                case 4: hipLaunchKernelGGL(k_rows<4>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                default: hipLaunchKernelGGL(k_rows<5>, dim3(grid), dim3(64), 0, 0, src, out, cyc, iters, 1); break;
                }
                (void) hipDeviceSynchronize();
            }
            unsigned long long c[256];
            (void) hipMemcpy(c, cyc, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < grid; i++) s += c[i];
            printf("grid %3d %-22s: %.1f cycles per row\n", grid, (const char *[]){"registers only", "both stores", "dwordx4 store only", "dword store only", "LDS hand-off", "unconditional stores"}[var], s / grid / (iters * 16.0));
        }
    }
    return 0;
}
