// Where do the waves of a grid of 128-thread workgroups with a large register footprint land?  (round 6: k_band_levels runs 2 waves of
// ~240 VGPRs per workgroup; 448 workgroups on 256 CUs.)  Every workgroup stays resident until all have started (a counter), records
// (XCC, SE, CU, SIMD) of its two waves; the host counts workgroups per CU and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 scripts/dbg/t_place.hip -o /tmp/t_place && /tmp/t_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
template <int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(unsigned *out, int *ctr, int n)
{
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("v_mov_b32 v230, 0" ::: "v230");          // claim the register footprint of the level kernel
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (threadIdx.x == 0) atomicAdd(ctr, 1);
    for (int i = 0; i < 20000 && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n; i++) __builtin_amdgcn_s_sleep(20);
}
template <int THREADS>
static void probe(int n)
{
    unsigned *d; int *ctr;
    const int wpw = THREADS / 64;
    std::vector<unsigned> h((size_t) 2 * n * wpw);
    (void) hipMalloc(&d, h.size() * 4); (void) hipMalloc(&ctr, 4); (void) hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(k<THREADS>, dim3(n), dim3(THREADS), 0, 0, d, ctr, n);
    (void) hipDeviceSynchronize();
    (void) hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned>, int> wg_per_cu;
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, int> waves_per_simd;
    for (int b = 0; b < n; b++)
        for (int w = 0; w < wpw; w++) {
            const unsigned id = h[(size_t) (b * wpw + w) * 2], xcc = h[(size_t) (b * wpw + w) * 2 + 1] & 15;
            const unsigned cu = (id >> 8) & 15, se = (id >> 13) & 7, sh = (id >> 12) & 1, simd = (id >> 4) & 3;
            if (w == 0) wg_per_cu[{xcc, se * 2 + sh, cu}]++;
            waves_per_simd[{xcc, se * 2 + sh, cu, simd}]++;
        }
    int hist_cu[9] = {0}, hist_simd[9] = {0};
    for (auto &kv : wg_per_cu) hist_cu[kv.second < 8 ? kv.second : 8]++;
    for (auto &kv : waves_per_simd) hist_simd[kv.second < 8 ? kv.second : 8]++;
    printf("%4d workgroups of %d threads: CUs used %zu; CUs with 1/2/3/4 workgroups: %d %d %d %d; SIMDs used %zu; SIMDs with 1/2/3/4 waves: %d %d %d %d\n",
           n, THREADS, wg_per_cu.size(), hist_cu[1], hist_cu[2], hist_cu[3], hist_cu[4], waves_per_simd.size(), hist_simd[1], hist_simd[2], hist_simd[3], hist_simd[4]);
    printf("     first workgroups (xcc se cu: simd of wave 0, 1 ..):");
    for (int b = 0; b < 20; b++) { const unsigned id = h[(size_t) (b * wpw) * 2]; printf(" [%u %u %u:", h[(size_t) (b * wpw) * 2 + 1] & 15, (id >> 13) & 7, (id >> 8) & 15); for (int w = 0; w < wpw; w++) printf("%u", (h[(size_t) (b * wpw + w) * 2] >> 4) & 3); printf("]"); }
    printf("\n");
    (void) hipFree(d); (void) hipFree(ctr);
}
int main()
{
    for (int n : {56, 112, 224, 448, 512}) probe<128>(n);
    for (int n : {56, 112, 224, 256}) probe<256>(n);
    return 0;
}
