// Which compute units does a stream created with hipExtStreamCreateWithCUMask use?  (round 6: one masked stream per sub-batch)
// For each mask: a grid of 4096 short workgroups records (XCC_ID, HW_ID); the host prints the number of distinct CUs per XCD.
//   hipcc --offload-arch=gfx950 -O2 scripts/dbg/t_cumask.hip -o /tmp/t_cumask && /tmp/t_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <map>
#include <vector>
__global__ void k(unsigned *out)
{
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(20);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = id; out[2 * blockIdx.x + 1] = xcc; }
}
static void probe(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t s;
    if (mask.empty()) { if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { printf("%s: stream failed\n", name); return; } }
    else if (hipExtStreamCreateWithCUMask(&s, (uint32_t) mask.size(), mask.data()) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); (void) hipGetLastError(); return; }
    const int n = 4096;
    unsigned *d;
    std::vector<unsigned> h(2 * n);
    (void) hipMalloc(&d, h.size() * 4);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipEventRecord(e0, s);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, s, d);
    (void) hipEventRecord(e1, s);
    (void) hipStreamSynchronize(s);
    float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
    (void) hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < n; b++) per_xcc[h[2 * b + 1] & 15].insert((h[2 * b] >> 8) & 0xff);      // cu_id[11:8], sh[12], se[15:13]
    printf("%-28s %.3f ms; CUs per XCC:", name, ms);
    int total = 0;
    for (auto &kv : per_xcc) { printf(" x%u:%zu", kv.first, kv.second.size()); total += (int) kv.second.size(); }
    printf("  total %d\n", total);
    // is block b still on XCC b mod (number of XCCs in use)?
    int agree8 = 0; for (int b = 0; b < n; b++) agree8 += ((h[2 * b + 1] & 15) == (unsigned) (b & 7));
    printf("    blocks with xcc == b mod 8: %d of %d\n", agree8, n);
    (void) hipFree(d); (void) hipStreamDestroy(s);
}
int main()
{
    probe("no mask", {});
    probe("bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
    probe("bits 64..127", {0, 0, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    probe("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    probe("every 4th bit (k=0)", std::vector<uint32_t>(8, 0x11111111u));
    probe("every 4th bit (k=1)", std::vector<uint32_t>(8, 0x22222222u));
    probe("bytes 0,4,.. (8 of 32)", std::vector<uint32_t>(8, 0x000000ffu));
    probe("bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    probe("all 256", std::vector<uint32_t>(8, 0xffffffffu));
    return 0;
}
