// Host cost of feeding S streams round-robin with small dependent chains (5 kernels per round, like a seam round), as plain
// launches and as one captured graph per stream per R rounds.   hipcc --offload-arch=gfx950 -O2 t_launch.hip -o t_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { int v[40]; };     // a by-value argument block like DpK
__global__ void spin(Big b, int *out, long long cycles)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && out) out[blockIdx.x] = b.v[3];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int rounds = 200;
    int *d; CK(hipMalloc(&d, 4096));
    Big b{};
    for (int S : {1, 4, 8, 16}) {
        std::vector<hipStream_t> st(S);
        for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int us : {0, 20, 100}) {                       // kernel duration: 0 = launch-bound, 100 us x 5 = a 0.5 ms round
            const long long cyc = (long long) us * 100;      // wall_clock64: 100 MHz
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int r = 0; r < rounds; r++)
                for (int s = 0; s < S; s++)
                    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st[s], b, d, cyc);
            double t_issue = now() - t0;
            CK(hipDeviceSynchronize());
            double t_all = now() - t0;
            printf("plain  S=%2d kernel %3d us: issue %.2f us per launch, wall %.1f us per round (ideal %d)\n", S, us, t_issue * 1e6 / (rounds * S * 5), t_all * 1e6 / rounds, 5 * us);
        }
        // graphs: 20 rounds (100 kernel nodes) per graph per stream
        for (int us : {0, 20, 100}) {
            const long long cyc = (long long) us * 100;
            const int R = 20;
            std::vector<hipGraphExec_t> ex(S);
            double tc = now();
            for (int s = 0; s < S; s++) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
                for (int r = 0; r < R; r++) for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin, dim3(16), dim3(64), 0, st[s], b, d, cyc);
                CK(hipStreamEndCapture(st[s], &g));
                CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
                CK(hipGraphDestroy(g));
            }
            tc = now() - tc;
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int rep = 0; rep < rounds / R; rep++)
                for (int s = 0; s < S; s++) CK(hipGraphLaunch(ex[s], st[s]));
            double t_issue = now() - t0;
            CK(hipDeviceSynchronize());
            double t_all = now() - t0;
            printf("graph  S=%2d kernel %3d us: capture+instantiate %.1f us per node, issue %.2f us per node, wall %.1f us per round (ideal %d)\n", S, us,
                   tc * 1e6 / (S * R * 5), t_issue * 1e6 / (rounds * S * 5), t_all * 1e6 / rounds, 5 * us);
            for (size_t i = 0; i < ex.size(); i++) CK(hipGraphExecDestroy(ex[i]));
        }
        for (auto &s : st) CK(hipStreamDestroy(s));
    }
    return 0;
}
