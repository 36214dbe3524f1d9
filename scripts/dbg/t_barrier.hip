// micro-benchmark: cost of one "LDS write -> s_barrier -> LDS read" round per iteration, vs number of waves
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NW> __global__ __launch_bounds__(64*NW) void k_bar(int iters, long long* out, int* sink) {
  __shared__ int s[2][16];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; int acc = 0;
  if (threadIdx.x < 32) ((int*)s)[threadIdx.x] = 0;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (lane == 0) s[i & 1][wave] = acc + i;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    acc += s[i & 1][(wave + 1) % NW];
    acc = __builtin_amdgcn_readfirstlane(acc);
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = acc;
}
__global__ void k_lds(int iters, long long* out, int* sink) {   // dependent LDS read chain, single wave
  __shared__ int s[64]; s[threadIdx.x] = (threadIdx.x + 1) & 63; __syncthreads();
  int p = 0; long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) { p = s[p]; p = __builtin_amdgcn_readfirstlane(p); }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0; sink[threadIdx.x] = p;
}
int main() {
  long long* out; int* sink; hipMalloc(&out, 64); hipMalloc(&sink, 4096); long long c; const int it = 4000;
#define RUN(NW) hipLaunchKernelGGL(k_bar<NW>, dim3(1), dim3(64*NW), 0, 0, it, out, sink); hipDeviceSynchronize(); hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); printf("NW=%d: %.1f cycles per write-barrier-read round\n", NW, (double)c/it);
  RUN(1) RUN(2) RUN(4) RUN(8) RUN(16)
  hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, it, out, sink); hipDeviceSynchronize(); hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); printf("dependent LDS read + readfirstlane: %.1f cycles\n", (double)c/it);
  return 0;
}
