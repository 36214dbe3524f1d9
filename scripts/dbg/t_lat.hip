// micro-benchmark: latency of a dependent global load that walks an image plane row by row
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_lat(const float* __restrict__ p, int stride, int h, int x, long long* out, float* sink) {
  float acc = 0; int xx = x + threadIdx.x;
  long long t0 = __builtin_readcyclecounter();   // s_memtime
  for (int y = 0; y < h; y++) { float v = p[(size_t)y*stride + xx + (int)(acc*0.0f)]; acc += v; }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; sink[blockIdx.x*64+threadIdx.x] = acc;
}
int main() {
  const int stride = 3904, h = 2160; size_t n = (size_t)stride*(h+1);
  float* d[8]; long long* out; float* sink; hipMalloc(&out, 64*8); hipMalloc(&sink, 8*64*4*8);
  for (int i = 0; i < 8; i++) { hipMalloc(&d[i], n*4); hipMemset(d[i], 0, n*4); }
  for (int rep = 0; rep < 3; rep++) {
    // touch everything with a streaming kernel equivalent: memset again to evict
    for (int i = 0; i < 8; i++) hipMemset(d[i], 0, n*4);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d[0], stride, h, 1000, out, sink);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("rep %d: cold-ish dependent row walk: %.1f cycles/load (s_memtime ticks @100MHz? raw %lld)\n", rep, (double)c/h, c);
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d[0], stride, h, 1000, out, sink);
    hipDeviceSynchronize();
    hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("rep %d: warm (just read): %.1f ticks/load\n", rep, (double)c/h);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 8; i++) hipMemset(d[i], 0, n*4);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d[1], stride, h, 2000, out, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("event time cold walk: %.3f ms -> %.3f us/load\n", ms, ms*1e3/h);
  hipEventRecord(e0); hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d[1], stride, h, 2000, out, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); printf("event time warm walk: %.3f ms -> %.3f us/load\n", ms, ms*1e3/h);
  return 0;
}
