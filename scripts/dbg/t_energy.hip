#include "../../gimp-lqr-plugin_amd/csrc/lqr_hip.hip"
__global__ void k_dbg(const uint32_t* pix, int stride, int w, int h, int ch, float* out) {
  int x = threadIdx.x, y = blockIdx.x;
  if (x >= w) return;
  for (int nrg = 0; nrg < 7; nrg++) out[(nrg*h + y)*w + x] = grad_energy(pix, stride, x, y, w, h, ch, nrg);
  if (y == 1 && x == 2) {
     double b0 = px_bright(pix[y*stride+1], ch, false), b1 = px_bright(pix[y*stride+3], ch, false);
     printf("b(1)=%.17g b(3)=%.17g half-diff=%.17g\n", b0, b1, (b1-b0)*0.5);
  }
}
int main() {
  const int w = 6, h = 3, stride = 64;
  uint32_t hp[3*64] = {0};
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) hp[y*stride+x] = (10*x + 7*y + 40) | ((20*x+3*y+5) << 8) | ((x*x+y+100) << 16) | (255u << 24);
  uint32_t* dp; float* dout; float ho[7*3*6];
  hipMalloc(&dp, sizeof hp); hipMalloc(&dout, sizeof ho);
  hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dbg, dim3(h), dim3(64), 0, 0, dp, stride, w, h, 4, dout);
  hipDeviceSynchronize();
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  for (int nrg = 0; nrg < 7; nrg++) { printf("nrg %d:", nrg); for (int x = 0; x < w; x++) printf(" %.6g", ho[(nrg*h+1)*w+x]); printf("\n"); }
  return 0;
}
