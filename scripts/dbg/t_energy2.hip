#include "../../gimp-lqr-plugin_amd/csrc/lqr_hip.hip"
int main(int argc, char** argv) {
  const int w = 8, h = 4, stride = 64;
  uint32_t hp[5*64] = {0};
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) hp[y*stride+x] = (10*x + 7*y + 40) | ((20*x+3*y+5) << 8) | ((x*x+y+100) << 16) | (255u << 24);
  uint32_t* dp; float* den; float ho[4*64]; DevCarver d; memset(&d, 0, sizeof d); DevCarver* dd;
  hipMalloc(&dp, sizeof hp); hipMalloc(&den, sizeof ho); hipMalloc(&dd, sizeof d);
  hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice);
  d.pix = dp; d.en = den; hipMemcpy(dd, &d, sizeof d, hipMemcpyHostToDevice);
  for (int nrg = 0; nrg < 7; nrg++) {
    DpK k; memset(&k, 0, sizeof k); k.delta = 1; k.nrg = nrg; k.radius = 1; k.w_start = w; k.ch = 4;
    #define L(N) hipLaunchKernelGGL((k_emap_full<N>), dim3(1, h, 1), dim3(256), 0, 0, dd, k, w, h, stride)
    NRG_DISPATCH(nrg, L)
#undef L
    hipDeviceSynchronize();
    hipMemcpy(ho, den, sizeof ho, hipMemcpyDeviceToHost);
    printf("nrg %d:", nrg); for (int x = 0; x < w; x++) printf(" %.6g", ho[1*stride+x]); printf("\n");
  }
  return 0;
}
