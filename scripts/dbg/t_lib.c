#include <stdio.h>
#include <stdlib.h>
#include "../../include/lqr.h"
int main(int argc, char** argv) {
  int w = 8, h = 4, ef = argc > 1 ? atoi(argv[1]) : 2;
  unsigned char* buf = malloc(w*h*4);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { unsigned char* p = buf + (y*w+x)*4; p[0] = 10*x+7*y+40; p[1] = 20*x+3*y+5; p[2] = x*x+y+100; p[3] = 255; }
  LqrCarver* c = lqr_carver_new(buf, w, h, 4);
  if (!c) return 1;
  lqr_carver_init(c, 1, 0);
  lqr_carver_set_energy_function_builtin(c, ef);
  float en[32];
  int r = lqrx_carver_get_energy(c, en);
  printf("ret %d\n", r);
  for (int y = 0; y < h; y++) { for (int x = 0; x < w; x++) printf(" %.6g", en[y*w+x]); printf("\n"); }
  lqr_carver_destroy(c);
  return 0;
}
