/* A plain C99 host of libwvn_hip.so: what a C / C++ maintainer links instead of the Python layer.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -o c_host -Lwild_visual_navigation_amd/lib -lwvn_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/wild_visual_navigation_amd/lib -Wl,-rpath,/opt/rocm/lib
 *
 * It runs SimpleMLP.forward (wild_visual_navigation/model/simple_mlp.py:10-39 -> wvn_mlp_forward) on 96 rows of 90-d
 * features with parameters it fills itself and compares with the same three layers computed in this file.  Device memory comes
 * from the HIP runtime's C entry points (declared below: no HIP header, no C++).  Without a GPU it prints the sizes the library
 * reports and exits 0 -- the C-ABI links and answers -- which is what the CPU test-suite checks. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "wvn_hip.h"

/* HIP runtime, C linkage (hipError_t is an int-sized enum, 0 = success; 1 = hipMemcpyHostToDevice, 2 = DeviceToHost) */
int hipGetDeviceCount(int* count);
int hipMalloc(void** ptr, size_t bytes);
int hipFree(void* ptr);
int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
int hipDeviceSynchronize(void);

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(void) {
  const wvn_mlp_desc d = {90, 256, 32, 0};
  const int R = 96, O = 1 + d.D;
  const size_t np = wvn_mlp_param_count(&d), ws_bytes = wvn_mlp_workspace_bytes(&d, R);
  printf("libwvn_hip version %d; SimpleMLP(90,[256,32,1]) parameters %zu, workspace for %d rows %zu bytes; wire image of a "
         "448x448 frame with 100 segments %zu bytes\n", wvn_version(), np, R, ws_bytes, wvn_wire_bytes(448, 448, 100, 90));
  if (np != (size_t)(256 * 90 + 256 + 32 * 256 + 32 + 91 * 32 + 91)) { printf("FAIL: parameter count\n"); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != 0 || ndev == 0) { printf("no GPU: the C-ABI links and answers\n"); return 0; }

  unsigned seed = 7;
  float* p = (float*)malloc(np * sizeof(float));
  float* x = (float*)malloc((size_t)R * d.D * sizeof(float));
  float* out = (float*)malloc((size_t)R * O * sizeof(float));
  for (size_t i = 0; i < np; ++i) p[i] = 0.2f * frand(&seed);
  for (int i = 0; i < R * d.D; ++i) x[i] = 2.f * frand(&seed);

  void *dp, *dx, *dout, *dws;
  if (hipMalloc(&dp, np * 4) || hipMalloc(&dx, (size_t)R * d.D * 4) || hipMalloc(&dout, (size_t)R * O * 4) || hipMalloc(&dws, ws_bytes)) {
    printf("FAIL: hipMalloc\n");
    return 1;
  }
  hipMemcpy(dp, p, np * 4, 1);
  hipMemcpy(dx, x, (size_t)R * d.D * 4, 1);
  const int rc = wvn_mlp_forward(&d, (const float*)dp, (const float*)dx, d.D, R, (float*)dout, NULL, NULL, dws, ws_bytes, NULL);
  if (rc != 0 || hipDeviceSynchronize() != 0) { printf("FAIL: wvn_mlp_forward rc %d\n", rc); return 1; }
  hipMemcpy(out, dout, (size_t)R * O * 4, 2);

  /* the same network here: flat layout [W1 | b1 | W2 | b2 | W3 | b3], Linear layout [out][in] */
  const float *W1 = p, *b1 = W1 + 256 * 90, *W2 = b1 + 256, *b2 = W2 + 32 * 256, *W3 = b2 + 32, *b3 = W3 + 91 * 32;
  double worst = 0.0;
  for (int r = 0; r < R; ++r) {
    float h1[256], h2[32];
    for (int j = 0; j < 256; ++j) { double a = b1[j]; for (int k = 0; k < 90; ++k) a += (double)W1[j * 90 + k] * x[r * 90 + k]; h1[j] = a > 0 ? (float)a : 0.f; }
    for (int j = 0; j < 32; ++j) { double a = b2[j]; for (int k = 0; k < 256; ++k) a += (double)W2[j * 256 + k] * h1[k]; h2[j] = a > 0 ? (float)a : 0.f; }
    for (int j = 0; j < O; ++j) {
      double a = b3[j];
      for (int k = 0; k < 32; ++k) a += (double)W3[j * 32 + k] * h2[k];
      if (j == 0) a = 1.0 / (1.0 + exp(-a));
      const double e = fabs(a - (double)out[r * O + j]);
      if (e > worst) worst = e;
    }
  }
  printf("wvn_mlp_forward vs the C reference on %d rows: max |diff| = %.3g\n", R, worst);
  hipFree(dp); hipFree(dx); hipFree(dout); hipFree(dws);
  free(p); free(x); free(out);
  if (!(worst < 1e-4)) { printf("FAIL\n"); return 1; }
  printf("ok\n");
  return 0;
}
