// fp32 VALU issue rate on MI355X: v_fma_f32 (plain, with an SGPR operand like the k-means assign kernel; inline asm) and v_pk_fma_f32, at 1 .. 8
// waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
template <int PK>
__global__ void k(float* out, const float* in, int iters) {
  float a[16];
  f2 p[8];
  const float s0 = in[0], s1 = in[1];
  for (int i = 0; i < 16; ++i) a[i] = in[i] + threadIdx.x;
  for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
    if (PK == 2) {   // v_pk_fma_f32 with an SGPR PAIR operand (two different scalars: the packed dot products of the k-means assign kernel)
      const f2 sp = {s0, s1}, vp = {s1, s0};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(sp), "v"(vp));
    } else if (PK) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = p[i] * f2{s0, s0} + f2{s1, s1};
    } else {
      // (inline asm: left as __builtin_fmaf, hipcc's SLP vectoriser PACKS these sixteen chains into eight v_pk_fma_f32 -- the first
      //  version of this benchmark therefore measured the packed rate twice and reported 147 TFLOP/s for "plain" v_fma_f32)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(s1));
    }
  }
  float t = 0;
  for (int i = 0; i < 16; ++i) t += a[i];
  for (int i = 0; i < 8; ++i) t += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
  float *out, *in;
  hipMalloc(&out, 1 << 26);
  hipMalloc(&in, 256);
  float h[64];
  for (int i = 0; i < 64; ++i) h[i] = 1.0f + 1e-7f * i;
  hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int pk = 0; pk < 3; ++pk)
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
      const int threads = 256, blocks = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
        else if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double fma = (double)blocks * threads * iters * 64.0;   // 64 FMAs per lane and iteration in both forms
      printf("%s  %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s\n", pk == 2 ? "v_pk_fma_f32 (SGPR pair)" : pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, 2 * fma / ms / 1e9);
    }
  return 0;
}
