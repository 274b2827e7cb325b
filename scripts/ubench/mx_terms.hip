// Micro-benchmark (round 5): what would MX-format lo-terms buy the split-operand linears AT THE MATRIX PIPE?  Per 64 k of one 32 x 32 output
// tile the <= 1e-3 mode issues today 12 v_mfma_f32_32x32x16_bf16 (hi*lo + lo*hi + hi*hi, four k-steps): 384 cycles.  The error budget
// (scripts/error_budget.py, modes c8 / c6) allows the two correction products on block-scaled e4m3 / e2m3 operands through
// v_mfma_scale_f32_32x32x64_f8f6f4: 4 x 32x32x16_f16 (hi*hi) + 2 x 32x32x64 fp8 (2 x 64 cycles) = 256, or + 2 x fp6 (2 x 32) = 192.
// Bare MFMA streams (no operand traffic): shader ticks per 64-k tile step and wall-clock rate, at 1 / 2 / 4 waves per SIMD -- the chip is
// power-managed in this regime, so the wall clock, not the cycle count, is the answer.
//   hipcc --offload-arch=gfx950 -O3 -o mx_terms mx_terms.hip && ./mx_terms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int MODE>
__global__ void k(const unsigned* src, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16_t acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  i32x8_t a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (int)src[lane * 8 + i]; b8[i] = (int)src[512 + lane * 8 + i]; }
  const bf16x8_t ab = __builtin_bit_cast(bf16x8_t, *(const __attribute__((ext_vector_type(4))) int*)&a8), bb = __builtin_bit_cast(bf16x8_t, *(const __attribute__((ext_vector_type(4))) int*)&b8);
  const f16x8_t ah = __builtin_bit_cast(f16x8_t, ab), bh = __builtin_bit_cast(f16x8_t, bb);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {   // four output tiles per iteration (independent accumulators), 64 k each
      if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 12; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
        constexpr int F = MODE == 1 ? 0 : 2;   // 0: e4m3, 2: e2m3 (fp6)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd, const unsigned* src, float* out, long long* cyc) {
  const int iters = 4000, blocks = 256, threads = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(blocks * threads / 64);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (auto v : h) m += (double)v;
  m /= h.size();
  const double tiles = (double)blocks * (threads / 64) * iters * 4;             // 64-k tile steps executed
  const double eq = tiles * 32.0 * 32 * 64 * 2 / (ms * 1e-3) / 1e12;            // algorithmic TFLOP/s of the split product (one product per tile step)
  printf("%-44s %d wave(s)/SIMD: %7.1f ticks per tile step and wave, %7.3f ms, %6.0f algorithmic TFLOP/s\n", name, waves_per_simd, m / (iters * 4.0), ms, eq);
}

int main() {
  unsigned* src; float* out; long long* cyc;
  (void)hipMalloc(&src, 1 << 16); (void)hipMemset(src, 0x3c, 1 << 16);
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
  run<0>("(warm-up)", 1, src, out, cyc);
  for (int w : {1, 2, 4}) {
    run<0>("today: 12 x 32x32x16 bf16", w, src, out, cyc);
    run<1>("c8: 4 x 32x32x16 f16 + 2 x 32x32x64 e4m3", w, src, out, cyc);
    run<2>("c6: 4 x 32x32x16 f16 + 2 x 32x32x64 e2m3", w, src, out, cyc);
  }
  return 0;
}
