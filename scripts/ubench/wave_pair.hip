// Micro-benchmark (round 5): would a wave PAIR per 32 rows with the output columns split -- two waves per SIMD, each
// with half the accumulator and half the DMA pieces -- hide the cost of issuing the weight stream that bounds the one-wave-per-SIMD
// row-panel kernels (csrc/gemm_n384_x3.hip: 36 MFMAs + 6 one-kilobyte LDS-DMA pieces + 26 ds_read_b128 per k-step and wave, one s_barrier
// per k-step; measured 1451 cycles per k-step against 1152 of MFMAs)?
//   form A: 4 waves per workgroup (one per SIMD): per k-step and wave 36 MFMAs (12 accumulators), 6 pieces, 26 fragment reads
//   form B: 8 waves per workgroup (two per SIMD): per k-step and wave 18 MFMAs (6 accumulators), 3 pieces, 14 fragment reads
// Same matrix work per SIMD and k-step (36 MFMAs = 1152 cycles); reports shader-clock ticks per k-step.
//   hipcc --offload-arch=gfx950 -O3 -o wave_pair wave_pair.hip && ./wave_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int WAVES, bool DMA, bool READS>
__global__ __launch_bounds__(WAVES * 64, 1) void k(const unsigned char* src, float* out, long long* cyc, int iters, unsigned bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 4 stages x 32 KB
  constexpr int NT = 48 / WAVES;          // accumulator tiles per wave (12 / 6)
  constexpr int NP = 24 / WAVES;          // DMA pieces per wave and k-step (6 / 3)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  f32x16_t acc[NT];
  for (int c = 0; c < NT; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8_t a_hi = *(const bf16x8_t*)(src + lane * 16), a_lo = *(const bf16x8_t*)(src + (64 + lane) * 16);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    const int stage = i & 3;
    unsigned char* st = lds + stage * 32768;
    unsigned char* nx = lds + ((i + 3) & 3) * 32768;
    // A fragments of this k-step (two planes): in the real kernel one coalesced global load per plane, six k-steps ahead
    if (READS) { a_hi = *(const bf16x8_t*)(st + 24576 + lane * 16); a_lo = *(const bf16x8_t*)(st + 24576 + 4096 + lane * 16); }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bf16x8_t w_hi, w_lo;
      if (READS) {
        w_hi = *(const bf16x8_t*)(st + ((wave * NT + t) % 12) * 1024 + lane * 16);
        w_lo = *(const bf16x8_t*)(st + 12288 + ((wave * NT + t) % 12) * 1024 + lane * 16);
      } else { w_hi = a_hi; w_lo = a_lo; }
      if (DMA && t < NP) {   // one piece of the slice three k-steps ahead rides per tile
        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((((i * 24 + wave * NP + t)) * 1024u) % (bytes - 1024u)));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(nx + (wave * NP + t) * 1024), 16, lane * 16, so, 0, 0);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, w_lo, acc[t], 0, 0, 0);
      acc[(t + 1) % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, w_hi, acc[(t + 1) % NT], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, w_hi, acc[t], 0, 0, 0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (24 / WAVES)) : "memory");   // two slices stay in flight
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < NT; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int WAVES, bool DMA, bool READS>
double run(const char* name, const unsigned char* src, float* out, long long* cyc, unsigned bytes) {
  const int iters = 3000, blocks = 256;
  (void)hipFuncSetAttribute((const void*)k<WAVES, DMA, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<WAVES, DMA, READS>), dim3(blocks), dim3(WAVES * 64), 131072, 0, src, out, cyc, iters, bytes);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(blocks * WAVES);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (auto v : h) m += (double)v;
  m /= h.size();
  const double tf = 256.0 * 4 * 36 * 32768.0 * iters / (ms * 1e-3) / 1e12;
  printf("%-78s %8.2f ticks per k-step   %7.3f ms   %6.0f TFLOP/s issued\n", name, m / iters, ms, tf);
  return m / iters;
}

int main() {
  const unsigned bytes = 8u << 20;
  unsigned char* src; float* out; long long* cyc;
  (void)hipMalloc(&src, bytes); (void)hipMemset(src, 0, bytes);
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  run<4, false, false>("(warm-up)", src, out, cyc, bytes);
  const double a0 = run<4, false, false>("A  4 waves (1 per SIMD): 36 MFMAs per wave, nothing else", src, out, cyc, bytes);
  run<8, false, false>("B  8 waves (2 per SIMD): 18 MFMAs per wave, nothing else", src, out, cyc, bytes);
  run<4, false, true>("A  + 26 ds_read_b128 per wave", src, out, cyc, bytes);
  run<8, false, true>("B  + 14 ds_read_b128 per wave", src, out, cyc, bytes);
  run<4, true, false>("A  + 6 LDS-DMA pieces per wave", src, out, cyc, bytes);
  run<8, true, false>("B  + 3 LDS-DMA pieces per wave", src, out, cyc, bytes);
  const double a = run<4, true, true>("A  the k-step of gemm_n384_x3: 36 MFMAs + 6 pieces + 26 reads, barrier", src, out, cyc, bytes);
  const double b = run<8, true, true>("B  the wave-pair k-step: 18 MFMAs + 3 pieces + 14 reads per wave, barrier", src, out, cyc, bytes);
  printf("ticks are s_memtime units; MFMA-only k-step = 36 x 32 = 1152 shader cycles = %.2f ticks -> A = %.0f cycles, B = %.0f cycles per k-step and SIMD\n", a0, a / a0 * 1152, b / a0 * 1152);
  return 0;
}
