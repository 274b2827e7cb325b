// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 from one wave per SIMD, as a function of the number of independent
// accumulator chains, in shader cycles (s_memtime) and wall time.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NCHAIN, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, 1) void k(const bf16x8_t* in, float* out, long long* cyc, int iters) {
  bf16x8_t a[NCHAIN], b = in[threadIdx.x & 63];
  f32x16_t acc[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) {
    a[c] = in[64 + c * 64 + (threadIdx.x & 63)];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32 / NCHAIN; ++u)
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], b, acc[c], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// FILL: 1 = one s_nop after every MFMA, 2 = the fused-MLP pattern: every MFMA's A operand comes from a ds_read_b128 issued 4 MFMAs
// earlier (ring of 4 fragments), 3 = same + 4 VALU ops per MFMA
template <int NCHAIN, int FILL, int W = 1>   // W = waves per SIMD (256 * W threads per workgroup, one workgroup per CU)
__global__ __launch_bounds__(256 * W, 1) void kf(const bf16x8_t* in, float* out, long long* cyc, int iters) {
  __shared__ bf16x8_t lds[64 * 32];
  for (int i = threadIdx.x; i < 64 * 32; i += 256 * W) lds[i] = in[i];
  bf16x8_t b = in[threadIdx.x & 63];
  f32x16_t acc[NCHAIN];
  float v[4] = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  __syncthreads();
  const bf16x8_t* base = lds + (threadIdx.x & 63);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (FILL == 1) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        acc[u % NCHAIN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc[u % NCHAIN], 0, 0, 0);
        asm volatile("s_nop 0");
      }
    } else {
      bf16x8_t wf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wf[q] = base[q * 64];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        acc[u % NCHAIN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u & 3], b, acc[u % NCHAIN], 0, 0, 0);
        if (u + 4 < 32) wf[u & 3] = base[((u + 4) & 31) * 64];
        if constexpr (FILL == 3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
        }
        if constexpr (FILL == 4) {   // 4 transcendentals per MFMA
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = __builtin_amdgcn_exp2f(v[q]);
        }
        if constexpr (FILL == 5) {   // the attention mix per MFMA: 2 exp2, 1 max3, 1 cvt_pk, 1 dot2c
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
          v[2] = __builtin_fmaxf(__builtin_fmaxf(v[2], v[0]), v[1]);
          unsigned pk;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[0]), "v"(v[1]));
          typedef __attribute__((ext_vector_type(2))) __bf16 b2;
          v[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, pk), __builtin_bit_cast(b2, 0x3f803f80u), v[3], false);
        }
        if constexpr (FILL == 7) {   // lazy-max attention mix: 2 exp2, cvt_pk, dot2c
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
          unsigned pk;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[0]), "v"(v[1]));
          typedef __attribute__((ext_vector_type(2))) __bf16 b2;
          v[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, pk), __builtin_bit_cast(b2, 0x3f803f80u), v[3], false);
        }
        if constexpr (FILL == 8 || FILL == 9) {   // one of the two exp2 (8) / every fourth (9: only when u & 1) as a VALU polynomial: fract, sub, 3 fma, cvt, ldexp
          const bool poly = FILL == 8 || (u & 1);
          if (poly) {
            const float f = __builtin_amdgcn_fractf(v[0]);
            const float n = v[0] - f;
            float pz = __builtin_fmaf(f, 0.0555f, 0.2402f);
            pz = __builtin_fmaf(pz, f, 0.6931f);
            pz = __builtin_fmaf(pz, f, 1.0f);
            v[0] = __builtin_amdgcn_ldexpf(pz, (int)n);
          } else {
            v[0] = __builtin_amdgcn_exp2f(v[0]);
          }
          v[1] = __builtin_amdgcn_exp2f(v[1]);
          unsigned pk;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[0]), "v"(v[1]));
          typedef __attribute__((ext_vector_type(2))) __bf16 b2;
          v[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, pk), __builtin_bit_cast(b2, 0x3f803f80u), v[3], false);
        }
        if constexpr (FILL >= 10 && FILL <= 13) {   // cost of the pieces: 10: 2 exp + cvt_pk; 11: 2 exp + dot2c; 12: 2 exp + cvt_pk + 2 v_add_f32; 13: 2 exp + cvt_pk + 1 v_add (sum of the pair by one 3-op add? no: v_add3 is integer) -> v_fma(p0, 1, p1) style
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
          unsigned pk = 0x3f803f80u;
          if constexpr (FILL != 11) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[0]), "v"(v[1]));
          typedef __attribute__((ext_vector_type(2))) __bf16 b2;
          if constexpr (FILL == 11) { pk = __builtin_bit_cast(unsigned, v[0]); v[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, pk), __builtin_bit_cast(b2, 0x3f803f80u), v[3], false); }
          if constexpr (FILL == 12) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[3]) : "v"(v[0])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[2]) : "v"(v[1])); }
          if constexpr (FILL == 13) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[3]) : "v"(v[0])); }
          if constexpr (FILL != 11) asm volatile("" :: "v"(pk));
        }
        if constexpr (FILL >= 14 && FILL <= 17) {   // 2 exp + a pack: 14 v_cvt_pkrtz_f16_f32, 15 v_perm_b32 (bf16 truncation), 16 v_cvt_pk_f16_f32, 17 v_and_or (bf16 truncation in 1 VOP3)
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
          unsigned pk;
          if constexpr (FILL == 14) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[0]), "v"(v[1]));
          if constexpr (FILL == 15) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk) : "v"(v[1]), "v"(v[0]), "s"(0x07060302u));
          if constexpr (FILL == 16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[0]), "v"(v[1]));
          if constexpr (FILL == 17) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(pk) : "v"(v[1]), "s"(0xffff0000u), "v"(v[0]));
          asm volatile("" :: "v"(pk));
        }
        if constexpr (FILL == 6) {   // 2 exp2 only
          v[0] = __builtin_amdgcn_exp2f(v[0]);
          v[1] = __builtin_amdgcn_exp2f(v[1]);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int u = 0; u < 28; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (FILL == 3 || FILL == 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if constexpr (FILL == 5) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        if constexpr (FILL == 7) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if constexpr (FILL == 10 || FILL == 11 || (FILL >= 14 && FILL <= 17)) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        if constexpr (FILL == 12) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        if constexpr (FILL == 13) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if constexpr (FILL == 8) __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
        if constexpr (FILL == 9) { if (u & 1) __builtin_amdgcn_sched_group_barrier(0x002, 10, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
        if constexpr (FILL == 6) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 * W + threadIdx.x / 64] = t1 - t0;
}

template <int NCHAIN, int FILL, int W = 1>
void runf(int grid, const bf16x8_t* in, float* out, long long* cyc, const char* tag) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((kf<NCHAIN, FILL, W>), dim3(grid), dim3(256 * W), 0, 0, in, out, cyc, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((kf<NCHAIN, FILL, W>), dim3(grid), dim3(256 * W), 0, 0, in, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 4 * W);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= h.size();
  const double n = (double)iters * 32;
  // W waves share a SIMD: ticks per MFMA per SIMD = ticks per MFMA per wave / W
  printf("%-36s grid %4d W %d: %.1f ticks/MFMA/SIMD, wall %.3f ms -> %.2f ns/MFMA/SIMD, %.0f TFLOP/s\n", tag, grid, W, mean / n / W, ms,
         ms * 1e6 / n / W, n * grid * 4 * W * 32768.0 / (ms * 1e-3) / 1e12);
}

template <int NCHAIN, int W>
void run(int grid, const bf16x8_t* in, float* out, long long* cyc, const char* tag) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NCHAIN, W>), dim3(grid), dim3(256 * W), 0, 0, in, out, cyc, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NCHAIN, W>), dim3(grid), dim3(256 * W), 0, 0, in, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid * 4 * W);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= h.size();
  const double n = (double)iters * 32;
  printf("%-28s grid %4d: %.1f ticks/MFMA/wave, wall %.3f ms -> %.2f ns/MFMA/wave, %.0f TFLOP/s\n", tag, grid, mean / n, ms, ms * 1e6 / n,
         n * grid * 4 * W * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  bf16x8_t* in; float* out; long long* cyc;
  (void)hipMalloc(&in, 64 * 64 * sizeof(bf16x8_t)); (void)hipMalloc(&out, 1024 * 1024 * 4); (void)hipMalloc(&cyc, 1024 * 16 * 8);
  std::vector<unsigned short> h(64 * 64 * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 20 & 0x3ff);  // random-ish bf16 near 1
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int grid : {1, 256}) {
    runf<2, 1>(grid, in, out, cyc, "2 chains + s_nop");
    runf<4, 1>(grid, in, out, cyc, "4 chains + s_nop");
    runf<2, 2>(grid, in, out, cyc, "2 chains + ds_read ring");
    runf<4, 2>(grid, in, out, cyc, "4 chains + ds_read ring");
    runf<8, 2>(grid, in, out, cyc, "8 chains + ds_read ring");
    runf<2, 3>(grid, in, out, cyc, "2 chains + ds_read ring + 4 VALU");
    runf<4, 3>(grid, in, out, cyc, "4 chains + ds_read ring + 4 VALU");
    runf<8, 3>(grid, in, out, cyc, "8 chains + ds_read ring + 4 VALU");
    runf<4, 4>(grid, in, out, cyc, "4 chains + ds_read + 4 exp2");
    runf<4, 6>(grid, in, out, cyc, "4 chains + ds_read + 2 exp2");
    runf<4, 5>(grid, in, out, cyc, "4 chains + ds_read + attn mix (5)");
    runf<4, 7>(grid, in, out, cyc, "lazy attn mix: 2 exp, cvt, dot2c");
    runf<4, 10>(grid, in, out, cyc, "2 exp + cvt_pk");
    runf<4, 14>(grid, in, out, cyc, "2 exp + v_cvt_pkrtz_f16_f32");
    runf<4, 15>(grid, in, out, cyc, "2 exp + v_perm_b32 (trunc bf16)");
    runf<4, 16>(grid, in, out, cyc, "2 exp + v_cvt_pk_f16_f32");
    runf<4, 17>(grid, in, out, cyc, "2 exp + v_and_or_b32");
    runf<4, 11>(grid, in, out, cyc, "2 exp + dot2c");
    runf<4, 12>(grid, in, out, cyc, "2 exp + cvt_pk + 2 v_add_f32");
    runf<4, 13>(grid, in, out, cyc, "2 exp + cvt_pk + 1 v_add_f32");
    runf<4, 8>(grid, in, out, cyc, "1 exp + 1 poly exp2, cvt, dot2c");
    runf<4, 9>(grid, in, out, cyc, "1.5 exp + 0.5 poly exp2, cvt, dot2c");
    // the same mixes with two (three) waves per SIMD: does a second wave's MFMA fill the first one's VALU / transcendental time?
    runf<4, 2, 2>(grid, in, out, cyc, "4 chains + ds_read ring");
    runf<4, 6, 2>(grid, in, out, cyc, "4 chains + ds_read + 2 exp2");
    runf<4, 5, 2>(grid, in, out, cyc, "4 chains + ds_read + attn mix (5)");
    runf<4, 7, 2>(grid, in, out, cyc, "lazy attn mix: 2 exp, cvt, dot2c");
    runf<4, 12, 2>(grid, in, out, cyc, "2 exp + cvt_pk + 2 v_add_f32");
    runf<4, 4, 2>(grid, in, out, cyc, "4 chains + ds_read + 4 exp2");
    runf<2, 7, 3>(grid, in, out, cyc, "lazy attn mix, 2 chains");
    runf<2, 7, 4>(grid, in, out, cyc, "lazy attn mix, 2 chains");
    run<1, 1>(grid, in, out, cyc, "1 chain, 1 wave/SIMD");
    run<2, 1>(grid, in, out, cyc, "2 chains, 1 wave/SIMD");
    run<4, 1>(grid, in, out, cyc, "4 chains, 1 wave/SIMD");
    run<8, 1>(grid, in, out, cyc, "8 chains, 1 wave/SIMD");
    run<2, 2>(grid, in, out, cyc, "2 chains, 2 waves/SIMD");
    run<4, 2>(grid, in, out, cyc, "4 chains, 2 waves/SIMD");
  }
  return 0;
}
