// Row quantisers feeding the fp8 GEMMs (gemm_fp8.hip): one 64-lane wave per row, the row held in registers, two reductions
// (amax; for the LayerNorm form also the two LN statistics), then e4m3 conversion with the hardware converter
// (v_cvt_pk_fp8_f32, OCP e4m3fn on gfx950, round-to-nearest-even; inputs are pre-scaled into [-448, 448]).  Per-row
// ("per-token") scales need no global reduction and no calibration pass; the scale of a row is amax / 448 (1 for an all-zero row).
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr float FP8_MAX = 448.0f;

__device__ inline unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);   // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);        // bytes 2, 3
  return (unsigned)w;
}

// cols = 4 * 64 * VPT4: a lane owns VPT4 groups of 4 consecutive columns, group g at column 256 g + 4 lane
template <typename TIN, int VPT4, bool LN>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const TIN* __restrict__ src, int lds_, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            unsigned char* __restrict__ q, int ldq, float* __restrict__ scale,
                                                            int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TIN* xr = src + (size_t)row * lds_;
  float v[VPT4][4];
#pragma unroll
  for (int g = 0; g < VPT4; ++g) {
    const int c = 256 * g + 4 * lane;
    if (c < cols) {
      if constexpr (sizeof(TIN) == 4) {
        const f32x4_t t = *(const f32x4_t*)(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[g][e] = t[e];
      } else {
        const u32x2_t t = *(const u32x2_t*)(xr + c);
        v[g][0] = __uint_as_float(t[0] << 16); v[g][1] = __uint_as_float(t[0] & 0xffff0000u);
        v[g][2] = __uint_as_float(t[1] << 16); v[g][3] = __uint_as_float(t[1] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = 0.f;
    }
  }
  if constexpr (LN) {   // torch LayerNorm semantics (biased variance, eps inside the sqrt), two-pass fp32 statistics
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < VPT4; ++g) s += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
    const float mean = wave_sum(s) / (float)cols;
    float qq = 0.f;
#pragma unroll
    for (int g = 0; g < VPT4; ++g)
      if (256 * g + 4 * lane < cols) {   // lanes past the row hold zeros: they must not add mean^2 each (D = 384: 128 of them)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[g][e] - mean; qq += d * d; }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)cols + eps);
#pragma unroll
    for (int g = 0; g < VPT4; ++g) {
      const int c = 256 * g + 4 * lane;
      if (c < cols) {
        const f32x4_t gm = *(const f32x4_t*)(gamma + c), bt = *(const f32x4_t*)(beta + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[g][e] = (v[g][e] - mean) * rstd * gm[e] + bt[e];
      }
    }
  }
  float am = 0.f;
#pragma unroll
  for (int g = 0; g < VPT4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(v[g][e]));
  am = wave_max(am);
  const float sc = am > 0.f ? am / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int g = 0; g < VPT4; ++g) {
    const int c = 256 * g + 4 * lane;
    if (c < cols) {
      // |v * inv| <= 448 (1 + 2^-23): clamp so that the converter never sees a value beyond the largest finite e4m3
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = fminf(fmaxf(v[g][e] * inv, -FP8_MAX), FP8_MAX);
      *(unsigned*)(q + (size_t)row * ldq + c) = pack4_fp8(t[0], t[1], t[2], t[3]);
    }
  }
}

template <typename TIN, bool LN>
int dispatch(const TIN* src, int lds_, const float* g, const float* b, float eps, unsigned char* q, int ldq, float* scale, int rows,
             int cols, hipStream_t st) {
  if (cols % 4 || cols <= 0 || cols > 4096 || (lds_ % 4) || (ldq % 4)) return WVN_ERR_ARG;
  dim3 grid(ceil_div(rows, 4)), block(256);
  const int v = ceil_div(cols, 256);
#define WVN_Q(V) hipLaunchKernelGGL((quantize_rows_kernel<TIN, V, LN>), grid, block, 0, st, src, lds_, g, b, eps, q, ldq, scale, rows, cols)
  switch (v) {
    case 1: WVN_Q(1); break;
    case 2: WVN_Q(2); break;
    case 3: WVN_Q(3); break;
    case 6: WVN_Q(6); break;
    case 12: WVN_Q(12); break;
    default:
      if (v <= 4) WVN_Q(4);
      else if (v <= 8) WVN_Q(8);
      else WVN_Q(16);
  }
#undef WVN_Q
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

int wvn_quantize_rows_fp8_launch(const void* src, int src_bf16, int lds_, unsigned char* q, int ldq, float* scale, int rows,
                                 int cols, hipStream_t st) {
  if (!src || !q || !scale || rows <= 0) return WVN_ERR_ARG;
  if (src_bf16) return dispatch<bf16_t, false>((const bf16_t*)src, lds_, nullptr, nullptr, 0.f, q, ldq, scale, rows, cols, st);
  return dispatch<float, false>((const float*)src, lds_, nullptr, nullptr, 0.f, q, ldq, scale, rows, cols, st);
}

int wvn_layernorm_fp8_launch(const float* x, const float* gamma, const float* beta, unsigned char* q, int ldq, float* scale,
                             int rows, int D, float eps, hipStream_t st) {
  if (!x || !gamma || !beta || !q || !scale || rows <= 0) return WVN_ERR_ARG;
  return dispatch<float, true>(x, D, gamma, beta, eps, q, ldq, scale, rows, D, st);
}
