// Shared device/host helpers for the WVN gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WVN_OK 0
#define WVN_ERR_ARG 1001      // bad shape / null pointer / unsupported configuration
#define WVN_ERR_WORKSPACE 1002  // caller-provided workspace too small

#define WVN_WAVE 64

typedef uint16_t bf16_t;  // raw bf16 bits in HBM / LDS
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ---------------------------------------
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)h) << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two floats -> packed bf16x2 (lo in bits 0..15) with the hardware converter (RNE): hipcc selects
// v_cvt_pk_bf16_f32 for the vector conversion.  Deliberately NOT inline asm: the compiler pads no MFMA-result
// hazards for an asm statement's operands, and accumulators are often converted straight out of an MFMA.
typedef __attribute__((ext_vector_type(2))) float wvn_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 wvn_bf16x2_t;
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  const wvn_f32x2_t v = {lo, hi};
  const wvn_bf16x2_t b = __builtin_convertvector(v, wvn_bf16x2_t);
  return __builtin_bit_cast(uint32_t, b);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  __device__ static inline float load(const float* p) { return *p; }
  __device__ static inline void store(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  __device__ static inline float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static inline void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- wave / block reductions (64-wide) ----------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ inline float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Observed dispatch places block b on XCD b % 8 (guide T1).  Remap the linear block id so each XCD
// works on a contiguous chunk of tiles (neighbouring tiles share operand panels in that XCD's L2).
// Bijective for any nblk (speed only; never correctness).
__device__ inline int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int q = nblk / nx, r = nblk % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

#define WVN_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
