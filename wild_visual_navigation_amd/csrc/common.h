// Shared device/host helpers for the WVN gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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

// ---- fp16 operand format (WVN_PREC_F16: the speed path with 11 significand bits, operand.h) -----------------------------------
typedef struct { uint16_t bits; } f16raw_t;  // raw fp16 bits in HBM, as a distinct type for the ElemIO dispatch below
typedef __attribute__((ext_vector_type(2))) _Float16 wvn_f16x2_t;
__host__ __device__ inline uint16_t f32_to_f16(float f) {
  const _Float16 h = (_Float16)f;  // round-to-nearest-even, overflow -> inf
  return __builtin_bit_cast(uint16_t, h);
}
__host__ __device__ inline float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ inline uint32_t pack_f16x2(float lo, float hi) {  // v_cvt_pk_f16_f32
  const wvn_f32x2_t v = {lo, hi};
  const wvn_f16x2_t b = __builtin_convertvector(v, wvn_f16x2_t);
  return __builtin_bit_cast(uint32_t, b);
}

// fp16 conversions SATURATE instead of overflowing to infinity (released DINO checkpoints carry outlier channels;
// one inf in q / k / v or in the hidden activation turns a whole frame's tokens into NaN): MODE.FP16_OVFL (hwreg 1, bit 23) makes every
// fp16 VALU result that overflows -- v_cvt_f16_f32, v_cvt_pk_f16_f32 included -- +-65504 while true infinities and NaNs pass through.
// One s_setreg per wave at kernel entry, nothing per element; every kernel that writes fp16 operands calls it first (a no-op for the bf16
// builds' arithmetic).  tests/test_gpu_robustness.py::test_fp16_conversions_saturate pins the behaviour on the hardware.
#if defined(__HIPCC__)
__device__ inline void wvn_fp16_saturate() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
#endif

template <typename T> struct ElemIO;
template <> struct ElemIO<f16raw_t> {
  __device__ static inline float load(const f16raw_t* p) { return f16_to_f32(p->bits); }
  __device__ static inline void store(f16raw_t* p, float v) { p->bits = f32_to_f16(v); }
};
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

// ---- bilinear sampling with align_corners=True (dino_interface.py:87-90, stego_interface.py:107) in ONE fixed, explicitly
// rounded operation order: the up-sampling kernel and the kernels that interpolate on the fly (pixel-resolution k-means)
// must produce the same bits whatever contraction flags their translation unit is compiled with.
// ATen: src = dst * (G-1)/(H-1) (fp32), i0 = (int)src, i1 = i0 + (i0 < G-1), w1 = src - i0, w0 = 1 - w1.
struct LerpTap { int i0, i1; float w0, w1; };
// align_corners = 0 (round 6: the OTHER reading of the absent STEGO package's code interpolation, StegoInterface(code_align_corners=False)): ATen's
// half-pixel coordinates src = max((G / H) (o + 0.5) - 0.5, 0), every operation rounded on its own; the mode travels as the SIGN of the scale
// (-G / H), so every kernel that derives its taps through lerp_tap serves both readings
__host__ __device__ inline float lerp_scale(int G, int H, int align_corners = 1) {
  if (!align_corners) return -((float)G / (float)H);
  return H > 1 ? (float)(G - 1) / (float)(H - 1) : 0.f;
}
__device__ inline LerpTap lerp_tap(int o, int G, float scale) {
  // __fmul_rn / __fsub_rn are plain operators to the compiler, and under HIP's default -ffp-contract=fast the backend fuses
  // s - i0 into fma(scale, o, -i0) whatever pragma the source carries: the product is made opaque before it is used again
  float s = __fmul_rn(scale, (float)o);
  asm volatile("" : "+v"(s));
  LerpTap t;
  t.i0 = (int)s;
  t.i1 = t.i0 + (t.i0 < G - 1 ? 1 : 0);
  t.w1 = __fsub_rn(s, (float)t.i0);
  t.w0 = __fsub_rn(1.f, t.w1);
  return t;
}
// the same for a scale of either sign (lerp_scale's align_corners = 0 form); a separate function so that the kernels that only ever see
// align_corners=True keep their instruction streams (the packed k-means assign kernel's hand-issued scalar loads sit on a knife's edge of
// the register allocator: the hazard screen of csrc/build.py fired when this branch was added to lerp_tap itself)
__device__ inline LerpTap lerp_tap_ac(int o, int G, float scale) {
  float s;
  if (scale >= 0.f) {
    s = __fmul_rn(scale, (float)o);
  } else {
    float m = __fmul_rn(-scale, (float)o + 0.5f);
    asm volatile("" : "+v"(m));
    s = fmaxf(__fsub_rn(m, 0.5f), 0.f);
  }
  asm volatile("" : "+v"(s));
  LerpTap t;
  t.i0 = (int)s;
  t.i1 = t.i0 + (t.i0 < G - 1 ? 1 : 0);
  t.w1 = __fsub_rn(s, (float)t.i0);
  t.w0 = __fsub_rn(1.f, t.w1);
  return t;
}
__device__ inline float bilerp_fixed(float v00, float v01, float v10, float v11, float wx0, float wx1, float wy0, float wy1) {
  const float t0 = __fmaf_rn(wx1, v01, __fmul_rn(wx0, v00));
  const float t1 = __fmaf_rn(wx1, v11, __fmul_rn(wx0, v10));
  return __fmaf_rn(wy1, t1, __fmul_rn(wy0, t0));
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

// Opt kernels in to more than 64 KB of dynamic LDS.  The attribute is per DEVICE and one process may drive several GPUs
// (DinoInterface.change_device, dino_interface.py:61-68), so `done` keeps one bit per device ordinal.  Setting the attribute
// twice is harmless: two host threads racing here cost one redundant call, never a missing one.
struct LdsOptIn {
  std::atomic<unsigned long long> done{0};
  template <typename... Fn>
  int operator()(int bytes, Fn... fns) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    const void* list[] = {fns...};
    for (const void* f : list) {
      e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e != hipSuccess) return (int)e;
    }
    done.fetch_or(bit, std::memory_order_release);
    return 0;
  }
};

// One 16-byte buffer store whose soffset is an SGPR.  gfx950: such a store is still reading its data registers for a few cycles after
// issue, and a VALU write to them in the next two issue slots corrupts some lanes of the stored data; LLVM's hazard recognizer covers
// only the immediate-soffset form (scripts/check_store_hazard.py screens the library, tests/test_isa_hazards.py keeps it clean).  The
// two wait states behind the store take the data registers as an operand, so the allocator cannot recycle them inside the window,
// while the scheduler stays free to move other work around the pair (no scheduling barrier: these stores ride between MFMAs).
#if defined(__HIPCC__)
__device__ inline void wvn_store_b128_guarded(u32x4_t v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
  asm volatile("s_nop 1" ::"v"(v));
}
#endif

#if defined(__HIPCC__)
// (a, b) -> packed fp16 pair and the packed fp16 pair of what the rounding left behind
__device__ inline void wvn_split2_f16(float a, float b, uint32_t& hi, uint32_t& lo) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  hi = pack_f16x2(a, b);
  const h2_t hh = __builtin_bit_cast(h2_t, hi);
  lo = pack_f16x2(a - (float)hh[0], b - (float)hh[1]);
}
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
