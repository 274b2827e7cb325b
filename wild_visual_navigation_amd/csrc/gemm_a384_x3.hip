// A-stationary SPLIT-OPERAND MFMA GEMM for the K = 384 linears of the ViT (QKV, attention projection, fc1) on gfx950:
//     C = epilogue(A[M,384] * W[N,384]^T + bias),   A and W as hi + lo bf16 planes, every product hi*hi + hi*lo + lo*hi
// (three v_mfma_f32_32x32x16_bf16 per fragment pair, fp32 accumulation: the arithmetic of gemm_x3.hip, fp32-class results).
// The kernel behind precisions "mixed" (WVN_PREC_MIX) and "exact" (WVN_PREC_X3) where the shape allows.
//
// Why: the output-tiled gemm_x3_kernel issues 320 - 580 TFLOP/s of matrix work on these shapes (profiles/r04c_*): with K = 384 a
// 128 x 128 tile has twelve K-tiles, so its pipeline fill and its epilogue are as long as its MFMA loop.  Here -- the structure of
// gemm_a384.hip, re-sized for two planes per operand --
//   * a workgroup (4 waves, ONE per SIMD: the 512-entry register file) owns 128 rows of A for the whole N range; a wave keeps its
//     32 rows x 384 of BOTH planes in registers (192 VGPRs, loaded once per row block, already in MFMA operand layout);
//   * W streams through a 3-deep LDS ring in slices of 64 (n) x 128 (k) x 2 planes = 32 KB, written by direct-to-LDS loads
//     (buffer_load ... lds), one s_barrier per slice, counted vmcnt; LDS rows are 256 B with the 16-byte chunks XOR-swizzled by
//     (row & 15): conflict-free ds_read_b128;
//   * a fragment pair read from LDS (W hi, W lo of one 32 x 16 block: two ds_read_b128) feeds THREE MFMAs, so the LDS pipe and the
//     DMA issue cost per MFMA are two thirds / one half of the single-plane kernel's; 48 MFMAs per wave and slice;
//   * the epilogue of column tile j - 1 rides in the slice periods of tile j, one third per slice (bias in the accumulator
//     initialisation, activation + plane split + wave-private LDS transpose, 16-byte coalesced row stores).
// X_GELU_FRAG: the hidden activation leaves fc1 FRAGMENT-MAJOR -- for every 32-row group R and k-step s of the consumer (fc2) one 1 KB
// block per plane at ((R * N / 16 + s) * 1024), lane-major: 16 bytes per lane = the eight values of the lane's row whose column
// indices are 16 s + swap23(8 hi + j) (bits 2 and 3 swapped: exactly what the accumulators of a TR tile hold per lane), i.e. ready-made
// MFMA operand fragments for a consumer whose weight has the same column permutation (gemm_n384_x3.hip, AFRAG).  No LDS transpose,
// and every store is one contiguous kilobyte per wave.
// Outputs: q | k (one fp16 plane each, q pre-scaled: the operands of the fp16 attention kernel; or hi / lo bf16 planes), v^T
// likewise, hi / lo planes with the exact erf GELU (fc1), fp32 residual read-modify-write (projection).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int KD = 384;
constexpr int BNT = 64;                         // output columns per column tile
constexpr int SLK = 128;                        // k per ring slice
constexpr int NSL = KD / SLK;                   // slices per column tile (3)
constexpr int NS = 3;                           // ring depth
constexpr int PLANE_BYTES = BNT * SLK * 2;      // 16 KB: one plane of a slice
constexpr int SLICE_BYTES = 2 * PLANE_BYTES;    // hi | lo
constexpr int RING_BYTES = NS * SLICE_BYTES;    // 96 KB
constexpr int IMG_BF16 = 32 * 144;              // a wave's 32 x 64 bf16 image (rows of 144 B)
constexpr int STG_BYTES = 2 * 64 * 80;          // two V^T plane images (64 rows of 80 B each: 10240 B) >= two bf16 images (9216) >= the fp32 image (8704)
constexpr int STG_OFF = RING_BYTES;
constexpr int BIAS_OFF = STG_OFF + 4 * STG_BYTES;
constexpr int BM = 128;
constexpr int PIECES = SLICE_BYTES / 1024 / 4;  // 1 KB DMA pieces per wave and slice (8)
static_assert(PIECES == 8, "one DMA piece per k-step of a slice");

enum { X_GELU = 0, X_RESID = 1, X_QK = 2, X_V = 3, X_PLANES = 4, X_GELU_FRAG = 5, X_QK_F16 = 6, X_V_F16 = 7, X_QKV = 8, X_QKV_F16 = 9 };   // _F16: one fp16 plane per tensor
// X_QKV / X_QKV_F16: q | k | v^T in ONE launch -- the column tiles of q and k run in the TR orientation, those of v^T in the other; a row
// block's two planes (192 registers) are loaded once instead of twice (617 MB per launch at 128 frames) and there is one prologue / tail

struct X384Params {
  const bf16_t* A; const bf16_t* A_lo; int lda;
  const bf16_t* W; size_t w_plane;      // [2][N][384]: lo plane w_plane elements behind the hi plane
  const float* bias;
  void* C; void* C_lo; int ldc;         // planes (bf16) or fp32 (X_RESID: in/out; C_lo unused)
  void* C_h8;                           // MX, X_GELU_FRAG: C = fp16 fragments, C_lo = the l8 plane, C_h8 = the h8 plane (gemm_n384_x3.hip's MX operand)
  int M, N;
  int heads, npad, ntok_s;
  float q_scale;
  int f16_out;                          // X_QK / X_V: 1 = ONE fp16 plane per tensor, 0 = hi / lo bf16 planes
  int q_lo_f16;                         // with f16_out: q leaves as TWO fp16 planes (the second at qkv_base_lo + q_off)
  bf16_t* qkv_base; unsigned q_off, k_off, v_off, qkv_bytes;   // one buffer descriptor over q / k / v^T (hi planes or the fp16 planes)
  bf16_t* qkv_base_lo;                  // the same span of the lo planes (f16_out == 0)
  const float* ls;                      // X_RESID: optional LayerScale
  long long* dbg;                       // TIMING builds: per wave {wait + barrier, DMA issue, MFMA steps (+ epilogue chunks), total} shader cycles
  // LNA: A = LayerNorm(ln_x) formed while the row block is loaded: ln_x fp32 [M][ln_ldx], ln_stats[m] = {mean, rstd} (left by the kernel
  // that wrote the rows: gemm_n384_x3.hip), gamma / beta [384]
  const float* ln_x; int ln_ldx; const float* ln_stats; const float* ln_g; const float* ln_b;
};

__device__ inline void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
  lo = pack_bf16x2(a - ah, b - bh);
}
// erf GELU to fp32 rounding with ONE transcendental: for a = |x|
//     gelu(x) = max(x, 0) - a * erfc(a / sqrt 2) / 2 = max(x, 0) - a * 2^R(a),     R(a) = log2 erfc(a / sqrt 2) - 1
// R is smooth and nearly quadratic; a degree-6 polynomial fitted with the weight a * erfc(a / sqrt 2) (the derivative of the result with
// respect to R) reproduces gelu within 2.8e-7 absolute on the whole line (float64 erfc reference, fp32 Horner; weighted least squares
// reweighted towards minimax on [0, 7]; tests/test_host_logic.py pins the bound) -- the same as Abramowitz-Stegun 7.1.26 (1.5e-7 on erf), which this
// replaces: that form costs 15 VALU operations and TWO transcendentals (v_rcp, v_exp: four issue slots each) per value, and the fc1
// epilogue rides in the shadow of the MFMAs of a wave that is alone on its SIMD (fc1 18.6 -> 17.6 ms per 64-frame step of the mixed mode).
// (The same idea with a cubic R in the fp16 path's mlp_fused kernel: 22.8 -> 22.55 ms per step, headline within the noise, and it gives
//  up that path's RELATIVE accuracy on the negative tail -- not adopted.)
// Beyond a = 7 the subtracted term is below 1e-11: a is clamped there (the polynomial is only trusted on the fitted interval).
// Two values at a time: the polynomial and the final fma on v_pk_fma_f32 (written on vectors: left to hipcc the scalar form becomes
// v_fmaak_f32 with literal constants, one issue slot per value and step).
__device__ inline void gelu_pair(float& x0, float& x1) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const f2 a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x0), 0.f, 7.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x1), 0.f, 7.0f)};
  const f2 c6 = {3.309327076e-05f, 3.309327076e-05f}, c5 = {-7.692237268e-04f, -7.692237268e-04f}, c4 = {8.080729283e-03f, 8.080729283e-03f},
           c3 = {-5.341212451e-02f, -5.341212451e-02f}, c2 = {-4.587709606e-01f, -4.587709606e-01f}, c1 = {-1.151201725e+00f, -1.151201725e+00f},
           c0 = {-9.999930859e-01f, -9.999930859e-01f};
  f2 r = __builtin_elementwise_fma(a, c6, c5);
  r = __builtin_elementwise_fma(a, r, c4);
  r = __builtin_elementwise_fma(a, r, c3);
  r = __builtin_elementwise_fma(a, r, c2);
  r = __builtin_elementwise_fma(a, r, c1);
  r = __builtin_elementwise_fma(a, r, c0);
  const f2 e = {__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};
  const f2 m = {__builtin_amdgcn_fmed3f(x0, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(x1, 0.f, 3.0e38f)};   // max(x, 0) in ONE v_med3 (fmaxf: canonicalise + v_max)
  const f2 g = __builtin_elementwise_fma(-a, e, m);
  x0 = g[0]; x1 = g[1];
}

// MX (round 6; LNA only, X_GELU_FRAG / X_QKV_F16): the operand representation of gemm_n384_x3.hip's MX kernel -- h = fp16(v), l8 = e5m2((v - h) * 2^12),
// h8 = e5m2(v) -- for both operands: per k-step region TWO fp16 MFMAs (hi * hi of the two column halves) and ONE scaled e5m2 MFMA of K = 64 (one of
// the slice's eight correction products: 2 column halves x 2 64-k steps x {a_h8 w_l8, a_l8 w_h8}) = 128 matrix-pipe cycles instead of 192, the same
// four ds_read_b128.  W: plane 0 = fp16 [N][384]; plane 1 = bytes [N][768] in the chunk order the kernel reads them (backbone.pack_a384_mx).
// The row block's operands stay resident: 24 fp16 fragments + 6 eight-register l8 operands = 144 registers; a_h8 = e5m2(a_h) is derived from the fp16 fragments once per
// (slice, 64-k step) -- 16 conversions for two regions -- which is also what the row-panel consumer does with ITS activations (one definition of h8 for activations: e5m2(fp16(v))).
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
constexpr int MX_SC_ONE = 0x7f7f7f7f, MX_SC_RES = 0x73737373;   // E8M0 scale bytes 2^0 / 2^-12 (every byte alike)
constexpr float MX_RES_INV = 1.0f / 4096.0f;
#ifndef WVN_MXG
#define WVN_MXG 12
#endif
constexpr int MXG = WVN_MXG;   // k-steps of fp32 rows requested at a time by the MX LayerNorm-on-load prologue (12: two HBM round trips per row block)                    // v_cvt_scalef32_* DIVIDES by its scale operand (scripts/ubench/mx_formats.hip)
__device__ inline uint32_t mx_pk8(uint32_t old, float a, float b, float inv_scale, bool hi_word) {
  const s16x2_t o = __builtin_bit_cast(s16x2_t, old);
  return __builtin_bit_cast(uint32_t, hi_word ? __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(o, a, b, inv_scale, true)
                                              : __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(o, a, b, inv_scale, false));
}

constexpr int X384_LDS_MAX = 160 * 1024;

template <int EPI, bool TIMING = false, bool LNA = false, bool MX = false, bool BURST = false>
__global__ __launch_bounds__(256, 1) void gemm_a384_x3_kernel(X384Params p) {
  static_assert(!MX || (LNA && (EPI == X_GELU_FRAG || EPI == X_QKV_F16)), "the MX form exists for the LayerNorm-on-load fc1 / QKV instantiations");
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int NT = p.N / BNT;
  // persistent, unit-balanced schedule (gemm_a384.hip): (row block, column tile) units in row-block-major order
  const int NRB = (p.M + BM - 1) / BM;
  const long long U = (long long)NRB * NT;
  const int u_begin = (int)(U * blockIdx.x / gridDim.x), u_end = (int)(U * (blockIdx.x + 1) / gridDim.x);
  const int total = (u_end - u_begin) * NSL;
  int m0w = 0;
  unsigned char* stg = smem + STG_OFF + wave * STG_BYTES;
  const float* bias_l = (const float*)(smem + BIAS_OFF);

  // ---- W ring producer: 32 wave-instructions of 1 KB (4 rows x 256 B of one plane) per slice, 8 per wave ----
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((p.w_plane + (size_t)p.N * KD) * 2), 0x00020000);
  unsigned wvoff[PIECES];
#pragma unroll
  for (int u = 0; u < PIECES; ++u) {
    const int piece = wave * PIECES + u, plane = piece >> 4, inst = piece & 15;
    const int row = inst * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (row & 15);
    wvoff[u] = (unsigned)(plane * p.w_plane * 2 + (row * KD + chunk * 8) * 2);
  }
  int iss_j = u_begin % NT, iss_ks = 0;
  unsigned iss_soff = 0;
  auto issue_begin = [&]() __attribute__((always_inline)) { iss_soff = __builtin_amdgcn_readfirstlane((iss_j * BNT * KD + iss_ks * SLK) * 2); };
  auto issue_piece = [&](int i, int u) __attribute__((always_inline)) {   // piece u of this wave's PIECES of slice i (u is a compile-time constant at every call site)
    unsigned char* dst = smem + (i % NS) * SLICE_BYTES + wave * PIECES * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, wvoff[u], iss_soff, 0, 0);
  };
  auto issue_end = [&]() __attribute__((always_inline)) { if (++iss_ks == NSL) { iss_ks = 0; if (++iss_j == NT) iss_j = 0; } };
  auto issue = [&](int i) __attribute__((always_inline)) {
    issue_begin();
#pragma unroll
    for (int u = 0; u < PIECES; ++u) issue_piece(i, u);
    issue_end();
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < total) issue(i);

  for (int i = tid; i < p.N; i += 256) ((float*)(smem + BIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
  const float* gam_l = (const float*)(smem + BIAS_OFF + p.N * 4);   // LNA: gamma [384], beta [384] behind the bias table
  if constexpr (LNA)
    for (int i = tid; i < 2 * KD; i += 256) ((float*)(smem + BIAS_OFF + p.N * 4))[i] = i < KD ? p.ln_g[i] : p.ln_b[i - KD];
  bf16x8_t xh[KD / 16], xl[KD / 16];
  u32x4_t mh[KD / 16];                    // MX: the fp16 fragments
  u32x4_t m8[KD / 64][2];                 // MX: the l8 operands [64-k step][half]: dword 2 (s & 1) + e of half (s >> 1) & 1 = bytes j = 4 e .. 4 e + 3 of k-step s
  //                                          (the h8 operands are derived per use from mh: e5m2 of the fp16 image)
  auto load_a = [&]() __attribute__((always_inline)) {
    if constexpr (LNA && MX) {
      const int row = min(m0w + l31, p.M - 1);
      const wvn_f32x2_t st = *(const wvn_f32x2_t*)(p.ln_stats + 2 * (size_t)row);
      const float a1 = st[1], a0 = -st[0] * st[1];
      const float* xr = p.ln_x + (size_t)row * p.ln_ldx + hi * 8;
#pragma unroll
      for (int s0 = 0; s0 < KD / 16; s0 += MXG) {
        f32x4_t u[2 * MXG];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MXG; ++i) {
          u[2 * i] = *(const f32x4_t*)(xr + (s0 + i) * 16);
          u[2 * i + 1] = *(const f32x4_t*)(xr + (s0 + i) * 16 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MXG; ++i) {
          const int s = s0 + i;
          const f32x4_t g0 = *(const f32x4_t*)(gam_l + s * 16 + hi * 8), g1 = *(const f32x4_t*)(gam_l + s * 16 + hi * 8 + 4);
          const f32x4_t b0 = *(const f32x4_t*)(gam_l + KD + s * 16 + hi * 8), b1 = *(const f32x4_t*)(gam_l + KD + s * 16 + hi * 8 + 4);
          float y[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = fmaf(fmaf(u[2 * i][e], a1, a0), g0[e], b0[e]);
            y[4 + e] = fmaf(fmaf(u[2 * i + 1][e], a1, a0), g1[e], b1[e]);
          }
          u32x4_t hv;
          uint32_t d8[2] = {0, 0};   // the l8 dwords of this k-step
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
            const uint32_t hb = pack_f16x2(y[2 * e], y[2 * e + 1]);   // (bit_cast of the SCALAR: clang reads element 0 when handed a vector element)
            hv[e] = hb;
            const h2_t hh = __builtin_bit_cast(h2_t, hb);
            d8[e >> 1] = mx_pk8(d8[e >> 1], y[2 * e] - (float)hh[0], y[2 * e + 1] - (float)hh[1], MX_RES_INV, e & 1);
          }
          asm volatile("" : "+v"(hv));
          mh[s] = hv;
          m8[s >> 2][(s >> 1) & 1][2 * (s & 1)] = d8[0];
          m8[s >> 2][(s >> 1) & 1][2 * (s & 1) + 1] = d8[1];
        }
      }
    } else if constexpr (LNA) {
      // the rows arrive as fp32 (the residual stream itself: the same bytes as two bf16 planes) and are normalised with the statistics
      // their producer left, scaled, shifted and split on the way into the operand registers: ~5 VALU per value once per row block
      // (18 - 24 column tiles of three slice periods each), instead of a kernel that reads the rows and writes the planes
      const int row = min(m0w + l31, p.M - 1);
      const wvn_f32x2_t st = *(const wvn_f32x2_t*)(p.ln_stats + 2 * (size_t)row);
      const float a1 = st[1], a0 = -st[0] * st[1];
      const float* xr = p.ln_x + (size_t)row * p.ln_ldx + hi * 8;
      // twelve k-steps = 24 row pieces = 96 registers at a time: two memory round trips per row block, for a wave that has nothing to hide
      // them behind (four k-steps at a time: six round trips, +4.7 ms per step; all 48 pieces at once: ~180 spilled registers).  The
      // scheduling barriers keep each group's requests in front of its arithmetic, the empty asm keeps the arithmetic of a k-step in
      // front of the next group's requests.
#pragma unroll
      for (int s0 = 0; s0 < KD / 16; s0 += 12) {
        f32x4_t u[24];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          u[2 * i] = *(const f32x4_t*)(xr + (s0 + i) * 16);
          u[2 * i + 1] = *(const f32x4_t*)(xr + (s0 + i) * 16 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int s = s0 + i;
          const f32x4_t g0 = *(const f32x4_t*)(gam_l + s * 16 + hi * 8), g1 = *(const f32x4_t*)(gam_l + s * 16 + hi * 8 + 4);
          const f32x4_t b0 = *(const f32x4_t*)(gam_l + KD + s * 16 + hi * 8), b1 = *(const f32x4_t*)(gam_l + KD + s * 16 + hi * 8 + 4);
          float y[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = fmaf(fmaf(u[2 * i][e], a1, a0), g0[e], b0[e]);
            y[4 + e] = fmaf(fmaf(u[2 * i + 1][e], a1, a0), g1[e], b1[e]);
          }
          uint32_t h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split2(y[2 * e], y[2 * e + 1], h[e], l[e]);
          u32x4_t hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
          asm volatile("" : "+v"(hv), "+v"(lv));
          xh[s] = __builtin_bit_cast(bf16x8_t, hv);
          xl[s] = __builtin_bit_cast(bf16x8_t, lv);
        }
      }
    } else {
      const size_t ro = (size_t)min(m0w + l31, p.M - 1) * p.lda + hi * 8;
#pragma unroll
      for (int s = 0; s < KD / 16; ++s) {
        xh[s] = *(const bf16x8_t*)(p.A + ro + s * 16);
        xl[s] = *(const bf16x8_t*)(p.A_lo + ro + s * 16);
      }
    }
  };

  f32x16_t acc[2], prev[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; prev[t][r] = 0.f; }

  const int xorc = l31 & 15;
  const unsigned rd_base = l31 * 256;

  // ---- epilogue addressing ----
  constexpr unsigned OOB = 0x80000000u;
  constexpr bool MERGED = EPI == X_QKV || EPI == X_QKV_F16;   // q | k tiles (TR) and v^T tiles (!TR) in one launch: the orientation is the TILE's tag
  constexpr bool IS_QKV = EPI == X_QK || EPI == X_QK_F16 || EPI == X_V || EPI == X_V_F16 || MERGED;
  constexpr bool IS_V_ONLY = EPI == X_V || EPI == X_V_F16;
  constexpr bool F16OUT = EPI == X_QK_F16 || EPI == X_V_F16 || EPI == X_QKV_F16;   // (compile-time: a run-time flag leaves branches in the slice loop)
  using TRK = std::integral_constant<bool, !IS_V_ONLY>;
  const int nqk = MERGED ? 2 * p.heads : (1 << 30);      // MERGED: column tiles below nqk are q | k (one head each), the rest v^T
  unsigned voff[2] = {0, 0};
  unsigned vt_off = 0, stg_rd = 0;
  const unsigned c_bytes = IS_QKV ? 0u : (unsigned)((size_t)(EPI == X_GELU_FRAG ? (p.M + 31) / 32 * 32 : p.M) * p.ldc * (EPI == X_RESID ? 4 : 2));
  const __amdgpu_buffer_rsrc_t rs_c = IS_QKV ? __builtin_amdgcn_make_buffer_rsrc(p.qkv_base, 0, p.qkv_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c2 = IS_QKV ? __builtin_amdgcn_make_buffer_rsrc(p.qkv_base_lo ? p.qkv_base_lo : p.qkv_base, 0, p.qkv_bytes, 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc(p.C_lo ? p.C_lo : p.C, 0, MX ? c_bytes / 2 : c_bytes, 0x00020000);
  auto qkv_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int hblk = 0; hblk < 2; ++hblk) {
      const int m = m0w + hblk * 16 + (lane >> 3);
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      voff[hblk] = m < p.M ? (unsigned)((((size_t)b * p.heads * p.npad + tk) * 64 + (lane & 7) * 8) * 2) : OOB;
    }
    {
      const int m = m0w + (lane & 3) * 8;
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      vt_off = m < p.M ? (unsigned)((((size_t)b * p.heads * 64 + (lane >> 2)) * p.npad + tk) * 2) : OOB;
    }
  };
  if constexpr (EPI == X_RESID) {
    voff[0] = (unsigned)(((lane >> 4) * p.ldc + (lane & 15) * 4) * 4);
    stg_rd = (lane >> 4) * 272 + (lane & 15) * 16;
  } else if constexpr (!IS_QKV) {
    voff[0] = (unsigned)(((lane >> 3) * p.ldc + (lane & 7) * 8) * 2);
    stg_rd = (lane >> 3) * 144 + (lane & 7) * 16;
  }

  // ---- one k-step (16 of the slice's 128) of one ring slice: the (W hi, W lo) fragment pairs of the two column halves -> six MFMAs,
  // alternating between the two accumulators (a filler between two MFMAs on the SAME accumulator costs ~43 cycles, between different
  // ones ~6: MI355X guide); the four fragments of k-step s + 1 are requested during the six MFMAs of step s (192 cycles of cover) ----
  bf16x8_t wh[2][2], wl[2][2];   // [k-step parity][column half]
  u32x4_t resid_q[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  u32x4_t fragh = {0, 0, 0, 0}, fragl = {0, 0, 0, 0};   // X_GELU_FRAG: the fragment being assembled
  u32x4_t frag8l = {0, 0, 0, 0};  // MX: the 16-byte half of the l8 plane being assembled
  u32x4_t wq[2][2], w8[2][2];   // MX: [k-step parity][column half] fp16 fragments; [k-step parity][half] of the region's e5m2 operand
  u32x4_t dh8[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // MX: a_h8 of the current 64-k step, derived from its four fp16 fragments (not resident: -48 registers)
  auto frag_read = [&](int slot, int s, int par) __attribute__((always_inline)) {
    const unsigned char* base = smem + slot * SLICE_BYTES + rd_base;
    if constexpr (MX) {
      // region s: the fp16 fragments of k-step s (both column halves) and correction product s = (64-k step mm = s >> 2, which = (s >> 1) & 1:
      // 0 = w_l8 (x a_h8), 1 = w_h8 (x a_l8); column half t8 = s & 1): plane-1 chunks 2 s'' + hi, s'' = 4 which + 2 mm + x
      const int mm = s >> 2, which = (s >> 1) & 1, t8 = s & 1;
#pragma unroll
      for (int t = 0; t < 2; ++t) wq[par][t] = *(const u32x4_t*)(base + t * 8192 + (((2 * s + hi) ^ xorc) << 4));
#pragma unroll
      for (int x = 0; x < 2; ++x) w8[par][x] = *(const u32x4_t*)(base + PLANE_BYTES + t8 * 8192 + (((2 * (4 * which + 2 * mm + x) + hi) ^ xorc) << 4));
      return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned o = t * 8192 + (((2 * s + hi) ^ xorc) << 4);
      wh[par][t] = *(const bf16x8_t*)(base + o);
      wl[par][t] = *(const bf16x8_t*)(base + PLANE_BYTES + o);
    }
  };
  auto mfma_step = [&](int slot, int ks, int s, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int cur = s & 1;
    if (!MX && s + 1 < 8) frag_read(slot, s + 1, cur ^ 1);   // (MX: requested at the top of the region, in front of a scheduling barrier -- see period())
    if constexpr (MX) {
      const int mm = s >> 2, which = (s >> 1) & 1, t8 = s & 1;
      const f16x8_t af = __builtin_bit_cast(f16x8_t, mh[ks * 8 + s]);
      if (which == 0 && t8 == 0) {   // e5m2 of the fp16 image (what the row-panel consumer derives for ITS activations too), once per (slice, 64-k step): two regions use it
        typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
#pragma unroll
        for (int sfr = 0; sfr < 4; ++sfr) {
          uint32_t d[2] = {0, 0};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t pr = mh[ks * 8 + 4 * mm + sfr][e];   // (scalar copy before the bit_cast)
            const s16x2_t o = __builtin_bit_cast(s16x2_t, d[e >> 1]);
            d[e >> 1] = __builtin_bit_cast(uint32_t, (e & 1) ? __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f, true)
                                                             : __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f, false));
          }
          dh8[sfr >> 1][2 * (sfr & 1)] = d[0];
          dh8[sfr >> 1][2 * (sfr & 1) + 1] = d[1];
        }
      }
      const u32x4_t a0 = which ? m8[ks * 2 + mm][0] : dh8[0], a1 = which ? m8[ks * 2 + mm][1] : dh8[1];
      const i32x8_t av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
      const u32x4_t w0 = w8[cur][0], w1 = w8[cur][1];
      const i32x8_t wv = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
      // accumulators alternate: even regions 0, 1, 0 -- odd regions 1, 0, 1 (the scaled MFMA goes to column half t8 = s & 1)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int t = n ^ t8;
        const f16x8_t wf = __builtin_bit_cast(f16x8_t, wq[cur][t]);
        if constexpr (TR) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf, acc[t], 0, 0, 0);
      }
      // which 0: W_l8 (carries 2^12) x a_h8;  which 1: W_h8 x a_l8 (carries 2^12)
      if constexpr (TR) acc[t8] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, av, acc[t8], 1, 1, 0, which ? MX_SC_ONE : MX_SC_RES, 0, which ? MX_SC_RES : MX_SC_ONE);
      else acc[t8] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, wv, acc[t8], 1, 1, 0, which ? MX_SC_RES : MX_SC_ONE, 0, which ? MX_SC_ONE : MX_SC_RES);
      return;
    }
    const bf16x8_t ah = xh[ks * 8 + s], al = xl[ks * 8 + s];
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8_t w = term == 1 ? wl[cur][t] : wh[cur][t];
        const bf16x8_t a = term == 0 ? al : ah;          // hi*lo, lo*hi, hi*hi
        if constexpr (TR) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, acc[t], 0, 0, 0);
      }
  };

  // ---- the epilogue of column tile jp (accumulators in prev[]) in 3 x 8 chunks: part 0 / 1 = column half t of the tile (activation,
  // plane split, wave-private LDS image), one PAIR of values per chunk; part 2 = the image(s) -> global, one 16-byte store group per
  // chunk.  A chunk rides between the six MFMAs of one k-step.
  // TR tiles : images [32 rows m][64 cols n] (bf16 rows of 144 B: hi image at 0, lo image at IMG_BF16; fp32 rows of 272 B)
  // !TR tiles: images [64 rows n][32 cols m] (rows of 80 B; hi at 0, lo at 64 * 80)       -- V^T
  auto epi_chunk = [&](int part, int jp, int s, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = jp * BNT;
    if constexpr (EPI == X_RESID) {
      if (part == 1 && s >= 6)   // the first two residual row groups of part 2
        resid_q[s - 6] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, voff[0], __builtin_amdgcn_readfirstlane(((m0w + 4 * (s - 6)) * p.ldc + n0) * 4), 0);
    }
    if (part < 2) {
      const int t = part, g = s >> 1, h2 = (s & 1) * 2;     // values prev[t][4 g + h2], prev[t][4 g + h2 + 1]
      float v0 = prev[t][4 * g + h2], v1 = prev[t][4 * g + h2 + 1];
      if constexpr (TR) {
        const int c = 32 * t + 8 * g + 4 * hi + h2;
        if constexpr (EPI == X_GELU || EPI == X_GELU_FRAG) gelu_pair(v0, v1);
        if constexpr (EPI == X_GELU_FRAG && MX) {
          // part t fills half t of the tile's (= the consumer's 64-k step jp) e5m2 operands, chunk s its bytes 2 s, 2 s + 1; every fourth chunk completes
          // one fp16 fragment (consumer k-step 4 jp + 2 t + (g >> 1)), the eighth the 16-byte halves
          typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
          const uint32_t h = pack_f16x2(v0, v1);
          const h2_t hh = __builtin_bit_cast(h2_t, h);
          fragh[2 * (g & 1) + (h2 >> 1)] = h;
          // (no h8 plane: the consumer derives e5m2(h) from the fp16 fragments in registers -- gemm_n384_x3.hip: derive_h8)
          {
            typedef __attribute__((ext_vector_type(2))) float f2_t;
            const f2_t res = f2_t{v0, v1} - f2_t{(float)hh[0], (float)hh[1]};   // one v_pk_add_f32 for the two residues
            frag8l[s >> 1] = mx_pk8(frag8l[s >> 1], res[0], res[1], MX_RES_INV, s & 1);
          }
          if ((s & 3) == 3) {
            const unsigned so = __builtin_amdgcn_readfirstlane((((m0w >> 5) * (p.N >> 4)) + 4 * jp + 2 * t + (g >> 1)) * 1024);
            wvn_store_b128_guarded(fragh, rs_c, lane * 16, so);
          }
          if (s == 7) {
            const unsigned so8 = __builtin_amdgcn_readfirstlane((((m0w >> 5) * (p.N >> 6)) + jp) * 2048 + t * 1024);
            wvn_store_b128_guarded(frag8l, rs_c2, lane * 16, so8);
          }
          return;
        }
        if constexpr (EPI == X_GELU_FRAG) {
          uint32_t h, l;
          split2(v0, v1, h, l);
          const int e = 2 * (g & 1) + (h2 >> 1);            // dword of the lane's 16 bytes: elements j = 4 (g & 1) + h2, + 1
          fragh[e] = h; fragl[e] = l;
          if ((s & 3) == 3) {                                // the fragment of the consumer's k-step 4 jp + 2 t + (g >> 1) is complete
            const unsigned so = __builtin_amdgcn_readfirstlane((((m0w >> 5) * (p.N >> 4)) + 4 * jp + 2 * t + (g >> 1)) * 1024);
            wvn_store_b128_guarded(fragh, rs_c, lane * 16, so);
            wvn_store_b128_guarded(fragl, rs_c2, lane * 16, so);
          }
          return;
        }
        if constexpr (EPI == X_RESID) {
          if (p.ls) { v0 *= p.ls[n0 + c]; v1 *= p.ls[n0 + c + 1]; }
          const wvn_f32x2_t o = {v0, v1};
          *(wvn_f32x2_t*)(stg + l31 * 272 + c * 4) = o;
        } else if constexpr (IS_QKV && F16OUT) {   // (TR: a q | k tile)
          // fp16 planes; the rounding residue goes to a second image whatever the tile is -- only q tiles store it (part 2), and a branch
          // here would end the scheduling region
          const float qs = n0 < p.heads * 64 ? p.q_scale : 1.f;
          uint32_t h, l;
          wvn_split2_f16(v0 * qs, v1 * qs, h, l);
          *(uint32_t*)(stg + l31 * 144 + c * 2) = h;
          *(uint32_t*)(stg + IMG_BF16 + l31 * 144 + c * 2) = l;
        } else {
          uint32_t h, l;
          split2(v0, v1, h, l);
          *(uint32_t*)(stg + l31 * 144 + c * 2) = h;
          *(uint32_t*)(stg + IMG_BF16 + l31 * 144 + c * 2) = l;
        }
      } else {
        // V^T: tokens 8g + 4hi + e of the wave's 32, stored with bits 2 and 3 of the token index swapped inside aligned groups of 16
        const int mloc = 16 * (g >> 1) + 8 * hi + 4 * (g & 1) + h2;
        if constexpr (F16OUT) {
          *(uint32_t*)(stg + (32 * t + l31) * 80 + mloc * 2) = pack_f16x2(v0, v1);
        } else {
          uint32_t h, l;
          split2(v0, v1, h, l);
          *(uint32_t*)(stg + (32 * t + l31) * 80 + mloc * 2) = h;
          *(uint32_t*)(stg + 64 * 80 + (32 * t + l31) * 80 + mloc * 2) = l;
        }
      }
      return;
    }
    // part 2: wave-private image(s) -> global, 16 bytes per lane
    if constexpr (EPI == X_RESID) {
      const int it = s;
      const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 4 * it) * p.ldc + n0) * 4);
      f32x4_t v = *(const f32x4_t*)(stg + stg_rd + it * 4 * 272);
      // requested TWO chunks earlier: a wave alone on its SIMD has nobody to cover the round trip, and vmcnt retires in order -- the wait
      // for these rows is also a wait for every DMA piece requested before them (one per k-step)
      const u32x4_t r = resid_q[it & 1];
      if (it + 2 < 8)
        resid_q[it & 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, voff[0], __builtin_amdgcn_readfirstlane(((m0w + 4 * (it + 2)) * p.ldc + n0) * 4), 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(r[e]);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(v[e]);
      wvn_store_b128_guarded(o, rs_c, voff[0], so);
    } else if constexpr (IS_QKV && TR) {
      const int it = s & 3, pl = s >> 2;
      const int D = p.heads * 64;
      const int which = n0 / D, head = (n0 - which * D) >> 6;
      const unsigned so = __builtin_amdgcn_readfirstlane((which == 0 ? p.q_off : p.k_off) + head * p.npad * 64 * 2);
      const u32x4_t val = *(const u32x4_t*)(stg + pl * IMG_BF16 + ((lane >> 3) + it * 8) * 144 + (lane & 7) * 16);
      if (pl == 0) wvn_store_b128_guarded(val, rs_c, voff[it >> 1], so + (it & 1) * 1024);
      else {
        // the second plane: lo planes of q and k (hi / lo bf16 form), or -- fp16 form -- q's rounding residue where the caller asked for
        // a two-plane q (attention_bf16.hip QSPLIT); k tiles and a single-plane q drop the store through an out-of-range offset
        const bool keep = !F16OUT || (which == 0 && p.q_lo_f16);
        wvn_store_b128_guarded(val, rs_c2, keep ? voff[it >> 1] : OOB, so + (it & 1) * 1024);
      }
    } else if constexpr (IS_QKV && !TR) {
      const int it = s & 3, pl = s >> 2;
      if (pl == 1 && F16OUT) return;
      const int head = (MERGED ? n0 - 2 * p.heads * 64 : n0) >> 6;
      const u32x4_t val = *(const u32x4_t*)(stg + pl * 64 * 80 + ((lane >> 2) + it * 16) * 80 + (lane & 3) * 16);
      const unsigned so = __builtin_amdgcn_readfirstlane(p.v_off + (head * 64 + it * 16) * p.npad * 2);
      if (pl == 0) wvn_store_b128_guarded(val, rs_c, vt_off, so);
      else wvn_store_b128_guarded(val, rs_c2, vt_off, so);
    } else if constexpr (EPI == X_GELU_FRAG) {
      // (nothing: the fragments left in parts 0 and 1)
    } else {   // hi / lo planes [M][ldc]
      const int it = s >> 1, pl = s & 1;
      const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 8 * it) * p.ldc + n0) * 2);
      const u32x4_t val = *(const u32x4_t*)(stg + pl * IMG_BF16 + stg_rd + it * 8 * 144);
      const unsigned vo = m0w + 8 * it + (lane >> 3) < p.M ? voff[0] : OOB;
      if (pl == 0) wvn_store_b128_guarded(val, rs_c, vo, so);
      else wvn_store_b128_guarded(val, rs_c2, vo, so);
    }
  };

  auto init_acc = [&](int j, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = j * BNT;
    if constexpr (TR) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t b4 = *(const f32x4_t*)(bias_l + n0 + 32 * t + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b4[e];
        }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float b = bias_l[n0 + 32 * t + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b;
      }
    }
  };

  // ---- one slice period ----
  // VM queue at the boundary of slice i (oldest first): DMA(i) [requested in period i - 2], then DMA(i + 1); the stores of a tile
  // epilogue are issued in a ks == 2 period AFTER that period's DMA request, so they are the youngest operations at the next
  // boundary (ks == 0) only; everywhere else a wait for "all but the 8 youngest" covers them.
  // VM operations of one part 2 per lane -- exactly, or a LOWER bound where it depends on a run-time flag (allowing more operations to
  // stay outstanding than were issued would let DMA(i) itself slip through the wait)
  // VM operations an epilogue period issues per lane, exactly, or a LOWER bound where it depends on a run-time flag (allowing more
  // operations to stay outstanding than were issued would let DMA(i) itself slip through the wait).  ST2: the stores (and row fetches)
  // of part 2, in a ks == 2 period; ST01: the fragment stores of X_GELU_FRAG in the ks == 0 / 1 periods.
  constexpr int ST2 = EPI == X_RESID ? 14 : (EPI == X_GELU || EPI == X_PLANES ? 8 : (EPI == X_GELU_FRAG ? 0 : (F16OUT ? 4 : 8)));   // (QK fp16: 4 stores are the lower bound, a two-plane q issues 8)
  constexpr int ST01 = EPI == X_GELU_FRAG ? (MX ? 3 : 4) : 0;   // (MX: two fp16 fragments + one l8 half per column half)
  long long t_wait = 0, t_iss = 0, t_mfma = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto period = [&](int i, int ks, int j, bool stores_in_window, auto mtr, auto etr, auto epi_tag) __attribute__((always_inline)) {
    constexpr bool do_epi = decltype(epi_tag)::value;   // (compile-time: a branch around the epilogue would end the scheduling region)
    long long c0 = 0, c1 = 0, c2 = 0;
    if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
    if (i + 1 < total) {
      if (ks == 0 && stores_in_window && ST2 > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + ST2) : "memory");
      else if (ks != 0 && do_epi && ST01 > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + ST01) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
    // the DMA requests of slice i + 2 (into the ring slot every wave has just left): one 1 KB piece per k-step, riding with the
    // fillers, instead of eight behind the barrier (measured: 509 cycles of a 3700-cycle period went to issuing them in a burst)
    // (requested unconditionally: past the workgroup's last slice the cursor wraps to valid W rows and the data lands in a ring slot
    //  nobody reads again -- a uniform branch around every piece cuts each period into nine basic blocks)
    issue_begin();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TIMING) c2 = (long long)__builtin_amdgcn_s_memtime();
    // Eight scheduling regions per slice: the six MFMAs of a k-step, the LDS reads of the next step's fragments and ONE chunk of the
    // previous tile's epilogue.  This wave is alone on its SIMD, so the epilogue's VALU / LDS / store instructions must issue in the
    // MFMAs' shadow (<= 5 slots per 32-cycle MFMA): inside a region the issue order is pinned to MFMA, LDS read, VALU..., and the
    // regions keep the epilogue spread evenly over the slice (one region for the whole slice: the scheduler left the epilogue behind
    // the last MFMA)
    if constexpr (BURST) {   // (MX experiment) the slice's eight DMA pieces in one burst behind the barrier instead of one per region
#pragma unroll
      for (int s = 0; s < 8; ++s) issue_piece(i + NS - 1, s);
      __builtin_amdgcn_sched_barrier(0);
    }
    frag_read(i % NS, 0, 0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if constexpr (MX) {
        // the NEXT region's four fragments are requested first and pinned there: left inside the region the scheduler placed each read behind the MFMA
        // that frees its (coalesced) registers, i.e. one LDS round trip in front of every MFMA (2075 cycles per slice against 1024 of MFMAs)
        if (s + 1 < 8) frag_read(i % NS, s + 1, (s & 1) ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!BURST) issue_piece(i + NS - 1, s);
      mfma_step(i % NS, ks, s, mtr);
      if constexpr (do_epi) epi_chunk(ks, j - 1, s, etr);
      if constexpr (MX) {   // three MFMAs per region (32 + 32 + 64 cycles): the other instructions in three groups, the largest behind the scaled MFMA
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x030, 2, 0);
      } else {
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    issue_end();
    if constexpr (TIMING) { t_wait += c1 - c0; t_iss += c2 - c1; t_mfma += (long long)__builtin_amdgcn_s_memtime() - c2; }
  };
  int si = 0;
  auto tile = [&](int j, int jj, int j_end, auto mtr, auto etr, auto next_tr) __attribute__((always_inline)) {
    if (jj >= 1) {
#pragma unroll
      for (int ks = 0; ks < NSL; ++ks) period(si + ks, ks, j, jj >= 2, mtr, etr, std::true_type{});
    } else {
#pragma unroll
      for (int ks = 0; ks < NSL; ++ks) period(si + ks, ks, j, false, mtr, etr, std::false_type{});
    }
    si += NSL;
#pragma unroll
    for (int t = 0; t < 2; ++t) prev[t] = acc[t];
    if (j + 1 < j_end) init_acc(j + 1, next_tr);
  };

  __syncthreads();
  for (int u = u_begin; u < u_end;) {
    const int rb = u / NT, j0 = u - rb * NT, j1 = min(NT, j0 + (u_end - u));
    m0w = rb * BM + wave * 32;
    if constexpr (TIMING) {   // (the row block's operand load -- LayerNorm on load: two HBM round trips + the conversions -- is booked under "DMA issue")
      const long long l0 = (long long)__builtin_amdgcn_s_memtime();
      load_a();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t_iss += (long long)__builtin_amdgcn_s_memtime() - l0;
    } else {
      load_a();
    }
    if constexpr (IS_QKV) qkv_offsets();
    if constexpr (MERGED) {
      using T = std::true_type; using F = std::false_type;
      if (j0 < nqk) init_acc(j0, T{}); else init_acc(j0, F{});
      for (int j = j0; j < j1; ++j) {   // (tile's own orientation, the previous tile's for the epilogue riding in it, the next one's for the bias)
        if (j + 1 < nqk) tile(j, j - j0, j1, T{}, T{}, T{});
        else if (j < nqk) tile(j, j - j0, j1, T{}, T{}, F{});
        else if (j == nqk) tile(j, j - j0, j1, F{}, T{}, F{});
        else tile(j, j - j0, j1, F{}, F{}, F{});
      }
      if (j1 - 1 < nqk) {
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
          for (int c = 0; c < 8; ++c) epi_chunk(part, j1 - 1, c, T{});
      } else {
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
          for (int c = 0; c < 8; ++c) epi_chunk(part, j1 - 1, c, F{});
      }
    } else {
      init_acc(j0, TRK{});
      for (int j = j0; j < j1; ++j) tile(j, j - j0, j1, TRK{}, TRK{}, TRK{});
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int c = 0; c < 8; ++c) epi_chunk(part, j1 - 1, c, TRK{});
    }
    u += j1 - j0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last two periods' surplus DMA requests land before the wave ends)
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
      d[0] = t_wait; d[1] = t_iss; d[2] = t_mfma; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
    }
  }
}


// ======================================================================================================================================
// The MX form at TWO workgroups per CU (round 6, second half; the default of WVN_PREC_MIX's fc1 and q | k | v^T).  The kernel above keeps one wave per SIMD,
// and that wave issues everything itself: per k-step region 128 cycles of MFMAs against ~240 of its own instruction stream (one 1 KB DMA piece = ~54 cycles of
// issue, a chunk of the previous tile's GELU, four fragment reads and their waits).  Here the same work is cut so that TWO independent workgroups fit a CU
// (<= 256 registers, <= 80 KB of LDS each) and one workgroup's DMA issue, LDS round trips, barriers, LayerNorm-on-load prologue and epilogue run under the
// OTHER workgroup's MFMAs.  Measured (in-kernel counters, per wave and 64 output columns, fc1 at 403 456 rows): a wave alone on its SIMD 8.2 K cycles (regions
// 4.8 K against 3.1 K of MFMAs, epilogue 1.9 K, prologue 1.0 K, barriers 0.4 K) -- the same kernel time as the form above with none of its scheduling; two
// per SIMD 12.4 K each = 6.2 K per 64 columns and SIMD: fc1 1.21 -> 1.10 ms, q | k | v^T 0.87 -> 0.80 ms per 128 frame-passes, the step 1.8 ms shorter.
// Each wave stays bound by its own dependent chain (one accumulator: every MFMA waits for the one before it) and two of them do not fill the pipe; delaying the
// CU's second workgroup by half a tile (s_getreg HW_ID: its waves sit in the odd slots) changed nothing and is not in the code.
//   * a column tile is 32 wide: ONE 16-register accumulator (the first build kept 64 columns = two accumulators and had no room to prefetch
//     W fragments: a wave alone on its SIMD needed 219 cycles per 128 of MFMAs, two of them 286 each);
//   * the ring holds slices of 32 (n) x 128 (k) = 16 KB as THIRTY-TWO CHUNK IMAGES [32 rows][16 B]: sixteen of the fp16 plane (k-step s, half
//     hi -> image 2 s + hi: a fragment read is ONE contiguous kilobyte at lane * 16 + s * 1024), sixteen of the 8-bit planes (64-k step mm,
//     which (l8 | h8), half x, hi -> image 16 + 8 mm + 4 which + 2 x + hi); three slots; the weight is packed in exactly this order
//     (backbone.pack_a384_mx, second image): a slice is one contiguous 16 KB block, a DMA piece one contiguous kilobyte copied lane-linear,
//     no swizzle, one address register, every other offset an immediate;
//   * a period = one slice = four regions of 2 fp16 MFMAs (k-steps 2 r, 2 r + 1) + 1 scaled MFMA (64-k step mm = r >> 1; r even: W_l8 x
//     a_h8, r odd: W_h8 x a_l8) = 128 matrix-pipe cycles; region r + 1's four fragment reads are requested before region r's MFMAs (two
//     register sets);
//   * no second accumulator set: the tile's epilogue (16 values per lane) runs serially behind its third period;
//   * resident: 24 fp16 fragments + 6 l8 operands (144 registers) + 16 accumulators + 2 x 16 of W fragments + a_h8 (8).
constexpr int S2_BN = 32;                    // columns per tile
constexpr int S2_NSL = 3;                    // slices (of 128 k) per column tile
constexpr int S2_NS = 3;
constexpr int S2_SLICE = 16384;
constexpr int S2_RING = S2_NS * S2_SLICE;    // 48 KB
constexpr int S2_PIECES = S2_SLICE / 1024 / 4;   // 4 per wave and slice: one per region
constexpr int S2_STG = 5120;                 // per wave (QKV): two fp16 images of a tile [32][80 B] (q | k: the plane and its residue), or one V^T tile [32 n][80 B]
static_assert(S2_PIECES == 4, "one DMA piece per region");

template <int EPI, bool TIMING = false>
__global__ __launch_bounds__(256, 2) void gemm_a384_mx2_kernel(X384Params p) {
  static_assert(EPI == X_GELU_FRAG || EPI == X_QKV_F16, "fc1 + GELU -> MX planes, or q | k | v^T fp16");
  constexpr bool IS_QKV = EPI == X_QKV_F16;
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int NT = p.N / S2_BN;
  const int NRB = (p.M + BM - 1) / BM;
  const long long U = (long long)NRB * NT;
  const int u_begin = (int)(U * blockIdx.x / gridDim.x), u_end = (int)(U * (blockIdx.x + 1) / gridDim.x);
  const int total = (u_end - u_begin) * S2_NSL;
  constexpr int STG_OFF2 = S2_RING;
  constexpr int BIAS_OFF2 = STG_OFF2 + (IS_QKV ? 4 * S2_STG : 0);
  unsigned char* stg = smem + STG_OFF2 + wave * S2_STG;
  const float* bias_l = (const float*)(smem + BIAS_OFF2);
  const float* gam_l = (const float*)(smem + BIAS_OFF2 + p.N * 4);
  int m0w = 0;

  // ---- W ring producer: a slice = 16 contiguous kilobytes of the packed weight; a piece = two chunk images, copied lane-linear ----
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * KD * 4), 0x00020000);
  const unsigned l16 = lane * 16;
  const int n_slices = NT * S2_NSL;
  int iss_g = (u_begin % NT) * S2_NSL;    // the stream cursor: slice (tile j, k-slice) = j * 3 + ks, wrapping at the last tile
  unsigned iss_soff = 0;
  auto issue_begin = [&]() __attribute__((always_inline)) { iss_soff = __builtin_amdgcn_readfirstlane((unsigned)iss_g * (unsigned)S2_SLICE + wave * S2_PIECES * 1024); };
  auto issue_piece = [&](int i, int u) __attribute__((always_inline)) {   // (u: compile-time constant -> the instruction's immediate offset, which the hardware adds to the LDS address too)
    unsigned char* dst = smem + (i % S2_NS) * S2_SLICE + wave * S2_PIECES * 1024;
    __attribute__((address_space(3))) void* d3 = (__attribute__((address_space(3))) void*)dst;
    switch (u) {   // (the immediate must be a literal; u is a constant after unrolling and the switch folds)
      case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, iss_soff, 0, 0); break;
      case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, iss_soff, 1024, 0); break;
      case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, iss_soff, 2048, 0); break;
      default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, iss_soff, 3072, 0); break;
    }
  };
  auto issue_end = [&]() __attribute__((always_inline)) { if (++iss_g == n_slices) iss_g = 0; };
#pragma unroll
  for (int i = 0; i < S2_NS - 1; ++i)
    if (i < total) {
      issue_begin();
#pragma unroll
      for (int u = 0; u < S2_PIECES; ++u) issue_piece(i, u);
      issue_end();
    }
  for (int i = tid; i < p.N; i += 256) ((float*)(smem + BIAS_OFF2))[i] = p.bias ? p.bias[i] : 0.f;
  for (int i = tid; i < 2 * KD; i += 256) ((float*)(smem + BIAS_OFF2 + p.N * 4))[i] = i < KD ? p.ln_g[i] : p.ln_b[i - KD];

  u32x4_t mh[KD / 16];        // the row block's fp16 fragments
  u32x4_t m8[KD / 64][2];     // its l8 operands [64-k step][half]
  // LayerNorm on load: the prologue of the kernel above, eight k-steps of fp32 rows in flight (the other workgroup covers the round trips)
  auto load_a = [&]() __attribute__((always_inline)) {
    const int row = min(m0w + l31, p.M - 1);
    const wvn_f32x2_t st = *(const wvn_f32x2_t*)(p.ln_stats + 2 * (size_t)row);
    const float a1 = st[1], a0 = -st[0] * st[1];
    const float* xr = p.ln_x + (size_t)row * p.ln_ldx + hi * 8;
    constexpr int G = 8;   // k-steps of fp32 rows requested at a time (twelve: the same step time)
#pragma unroll
    for (int s0 = 0; s0 < KD / 16; s0 += G) {
      f32x4_t u[2 * G];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        u[2 * i] = *(const f32x4_t*)(xr + (s0 + i) * 16);
        u[2 * i + 1] = *(const f32x4_t*)(xr + (s0 + i) * 16 + 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int s = s0 + i;
        const f32x4_t g0 = *(const f32x4_t*)(gam_l + s * 16 + hi * 8), g1 = *(const f32x4_t*)(gam_l + s * 16 + hi * 8 + 4);
        const f32x4_t b0 = *(const f32x4_t*)(gam_l + KD + s * 16 + hi * 8), b1 = *(const f32x4_t*)(gam_l + KD + s * 16 + hi * 8 + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = fmaf(fmaf(u[2 * i][e], a1, a0), g0[e], b0[e]);
          y[4 + e] = fmaf(fmaf(u[2 * i + 1][e], a1, a0), g1[e], b1[e]);
        }
        u32x4_t hv;
        uint32_t d8[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
          const uint32_t hb = pack_f16x2(y[2 * e], y[2 * e + 1]);   // (bit_cast of the SCALAR: clang reads element 0 when handed a vector element)
          hv[e] = hb;
          const h2_t hh = __builtin_bit_cast(h2_t, hb);
          d8[e >> 1] = mx_pk8(d8[e >> 1], y[2 * e] - (float)hh[0], y[2 * e + 1] - (float)hh[1], MX_RES_INV, e & 1);
        }
        asm volatile("" : "+v"(hv), "+v"(d8[0]), "+v"(d8[1]));   // (pins the conversions HERE: left free they sink to their first use in the tile loop and the 192 fp32 values wait for them in scratch)
        mh[s] = hv;
        m8[s >> 2][(s >> 1) & 1][2 * (s & 1)] = d8[0];
        m8[s >> 2][(s >> 1) & 1][2 * (s & 1) + 1] = d8[1];
        __builtin_amdgcn_sched_barrier(0);   // (a k-step's residues are formed before the next one starts: left free, the scheduler parks all 192 fp32 values in scratch)
      }
    }
  };

  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // ---- epilogue addressing ----
  constexpr unsigned OOB = 0x80000000u;
  const int nqk = IS_QKV ? 4 * p.heads : (1 << 30);      // QKV: the 32-column tiles below nqk are q | k (half a head each), the rest v^T
  unsigned voff[2] = {0, 0};
  unsigned vt_off = 0;
  const unsigned c_bytes = IS_QKV ? 0u : (unsigned)((size_t)((p.M + 31) / 32 * 32) * p.ldc * 2);
  const __amdgpu_buffer_rsrc_t rs_c = IS_QKV ? __builtin_amdgcn_make_buffer_rsrc(p.qkv_base, 0, p.qkv_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c2 = IS_QKV ? __builtin_amdgcn_make_buffer_rsrc(p.qkv_base_lo ? p.qkv_base_lo : p.qkv_base, 0, p.qkv_bytes, 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc(p.C_lo ? p.C_lo : p.C, 0, c_bytes / 2, 0x00020000);
  auto qkv_offsets = [&]() __attribute__((always_inline)) {
    // q | k tile images: a row = 32 columns = 64 B = four lanes; a store instruction covers 16 rows
#pragma unroll
    for (int hblk = 0; hblk < 2; ++hblk) {
      const int m = m0w + hblk * 16 + (lane >> 2);
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      voff[hblk] = m < p.M ? (unsigned)((((size_t)b * p.heads * p.npad + tk) * 64 + (lane & 3) * 8) * 2) : OOB;
    }
    {
      const int m = m0w + (lane & 3) * 8;
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      vt_off = m < p.M ? (unsigned)((((size_t)b * p.heads * 64 + (lane >> 2)) * p.npad + tk) * 2) : OOB;
    }
  };

  u32x4_t wq[2][2], w8[2][2];   // [register set][k-step parity] fp16 fragments; [register set][half] of the region's e5m2 operand
  u32x4_t dh8[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  auto frag_read = [&](int slot, int r, int par) __attribute__((always_inline)) {
    const unsigned char* base = smem + slot * S2_SLICE + l16;
#pragma unroll
    for (int x = 0; x < 2; ++x) wq[par][x] = *(const u32x4_t*)(base + (2 * r + x) * 1024);
#pragma unroll
    for (int x = 0; x < 2; ++x) w8[par][x] = *(const u32x4_t*)(base + 8192 + (8 * (r >> 1) + 4 * (r & 1) + 2 * x) * 512);
  };
  auto derive_h8 = [&](int c) __attribute__((always_inline)) {   // a_h8 = e5m2 of the fp16 image of 64-k step c (what the row-panel consumer derives for its activations too)
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
#pragma unroll
    for (int sfr = 0; sfr < 4; ++sfr) {
      uint32_t d[2] = {0, 0};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t pr = mh[c * 4 + sfr][e];
        const s16x2_t o = __builtin_bit_cast(s16x2_t, d[e >> 1]);
        d[e >> 1] = __builtin_bit_cast(uint32_t, (e & 1) ? __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f, true)
                                                         : __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f, false));
      }
      dh8[sfr >> 1][2 * (sfr & 1)] = d[0];
      dh8[sfr >> 1][2 * (sfr & 1) + 1] = d[1];
    }
  };
  auto mfma_region = [&](int ks, int r, int par, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int c = 2 * ks + (r >> 1), which = r & 1;   // the 64-k step; 0: W_l8 (carries 2^12) x a_h8, 1: W_h8 x a_l8 (carries 2^12)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const f16x8_t af = __builtin_bit_cast(f16x8_t, mh[ks * 8 + 2 * r + x]);
      const f16x8_t wf = __builtin_bit_cast(f16x8_t, wq[par][x]);
      if constexpr (TR) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf, acc, 0, 0, 0);
    }
    const u32x4_t a0 = which ? m8[c][0] : dh8[0], a1 = which ? m8[c][1] : dh8[1];
    const i32x8_t av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const u32x4_t w0 = w8[par][0], w1 = w8[par][1];
    const i32x8_t wv = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
    if constexpr (TR) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, av, acc, 1, 1, 0, which ? MX_SC_ONE : MX_SC_RES, 0, which ? MX_SC_RES : MX_SC_ONE);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, wv, acc, 1, 1, 0, which ? MX_SC_RES : MX_SC_ONE, 0, which ? MX_SC_ONE : MX_SC_RES);
  };

  auto init_acc = [&](int j, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = j * S2_BN;
    if constexpr (TR) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t b4 = *(const f32x4_t*)(bias_l + n0 + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = b4[e];
      }
    } else {
      const float b = bias_l[n0 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = b;
    }
  };

  // ---- the tile's epilogue, serial (the CU's other workgroup has the matrix pipe meanwhile) ----
  auto epilogue = [&](int j, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = j * S2_BN;
    if constexpr (!IS_QKV) {
      // TR accumulators: lane (row l31, half hi) holds columns 8 g + 4 hi + e of the tile.  The consumer's 64-k step jp = j >> 1, half t = j & 1: fragment (k-step
      // 4 jp + 2 t + (g >> 1)) in the accumulators' own element order; l8 half t = the 16 bytes of the lane in register order
      const int jp = j >> 1, t = j & 1;
      u32x4_t frag8l = {0, 0, 0, 0};
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        u32x4_t fragh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // chunk s = 4 gg + c: g = s >> 1, h2 = (s & 1) * 2
          const int s = 4 * gg + c, g = s >> 1, h2 = (s & 1) * 2;
          float v0 = acc[4 * g + h2], v1 = acc[4 * g + h2 + 1];
          gelu_pair(v0, v1);
          typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
          typedef __attribute__((ext_vector_type(2))) float f2_t;
          const uint32_t h = pack_f16x2(v0, v1);
          const h2_t hh = __builtin_bit_cast(h2_t, h);
          fragh[2 * (g & 1) + (h2 >> 1)] = h;
          const f2_t res = f2_t{v0, v1} - f2_t{(float)hh[0], (float)hh[1]};
          frag8l[s >> 1] = mx_pk8(frag8l[s >> 1], res[0], res[1], MX_RES_INV, s & 1);
        }
        const unsigned so = __builtin_amdgcn_readfirstlane((((m0w >> 5) * (p.N >> 4)) + 4 * jp + 2 * t + gg) * 1024);
        wvn_store_b128_guarded(fragh, rs_c, l16, so);
      }
      const unsigned so8 = __builtin_amdgcn_readfirstlane((((m0w >> 5) * (p.N >> 6)) + jp) * 2048 + t * 1024);
      wvn_store_b128_guarded(frag8l, rs_c2, l16, so8);
    } else if constexpr (TR) {
      // a q | k tile = half a head's dims of 32 tokens: an fp16 image [32 tokens][32 dims] (rows of 80 B) and the image of the rounding residues behind it; q
      // pre-scaled.  Then 16 bytes per lane: 4 lanes per token row, 16 rows per store
      const int D = p.heads * 64;
      const int which = n0 / D, head = (n0 - which * D) >> 6, t = (n0 >> 5) & 1;
      const float qs = which == 0 ? p.q_scale : 1.f;
      const bool keep_lo = which == 0 && p.q_lo_f16;
      const unsigned so = __builtin_amdgcn_readfirstlane((which == 0 ? p.q_off : p.k_off) + head * p.npad * 64 * 2 + t * 64);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 4; h2 += 2) {
          uint32_t h, l;
          wvn_split2_f16(acc[4 * g + h2] * qs, acc[4 * g + h2 + 1] * qs, h, l);
          const int c = 8 * g + 4 * hi + h2;
          *(uint32_t*)(stg + l31 * 80 + c * 2) = h;
          *(uint32_t*)(stg + 2560 + l31 * 80 + c * 2) = l;
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const u32x4_t val = *(const u32x4_t*)(stg + ((lane >> 2) + it * 16) * 80 + (lane & 3) * 16);
        wvn_store_b128_guarded(val, rs_c, voff[it], so);
      }
      if (keep_lo) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const u32x4_t val = *(const u32x4_t*)(stg + 2560 + ((lane >> 2) + it * 16) * 80 + (lane & 3) * 16);
          wvn_store_b128_guarded(val, rs_c2, voff[it], so);
        }
      }
    } else {
      // V^T: lane (row n = l31 of the tile, half hi) holds tokens 8 g + 4 hi + e; image [32 n][32 tokens] with bits 2 and 3 of the token index swapped inside
      // aligned groups of 16 (the attention kernel's V^T fragment order), rows of 80 B; 4 lanes per row, 16 rows per store
      const int nv = n0 - 2 * p.heads * 64;     // row of V^T = head * 64 + dim
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 4; h2 += 2) {
          const int mloc = 16 * (g >> 1) + 8 * hi + 4 * (g & 1) + h2;
          *(uint32_t*)(stg + l31 * 80 + mloc * 2) = pack_f16x2(acc[4 * g + h2], acc[4 * g + h2 + 1]);
        }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const u32x4_t val = *(const u32x4_t*)(stg + ((lane >> 2) + it * 16) * 80 + (lane & 3) * 16);
        const unsigned so = __builtin_amdgcn_readfirstlane(p.v_off + (nv + it * 16) * p.npad * 2);
        wvn_store_b128_guarded(val, rs_c, vt_off, so);
      }
    }
  };

  // ---- one slice period: barrier (every wave has waited for its pieces of slice i at the END of its previous period), the four regions with the DMA
  // requests of slice i + 2 riding along, then the wait for this wave's pieces of slice i + 1 -- BEFORE the epilogue's stores are issued, so the queue at the
  // wait is exactly DMA(i + 1), DMA(i + 2) whatever follows (pieces are requested unconditionally; past the workgroup's last slice the cursor wraps to valid
  // rows and the data lands in a slot nobody reads again) ----
  long long t_bar = 0, t_loop = 0, t_epi = 0, t_pro = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto period = [&](int i, int ks, auto tr_tag) __attribute__((always_inline)) {
    long long c0 = 0, c1 = 0, c2 = 0;
    if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
    issue_begin();
    frag_read(i % S2_NS, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r + 1 < 4) frag_read(i % S2_NS, r + 1, (r & 1) ^ 1);
      issue_piece(i + S2_NS - 1, r);
      if ((r & 1) == 0) derive_h8(2 * ks + (r >> 1));
      __builtin_amdgcn_sched_barrier(0);
      mfma_region(ks, r, r & 1, tr_tag);
      __builtin_amdgcn_sched_barrier(0);
    }
    issue_end();
    if constexpr (TIMING) c2 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S2_PIECES) : "memory");
    if constexpr (TIMING) { t_bar += c1 - c0 + (long long)__builtin_amdgcn_s_memtime() - c2; t_loop += c2 - c1; }
  };
  int si = 0;
  auto tile = [&](int j, bool last, auto tr_tag, auto next_tr) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < S2_NSL; ++ks) period(si + ks, ks, tr_tag);
    si += S2_NSL;
    const long long e0 = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
    epilogue(j, tr_tag);
    if (!last) init_acc(j + 1, next_tr);
    if constexpr (TIMING) t_epi += (long long)__builtin_amdgcn_s_memtime() - e0;
  };

  __syncthreads();
  for (int u = u_begin; u < u_end;) {
    const int rb = u / NT, j0 = u - rb * NT, j1 = min(NT, j0 + (u_end - u));
    m0w = rb * BM + wave * 32;
    const long long l0 = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
    load_a();
    if constexpr (IS_QKV) qkv_offsets();
    // (the first barrier of a row block: this wave's pieces of the next slice -- and everything else it has in flight -- have landed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (TIMING) t_pro += (long long)__builtin_amdgcn_s_memtime() - l0;
    using T = std::true_type; using F = std::false_type;
    if constexpr (IS_QKV) {
      // the q | k tiles, then the v^T tiles, as two loops (one loop over three tile variants: 36 registers in scratch)
      if (j0 < nqk) {
        const int je = min(j1, nqk);
        init_acc(j0, T{});
        for (int j = j0; j < je; ++j) tile(j, j + 1 == je, T{}, T{});
      }
      if (j1 > nqk) {
        const int jb = max(j0, nqk);
        init_acc(jb, F{});
        for (int j = jb; j < j1; ++j) tile(j, j + 1 == j1, F{}, F{});
      }
    } else {
      init_acc(j0, T{});
      for (int j = j0; j < j1; ++j) tile(j, j + 1 == j1, T{}, T{});
    }
    u += j1 - j0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (TIMING) {   // per wave {wait + barrier, row-block prologues, regions, total}; the epilogues = total - the three
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
      d[0] = t_bar; d[1] = t_pro; d[2] = t_loop; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
      (void)t_epi;
    }
  }
}

bool g_x384_split_qkv = getenv("WVN_X384_SPLIT_QKV") != nullptr;   // A/B: q | k and v^T as two launches (the form before the merged kernel)

int x384_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int EPI>
int launch_mx(const X384Params& p, hipStream_t st) {
  if (!p.ln_x || !p.ln_stats || !p.ln_g || !p.ln_b || (p.ln_ldx % 4) || ((uintptr_t)p.ln_x & 15) || ((uintptr_t)p.ln_stats & 7)) return WVN_ERR_ARG;
  const int lds = BIAS_OFF + p.N * 4 + 2 * KD * 4;
  if (lds > X384_LDS_MAX) return WVN_ERR_ARG;
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(X384_LDS_MAX, (const void*)gemm_a384_x3_kernel<EPI, false, true, true>, (const void*)gemm_a384_x3_kernel<EPI, true, true, true>)) return rc;
  const long long units = (long long)ceil_div(p.M, BM) * (p.N / BNT);
  const int grid = (int)(units < x384_num_cus() ? units : x384_num_cus());
  static const bool burst = getenv("WVN_A384_MX_BURST") != nullptr;
  if (burst) {
    static LdsOptIn lds_opt_in_b;
    if (const int rc = lds_opt_in_b(X384_LDS_MAX, (const void*)gemm_a384_x3_kernel<EPI, false, true, true, true>, (const void*)gemm_a384_x3_kernel<EPI, true, true, true, true>)) return rc;
    if (p.dbg) hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI, true, true, true, true>), dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI, false, true, true, true>), dim3(grid), dim3(256), lds, st, p);
    WVN_LAUNCH_CHECK();
    return WVN_OK;
  }
  if (p.dbg) hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI, true, true, true>), dim3(grid), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI, false, true, true>), dim3(grid), dim3(256), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// (form 2: the two-workgroups-per-CU kernel; its weight image lies N * 1536 bytes behind the first one -- backbone.pack_a384_mx)
template <int EPI>
int launch_mx2(X384Params p, hipStream_t st) {
  if (!p.ln_x || !p.ln_stats || !p.ln_g || !p.ln_b || (p.ln_ldx % 4) || ((uintptr_t)p.ln_x & 15) || ((uintptr_t)p.ln_stats & 7)) return WVN_ERR_ARG;
  p.W = (const bf16_t*)((const unsigned char*)p.W + (size_t)p.N * KD * 4);
  const int lds = S2_RING + (EPI == X_QKV_F16 ? 4 * S2_STG : 0) + p.N * 4 + 2 * KD * 4;
  if (lds > 80 * 1024) return WVN_ERR_ARG;
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(80 * 1024, (const void*)gemm_a384_mx2_kernel<EPI>, (const void*)gemm_a384_mx2_kernel<EPI, true>)) return rc;
  const long long units = (long long)ceil_div(p.M, BM) * (p.N / S2_BN);
  static const bool one_per_cu = getenv("WVN_A384_MX2_ONE") != nullptr;   // (experiment: one workgroup per CU -- what a wave does when it has its SIMD to itself)
  const int cap = (one_per_cu ? 1 : 2) * x384_num_cus();
  const int grid = (int)(units < cap ? units : cap);
  if (p.dbg) hipLaunchKernelGGL((gemm_a384_mx2_kernel<EPI, true>), dim3(grid), dim3(256), lds, st, p);   // (dbg: 2 x CUs x 4 waves x 4 counters)
  else hipLaunchKernelGGL((gemm_a384_mx2_kernel<EPI>), dim3(grid), dim3(256), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// 0 (default) / 2: the two-workgroups-per-CU kernel (fc1 9 %, q | k | v^T 5 - 8 % faster); 1: the one-wave-per-SIMD kernel
int g_a384_mx_form = [] { const char* e = getenv("WVN_A384_MX_FORM"); return e && e[0] == '2' ? 2 : (e && e[0] == '1' ? 1 : 0); }();

template <int EPI>
int launch(const X384Params& p, hipStream_t st) {
  constexpr bool CAN_LNA = EPI == X_QKV || EPI == X_QKV_F16 || EPI == X_GELU_FRAG;   // (the LayerNorm-on-load form exists where the block kernels use it)
  const bool lna = p.ln_x != nullptr;
  if (lna && (!CAN_LNA || !p.ln_stats || !p.ln_g || !p.ln_b || (p.ln_ldx % 4) || ((uintptr_t)p.ln_x & 15) || ((uintptr_t)p.ln_stats & 7))) return WVN_ERR_ARG;
  const int lds = BIAS_OFF + p.N * 4 + (lna ? 2 * KD * 4 : 0);
  if (lds > X384_LDS_MAX) return WVN_ERR_ARG;
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(X384_LDS_MAX, (const void*)gemm_a384_x3_kernel<EPI>, (const void*)gemm_a384_x3_kernel<EPI, true>)) return rc;
  const long long units = (long long)ceil_div(p.M, BM) * (p.N / BNT);
  const int grid = (int)(units < x384_num_cus() ? units : x384_num_cus());
  if constexpr (CAN_LNA) {
    if (lna) {
      static LdsOptIn lds_opt_in_lna;
      if (const int rc = lds_opt_in_lna(X384_LDS_MAX, (const void*)gemm_a384_x3_kernel<EPI, false, true>)) return rc;
      hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI, false, true>), dim3(grid), dim3(256), lds, st, p);
      WVN_LAUNCH_CHECK();
      return WVN_OK;
    }
  }
  if (p.dbg) hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI, true>), dim3(grid), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((gemm_a384_x3_kernel<EPI>), dim3(grid), dim3(256), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

// Eligibility: K == 384, N % 64 == 0, stacked weight planes, 16-byte aligned operands, epilogues GELU planes / residual / QKV;
// WVN_ERR_ARG otherwise (the caller uses the tiled gemm_x3 kernel).  Worth it from about a chip of 128-row blocks on.
int wvn_gemm_a384_x3_launch(const GemmBf16Params& g, int epi, hipStream_t st) {
  const bool lna = g.ln_x != nullptr;   // A = LayerNorm(ln_x) on load: the A planes are not read
  if (g.K != KD || g.ldw != KD || (g.N % BNT) != 0 || g.M <= 0 || !g.W || !g.W_lo) return WVN_ERR_ARG;
  if (!lna && (!g.A || !g.A_lo || (g.lda % 8) != 0 || (((uintptr_t)g.A | (uintptr_t)g.A_lo) & 15))) return WVN_ERR_ARG;
  if (((uintptr_t)g.W | (uintptr_t)g.W_lo) & 15) return WVN_ERR_ARG;
  if (g.W_lo <= g.W || (size_t)(g.W_lo - g.W) + (size_t)g.N * KD >= (1ull << 30)) return WVN_ERR_ARG;   // one descriptor over both planes
  X384Params p{};
  p.A = g.A; p.A_lo = g.A_lo; p.lda = g.lda; p.W = g.W; p.w_plane = (size_t)(g.W_lo - g.W); p.bias = g.bias;
  p.C = g.C; p.C_lo = g.C_lo; p.ldc = g.ldc; p.M = g.M; p.N = g.N;
  p.heads = g.heads; p.npad = g.npad; p.ntok_s = g.ntok_s; p.q_scale = g.q_scale != 0.f ? g.q_scale : 1.f; p.f16_out = g.qkv_f16;
  p.ls = g.ls; p.dbg = g.dbg;
  p.ln_x = g.ln_x; p.ln_ldx = g.ln_ldx; p.ln_stats = g.ln_stats; p.ln_g = g.ln_g; p.ln_b = g.ln_b;
  switch (epi) {
    case EPI_GELU_BF16:
      if (!g.C || !g.C_lo || (g.ldc % 8) != 0 || (((uintptr_t)g.C | (uintptr_t)g.C_lo) & 15) || (size_t)g.M * g.ldc * 2 >= (1ull << 31)) return WVN_ERR_ARG;
      return launch<X_GELU>(p, st);
    case EPI_GELU_FRAG:   // fragment-major planes (see the header comment); C / C_lo hold ceil(M / 32) * 32 rows, ldc == N
      if (!g.C || !g.C_lo || g.ldc != g.N || (((uintptr_t)g.C | (uintptr_t)g.C_lo) & 15) || ((size_t)g.M + 32) * g.ldc * 2 >= (1ull << 31)) return WVN_ERR_ARG;
      return launch<X_GELU_FRAG>(p, st);
    case EPI_BF16:
      if (!g.C || !g.C_lo || (g.ldc % 8) != 0 || (((uintptr_t)g.C | (uintptr_t)g.C_lo) & 15) || (size_t)g.M * g.ldc * 2 >= (1ull << 31)) return WVN_ERR_ARG;
      return launch<X_PLANES>(p, st);
    case EPI_RESID_F32:
    case EPI_ACCUM_F32:
      if (!g.C || (g.ldc % 4) != 0 || ((uintptr_t)g.C & 15) || (size_t)g.M * g.ldc * 4 >= (1ull << 31)) return WVN_ERR_ARG;
      return launch<X_RESID>(p, st);
    case EPI_QKV: {
      if (g.N != 3 * g.heads * 64 || !g.q || !g.k || !g.vt || (g.ntok_s % 16) || (g.M % 16) || (g.npad % 16)) return WVN_ERR_ARG;
      if (!g.qkv_f16 && (!g.q_lo || !g.k_lo || !g.vt_lo)) return WVN_ERR_ARG;
      const uintptr_t lo = std::min({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const uintptr_t hi = std::max({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const size_t one = (size_t)(g.M / (g.ntok_s > 0 ? g.ntok_s : 1)) * g.heads * g.npad * 64 * 2;
      if (hi - lo + one >= (1ull << 31)) return WVN_ERR_ARG;
      p.qkv_base = (bf16_t*)lo; p.q_off = (unsigned)((uintptr_t)g.q - lo); p.k_off = (unsigned)((uintptr_t)g.k - lo);
      p.v_off = (unsigned)((uintptr_t)g.vt - lo); p.qkv_bytes = (unsigned)(hi - lo + one);
      if (g.qkv_f16 && g.q_lo) {   // two-plane fp16 q: only q's second plane exists
        if ((uintptr_t)g.q_lo < p.q_off) return WVN_ERR_ARG;
        p.qkv_base_lo = (bf16_t*)((uintptr_t)g.q_lo - p.q_off);
        p.q_lo_f16 = 1;
      }
      if (!g.qkv_f16) {   // the lo planes must sit at the same distances from their base
        const uintptr_t lo2 = std::min({(uintptr_t)g.q_lo, (uintptr_t)g.k_lo, (uintptr_t)g.vt_lo});
        if ((uintptr_t)g.q_lo - lo2 != p.q_off || (uintptr_t)g.k_lo - lo2 != p.k_off || (uintptr_t)g.vt_lo - lo2 != p.v_off) return WVN_ERR_ARG;
        p.qkv_base_lo = (bf16_t*)lo2;
      }
      const int D = g.heads * 64;
      if (!g_x384_split_qkv || lna) return g.qkv_f16 ? launch<X_QKV_F16>(p, st) : launch<X_QKV>(p, st);   // (p.N == 3 D: q | k | v^T in one launch)
      p.N = 2 * D;
      const int rc = g.qkv_f16 ? launch<X_QK_F16>(p, st) : launch<X_QK>(p, st);
      if (rc != WVN_OK) return rc;
      p.W = g.W + (size_t)2 * D * KD;
      p.bias = g.bias ? g.bias + 2 * D : nullptr;
      p.N = D;
      return g.qkv_f16 ? launch<X_V_F16>(p, st) : launch<X_V>(p, st);
    }
    default: return WVN_ERR_ARG;
  }
}

void wvn_gemm_a384_mx_set_form(int form) { g_a384_mx_form = form == 1 || form == 2 ? form : 0; }

// MX form (LayerNorm on load only): W = backbone.pack_a384_mx (plane 0 fp16 [N][384] at g.W, plane 1 bytes [N][768] at g.W_lo, the slice-major image of the
// two-workgroups-per-CU kernel behind them; WVN_A384_MX_FORM=1 / wvn_gemm_a384_mx_set_form(1): the one-wave-per-SIMD kernel; a cycle-counter request takes it too); EPI_GELU_FRAG writes the
// MX operand planes of gemm_n384_x3.hip's MX kernel (g.C fp16 fragments, g.C_lo = l8; h8 is derived by the consumer), EPI_QKV the fp16 q (| q_lo) | k | v^T planes.
int wvn_gemm_a384_mx_launch(const GemmBf16Params& g, int epi, hipStream_t st) {
  if (!g.ln_x || g.K != KD || g.ldw != KD || (g.N % BNT) != 0 || g.M <= 0 || !g.W || !g.W_lo) return WVN_ERR_ARG;
  if (((uintptr_t)g.W | (uintptr_t)g.W_lo) & 15) return WVN_ERR_ARG;
  if (g.W_lo <= g.W || (size_t)(g.W_lo - g.W) + (size_t)g.N * KD >= (1ull << 30)) return WVN_ERR_ARG;
  X384Params p{};
  p.lda = KD; p.W = g.W; p.w_plane = (size_t)(g.W_lo - g.W); p.bias = g.bias;
  p.C = g.C; p.C_lo = g.C_lo; p.C_h8 = g.C_h8; p.ldc = g.ldc; p.M = g.M; p.N = g.N;
  p.heads = g.heads; p.npad = g.npad; p.ntok_s = g.ntok_s; p.q_scale = g.q_scale != 0.f ? g.q_scale : 1.f; p.f16_out = 1;
  p.dbg = g.dbg;
  p.ln_x = g.ln_x; p.ln_ldx = g.ln_ldx; p.ln_stats = g.ln_stats; p.ln_g = g.ln_g; p.ln_b = g.ln_b;
  switch (epi) {
    case EPI_GELU_FRAG:
      if (!g.C || !g.C_lo || g.ldc != g.N || (g.N % 64) || (((uintptr_t)g.C | (uintptr_t)g.C_lo) & 15) || ((size_t)g.M + 32) * g.ldc * 2 >= (1ull << 31))
        return WVN_ERR_ARG;
      return g_a384_mx_form != 1 ? launch_mx2<X_GELU_FRAG>(p, st) : launch_mx<X_GELU_FRAG>(p, st);
    case EPI_QKV: {
      if (!g.qkv_f16 || g.N != 3 * g.heads * 64 || !g.q || !g.k || !g.vt || (g.ntok_s % 16) || (g.M % 16) || (g.npad % 16)) return WVN_ERR_ARG;
      const uintptr_t lo = std::min({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const uintptr_t hi = std::max({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const size_t one = (size_t)(g.M / (g.ntok_s > 0 ? g.ntok_s : 1)) * g.heads * g.npad * 64 * 2;
      if (hi - lo + one >= (1ull << 31)) return WVN_ERR_ARG;
      p.qkv_base = (bf16_t*)lo; p.q_off = (unsigned)((uintptr_t)g.q - lo); p.k_off = (unsigned)((uintptr_t)g.k - lo);
      p.v_off = (unsigned)((uintptr_t)g.vt - lo); p.qkv_bytes = (unsigned)(hi - lo + one);
      if (g.q_lo) {
        if ((uintptr_t)g.q_lo < p.q_off) return WVN_ERR_ARG;
        p.qkv_base_lo = (bf16_t*)((uintptr_t)g.q_lo - p.q_off);
        p.q_lo_f16 = 1;
      }
      return g_a384_mx_form != 1 ? launch_mx2<X_QKV_F16>(p, st) : launch_mx<X_QKV_F16>(p, st);
    }
    default: return WVN_ERR_ARG;
  }
}
