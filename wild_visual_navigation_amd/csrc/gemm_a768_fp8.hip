// A-stationary fp8 (OCP e4m3) GEMM for the K = 768 linears of ViT-Base (BASELINE.json configs[4]: DINOv2 ViT-B/14 -- QKV, attention projection, fc1) on gfx950:
//     C = epilogue((A_q[M,768] * W_q[N,768]^T) * sa[m] * sw[n] + bias[n]),     v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales.
//
// Why: the output-tiled gemm_fp8_kernel reaches 0.11 - 0.17 of the matrix pipe on these shapes (profiles/r06b_dinov2_fp8_pmc.md): with K = 768 a 128 x 128 tile
// has six K-tiles, every one of them moved global -> registers -> LDS behind two workgroup barriers, and an epilogue as long as its MFMA loop.  This is the
// structure of gemm_a384_mx2_kernel (gemm_a384_x3.hip) with one-byte operands:
//   * a workgroup (4 waves) owns 128 rows of A for a run of column tiles; a wave keeps its 32 rows x 768 bytes in registers (96 VGPRs, loaded once per row
//     block, already in MFMA operand layout: lane (row, hi) holds bytes 64 s + 32 hi .. + 31 of every 64-k step s);
//   * a column tile is 32 wide x the whole K = 24 KB of W, packed on the host (backbone.pack_a768_fp8) as 24 chunk images [64 lanes][16 B] in the order the
//     fragment reads take them: a tile is one contiguous 24 KB block, a DMA piece (buffer_load ... lds) one contiguous kilobyte copied lane-linear, a fragment
//     read one contiguous kilobyte at lane * 16 + immediate: no swizzle, no address arithmetic;
//   * three LDS slots: tile j + 2 is requested while tile j is multiplied (12 scaled MFMAs = 768 matrix-pipe cycles per wave, two accumulators alternating,
//     fragments two k-steps ahead in registers) -- with two slots a tile of 0.4 us of MFMAs waited for a weight tile requested 0.4 us earlier; one barrier per tile;
//   * NOTHING in a tile waits for memory it asked for in that tile: the tile's per-column scales, bias, LayerScale and residual rows are requested before its
//     MFMAs, and the wait at the end of a tile lets that tile's own stores and the next-but-one tile's pieces stay in flight
//     (the first build fetched scales and bias inside the epilogue: every tile then waited for its loads BEHIND the next tile's DMA pieces and the previous
//     tile's stores -- 5.5 us per 0.4 us of MFMAs);
//   * <= 256 registers, <= 80 KB of LDS: two workgroups per CU, whose barriers and epilogues cover one another;
//   * epilogues: e4m3 rows with MX block scales (GELU -> the A operand of fc2: EPI_GELU_MX8, no row quantiser between fc1 and fc2), bf16 rows (+ exact-erf GELU) and the q | k | v^T layouts of attention_bf16.hip through a 1 KB wave-private LDS image, half a tile at a time
//     (16-byte stores of 64-byte row pieces), fp32 residual rows (+ LayerScale) read-modified-written directly in 16-byte pieces.
// K != 768, N % 32 != 0 or an un-packed weight: WVN_ERR_ARG (the caller uses gemm_fp8_kernel).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "common.h"
#include "wvn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

constexpr int KD = 768;
constexpr int KS = KD / 64;                  // 12 MFMA k-steps
constexpr int BN = 32;                       // columns per tile
constexpr int BM = 128;
constexpr int TILE_BYTES = BN * KD;          // 24 KB
constexpr int NS = 3;                        // LDS slots
constexpr int PIECES = TILE_BYTES / 1024 / 4;   // 6 per wave and tile
constexpr int STG = 1024;                    // per wave: HALF a bf16 tile image [16 rows][64 B], 16-byte chunks XOR-swizzled by (row >> 2) & 3
constexpr int LDS_BYTES = NS * TILE_BYTES + 4 * STG;   // 76 KB: two workgroups per CU
static_assert(PIECES == 6, "six DMA pieces per wave and tile");

// erf GELU to fp32 rounding with one transcendental (gemm_a384_x3.hip: gelu_pair; tests/test_host_logic.py pins its 2.8e-7 bound)
__device__ inline void gelu_pair8(float& x0, float& x1) {
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
  const f2 m = {__builtin_amdgcn_fmed3f(x0, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(x1, 0.f, 3.0e38f)};
  const f2 g = __builtin_elementwise_fma(-a, e, m);
  x0 = g[0]; x1 = g[1];
}

// (Non-temporal stores and A loads were tried: the weight then stays in the XCD's L2 -- fc1's HBM reads 248 -> 92 MB -- and every kernel is a third SLOWER.
//  Same data-register guard as wvn_store_b128_guarded, common.h.)
__device__ inline void store_b128_nt(u32x4_t v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
  asm volatile("s_nop 1" ::"v"(v));
}

struct A768Params {
  const unsigned char* A; int lda; const float* sa;
  const unsigned char* Wp;              // backbone.pack_a768_fp8: [N / 32 tiles][12 k-steps][2 halves][64 lanes][16 B]
  const float* sw; const float* bias;
  void* C; int ldc;
  int M, N;
  const float* ls;
  unsigned char* c_scales;              // E_GELU_MX8: [M][N / 32] E8M0 bytes
  bf16_t* qkv_base; unsigned q_off, k_off, v_off, qkv_bytes;   // one buffer descriptor over q / k / v^T
  int heads, npad, ntok_s; float q_scale;
  long long* dbg;                       // TIMING build: per wave {barrier, MFMA loop, epilogue, end-of-tile wait, row-block prologue, total} shader cycles
};

enum { E_BF16 = 0, E_GELU = 1, E_RESID = 2, E_QKV = 3, E_GELU_MX8 = 4 };

template <int EPI, bool TIMING = false>
__global__ __launch_bounds__(256, 2) void gemm_a768_fp8_kernel(A768Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int NT = p.N / BN;
  const int NRB = (p.M + BM - 1) / BM;
  const long long U = (long long)NRB * NT;
  const int u_begin = (int)(U * blockIdx.x / gridDim.x), u_end = (int)(U * (blockIdx.x + 1) / gridDim.x);
  unsigned char* stg = smem + NS * TILE_BYTES + wave * STG;
  const unsigned l16 = lane * 16;
  int m0w = 0;
  // ---- W producer ----
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)((size_t)p.N * KD), 0x00020000);
  int iss_j = u_begin % NT;
  unsigned iss_soff = 0;
  auto issue_begin = [&]() __attribute__((always_inline)) { iss_soff = __builtin_amdgcn_readfirstlane((unsigned)iss_j * (unsigned)TILE_BYTES + wave * PIECES * 1024); };
  // (the immediate offset is added to the LDS address too; it is 12 bits wide: pieces 3 - 5 take a second scalar offset)
  unsigned iss_soff_b = 0;
  auto issue = [&](int i, int u) __attribute__((always_inline)) {
    unsigned char* dst = smem + (i % NS) * TILE_BYTES + wave * PIECES * 1024 + (u >= 3 ? 3072 : 0);
    __attribute__((address_space(3))) void* d3 = (__attribute__((address_space(3))) void*)dst;
    const unsigned so = u >= 3 ? iss_soff_b : iss_soff;
    switch (u % 3) {
      case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, so, 0, 0); break;
      case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, so, 1024, 0); break;
      default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, d3, 16, l16, so, 2048, 0); break;
    }
  };
  auto issue_tile_begin = [&]() __attribute__((always_inline)) { issue_begin(); iss_soff_b = iss_soff + 3072; };
  auto issue_end = [&]() __attribute__((always_inline)) { if (++iss_j == NT) iss_j = 0; };
  const int total_tiles = u_end - u_begin;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < total_tiles) {
      issue_tile_begin();
#pragma unroll
      for (int u = 0; u < PIECES; ++u) issue(i, u);
      issue_end();
    }

  // ---- the row block's operands ----
  i32x8_t af[KS];
  float sa_l = 0.f;            // TR tiles: the scale of the lane's row
  f32x4_t sa_r[4];             // V^T tiles: the scales of the rows the lane's registers hold
  auto load_a = [&]() __attribute__((always_inline)) {
    const int row = min(m0w + l31, p.M - 1);
    const unsigned char* ar = p.A + (size_t)row * p.lda + hi * 32;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4_t lo = *(const u32x4_t*)(ar + s * 64), h4 = *(const u32x4_t*)(ar + s * 64 + 16);
      af[s] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
    }
    sa_l = p.sa[row];
    if constexpr (EPI == E_QKV) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) sa_r[g][e] = p.sa[min(m0w + 8 * g + 4 * hi + e, p.M - 1)];
    }
  };

  // ---- epilogue addressing ----
  constexpr unsigned OOB = 0x80000000u;
  const int nqk = EPI == E_QKV ? 2 * (p.N / 3) / BN : (1 << 30);   // QKV: tiles below nqk are q | k (TR), the rest v^T
  const unsigned c_bytes = EPI == E_QKV ? 0u : (unsigned)((size_t)p.M * p.ldc * (EPI == E_RESID ? 4 : (EPI == E_GELU_MX8 ? 1 : 2)));
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(EPI == E_GELU_MX8 ? (void*)p.c_scales : (void*)p.Wp, 0, EPI == E_GELU_MX8 ? (unsigned)((size_t)p.M * (p.N / 32)) : 0u, 0x00020000);
  unsigned soff_row = 0;   // E_GELU_MX8: the lane's row in the scale array
  unsigned sc_acc = 0;     // ... and the scale bytes of the current group of four tiles (one dword store per group: a byte store costs a whole memory sector)
  int run_lo = 0, run_hi = 0;   // the tiles [run_lo, run_hi) this workgroup computes of the current row block
  const __amdgpu_buffer_rsrc_t rs_c = EPI == E_QKV ? __builtin_amdgcn_make_buffer_rsrc(p.qkv_base, 0, p.qkv_bytes, 0x00020000)
                                                   : __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
  unsigned voff[2] = {0, 0}, vt_off = 0, roff = 0;
  auto offsets = [&]() __attribute__((always_inline)) {
    if constexpr (EPI == E_QKV) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int m = m0w + hb * 16 + (lane >> 2);
        const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
        voff[hb] = m < p.M ? (unsigned)((((size_t)b * p.heads * p.npad + tk) * 64 + (lane & 3) * 8) * 2) : OOB;
      }
      const int m = m0w + (lane & 3) * 8;
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      vt_off = m < p.M ? (unsigned)((((size_t)b * p.heads * 64 + (lane >> 2)) * p.npad + tk) * 2) : OOB;
    } else if constexpr (EPI == E_RESID) {
      roff = m0w + l31 < p.M ? (unsigned)(((size_t)(m0w + l31) * p.ldc + 4 * hi) * 4) : OOB;
    } else if constexpr (EPI == E_GELU_MX8) {   // e4m3 rows: 32 bytes a tile = two lanes per row
      const int m = m0w + (lane >> 1);
      voff[0] = m < p.M ? (unsigned)((size_t)m * p.ldc + (lane & 1) * 16) : OOB;
      soff_row = m0w + l31 < p.M ? (unsigned)((size_t)(m0w + l31) * (p.N / 32)) : OOB;
    } else {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int m = m0w + hb * 16 + (lane >> 2);
        voff[hb] = m < p.M ? (unsigned)(((size_t)m * p.ldc + (lane & 3) * 8) * 2) : OOB;
      }
    }
  };

  u32x4_t resid[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // E_RESID: the tile's residual rows, requested before its MFMAs
  // the tile's per-column vectors, requested before its MFMAs too (three ring slots leave no room for them in LDS): TR tiles, register 4 g + e <-> column 8 g + 4 hi + e
  f32x4_t sw4[4], bi4[4], ls4[4];
  float sw1 = 0.f, bi1 = 0.f;   // V^T tiles: the lane's column
  auto load_vectors = [&](int j, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = j * BN;
    if constexpr (TR) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sw4[g] = *(const f32x4_t*)(p.sw + n0 + 8 * g + 4 * hi);
        bi4[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bi4[g] = *(const f32x4_t*)(p.bias + n0 + 8 * g + 4 * hi);
        if constexpr (EPI == E_RESID) {
          ls4[g] = f32x4_t{1.f, 1.f, 1.f, 1.f};
          if (p.ls) ls4[g] = *(const f32x4_t*)(p.ls + n0 + 8 * g + 4 * hi);
        }
      }
    } else {
      sw1 = p.sw[n0 + l31];
      bi1 = p.bias ? p.bias[n0 + l31] : 0.f;
    }
  };
  f32x16_t acc[2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  };

  // ---- the tile's epilogue ----
  auto stage_and_store = [&](const uint32_t (&h)[8], unsigned so, bool vt_layout, int nv) __attribute__((always_inline)) {
    // h[2 g + k]: the bf16 pair of image columns c, c + 1 (c = 8 g + 4 hi + 2 k for row-major tiles; the permuted token position for V^T tiles) of image row l31.
    // Half an image at a time (16 rows = 1 KB): the lanes of the half write their four 8-byte pieces, then every lane reads 16 bytes (4 lanes per row) and stores
    const int r16 = l31 & 15, f = (r16 >> 2) & 3;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if ((l31 >> 4) == it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c8 = vt_layout ? (16 * (g >> 1) + 8 * hi + 4 * (g & 1)) * 2 : (8 * g + 4 * hi) * 2;   // byte offset of the piece inside the 64-byte image row
          *(u32x2_t*)(stg + r16 * 64 + (((c8 >> 4) ^ f) << 4) + (c8 & 15)) = u32x2_t{h[2 * g], h[2 * g + 1]};
        }
      }
      const int row = lane >> 2;
      const u32x4_t val = *(const u32x4_t*)(stg + row * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4));
      if (!vt_layout) store_b128_nt(val, rs_c, voff[it], so);
      else store_b128_nt(val, rs_c, vt_off, __builtin_amdgcn_readfirstlane(so + (unsigned)((nv + it * 16) * p.npad * 2)));
    }
  };
  auto epilogue = [&](int j, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = j * BN;
    f32x16_t v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[0][r] + acc[1][r];
    if constexpr (TR) {
      // lane = row m (scale sa_l), register 4 g + e = column n0 + 8 g + 4 hi + e
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t s4 = sw4[g], b4 = bi4[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaf(v[4 * g + e], sa_l * s4[e], b4[e]);
      }
      if constexpr (EPI == E_RESID) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4_t o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          o *= ls4[g];
          const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((n0 + 8 * g) * 4));
          const u32x4_t r = resid[g];
          u32x4_t w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(o[e] + __uint_as_float(r[e]));
          store_b128_nt(w, rs_c, roff, so);
        }
      } else if constexpr (EPI == E_GELU_MX8) {
        // the tile's 32 columns of a row = ONE MX block of the next product's K: E8M0 scale 2^(floor(log2 amax) - 8) (OCP MX: e4m3's largest exponent is 8, elements
        // above 448 x scale saturate), elements e4m3 by the hardware converter; the row's two lanes (hi = 0 | 1) hold 16 values each
        float am = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float x0 = v[2 * k], x1 = v[2 * k + 1];
          gelu_pair8(x0, x1);
          v[2 * k] = x0; v[2 * k + 1] = x1;
          am = fmaxf(am, fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)));
        }
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const int sb = max((int)((__float_as_uint(am) >> 23) & 0xff) - 8, 0);     // the scale byte: 2^(sb - 127)
        const float inv = __uint_as_float((unsigned)(254 - sb) << 23);             // 2^(127 - sb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(v[4 * g + e] * inv, -448.f, 448.f);
          int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
          *(int*)(stg + l31 * 32 + 8 * g + 4 * hi) = w;
        }
        const u32x4_t val = *(const u32x4_t*)(stg + (lane >> 1) * 32 + (lane & 1) * 16);
        store_b128_nt(val, rs_c, voff[0], __builtin_amdgcn_readfirstlane((unsigned)n0));
        // the scale bytes of four consecutive tiles leave as ONE dword (2.1 million byte stores per launch cost 67 - 135 MB of sector traffic and their issue slots); a
        // group the run covers only partly leaves as bytes.  Every tile issues at least one scale store instruction (an out-of-range one where the group is not
        // complete yet): the end-of-tile wait counts on two stores per epilogue
        if ((j & 3) == 0 || j == run_lo) sc_acc = 0;
        sc_acc |= (unsigned)sb << (8 * (j & 3));
        const unsigned srow = hi == 0 ? soff_row : OOB;
        if ((j & 3) == 3 && j - 3 >= run_lo && ((p.N / 32) & 3) == 0) {   // (rows of the scale array a multiple of four bytes long: the dword is aligned)
          __builtin_amdgcn_raw_buffer_store_b32(sc_acc, rs_s, srow, __builtin_amdgcn_readfirstlane((unsigned)(j - 3)), 0);
        } else if ((j & 3) == 3 || j + 1 == run_hi) {
          for (int t = max(j & ~3, run_lo); t <= j; ++t)
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(sc_acc >> (8 * (t & 3))), rs_s, srow, __builtin_amdgcn_readfirstlane((unsigned)t), 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, rs_s, OOB, 0, 0);
        }
      } else {
        float qs = 1.f;
        unsigned so;
        if constexpr (EPI == E_QKV) {
          const int D = p.N / 3;
          const int which = n0 / D, head = (n0 - which * D) >> 6, t = (n0 >> 5) & 1;
          if (which == 0 && p.q_scale != 0.f) qs = p.q_scale;
          so = __builtin_amdgcn_readfirstlane((which == 0 ? p.q_off : p.k_off) + (unsigned)(head * p.npad * 64 * 2 + t * 64));
        } else {
          so = __builtin_amdgcn_readfirstlane((unsigned)(n0 * 2));
        }
        uint32_t h[8];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            float x0 = v[4 * g + 2 * k] * qs, x1 = v[4 * g + 2 * k + 1] * qs;
            if constexpr (EPI == E_GELU) gelu_pair8(x0, x1);
            h[2 * g + k] = pack_bf16x2(x0, x1);
          }
        stage_and_store(h, so, false, 0);
      }
    } else {
      // V^T: lane = column n = n0 + l31 (scale sw, bias), register 4 g + e = row m0w + 8 g + 4 hi + e (scale sa_r)
      const float sl = sw1, bl = bi1;
      uint32_t h[8];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          h[2 * g + k] = pack_bf16x2(fmaf(v[4 * g + 2 * k], sl * sa_r[g][2 * k], bl), fmaf(v[4 * g + 2 * k + 1], sl * sa_r[g][2 * k + 1], bl));
      stage_and_store(h, p.v_off, true, n0 - 2 * (p.N / 3));
    }
  };

  // ---- one tile: barrier (every wave has waited for its pieces of this tile at the end of the previous one), the DMA requests of the next tile riding between
  // the twelve MFMAs, the wait for them, the epilogue ----
  i32x8_t wf[3];
  constexpr int NST = EPI == E_RESID ? 4 : 2;   // stores of one epilogue per lane
  auto frag_read = [&](int slot, int s, int set) __attribute__((always_inline)) {
    const unsigned char* base = smem + slot * TILE_BYTES + l16 + s * 2048;
    const u32x4_t lo = *(const u32x4_t*)base, h4 = *(const u32x4_t*)(base + 1024);
    wf[set] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
  };
  int ti = 0;
  long long t_bar = 0, t_mm = 0, t_epi = 0, t_wait = 0, t_pro = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto tile = [&](int j, auto tr_tag) __attribute__((always_inline)) {
    constexpr bool TR = decltype(tr_tag)::value;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
    issue_tile_begin();
    load_vectors(j, tr_tag);
    if constexpr (EPI == E_RESID) {
#pragma unroll
      for (int g = 0; g < 4; ++g) resid[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, roff, __builtin_amdgcn_readfirstlane((unsigned)((j * BN + 8 * g) * 4)), 0);
    }
    frag_read(ti % NS, 0, 0);
    frag_read(ti % NS, 1, 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 2 < KS) frag_read(ti % NS, s + 2, (s + 2) % 3);
      if ((s & 1) == 0) issue(ti + NS - 1, s >> 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TR) acc[s & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[s % 3], af[s], acc[s & 1], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else acc[s & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[s], wf[s % 3], acc[s & 1], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      asm volatile("" : "+v"(acc[s & 1]));   // (pins the MFMA here: its only use is the epilogue, and left free all twelve sink behind the wait below -- with all twelve W fragments live)
      __builtin_amdgcn_sched_barrier(0);
    }
    issue_end();
    ++ti;
    if constexpr (TIMING) { asm volatile("" : "+v"(acc[0]), "+v"(acc[1])); c2 = (long long)__builtin_amdgcn_s_memtime(); }
    epilogue(j, tr_tag);
    zero_acc();
    if constexpr (TIMING) c3 = (long long)__builtin_amdgcn_s_memtime();
    // the NEXT tile's pieces (requested a tile ago) have landed: everything but the youngest PIECES + NST operations of the queue -- the pieces of the tile after it,
    // requested between this tile's MFMAs, and this epilogue's own stores; the previous tile's stores and this tile's vectors / residual rows (consumed above) are older
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + NST) : "memory");
    if constexpr (TIMING) { t_bar += c1 - c0; t_mm += c2 - c1; t_epi += c3 - c2; t_wait += (long long)__builtin_amdgcn_s_memtime() - c3; }
  };

  for (int u = u_begin; u < u_end;) {
    const int rb = u / NT, j0 = u - rb * NT, j1 = min(NT, j0 + (u_end - u));
    m0w = rb * BM + wave * 32;
    const long long l0 = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
    load_a();
    offsets();
    zero_acc();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the first barrier of a row block: this wave's pieces of the coming tile have landed)
    if constexpr (TIMING) { asm volatile("" : "+v"(af[0]), "+v"(af[11])); t_pro += (long long)__builtin_amdgcn_s_memtime() - l0; }
    using T = std::true_type; using F = std::false_type;
    if (j0 < nqk) {
      const int je = min(j1, nqk);
      run_lo = j0; run_hi = je;
      for (int j = j0; j < je; ++j) tile(j, T{});
    }
    if constexpr (EPI == E_QKV) {
      if (j1 > nqk)
        for (int j = max(j0, nqk); j < j1; ++j) tile(j, F{});
    }
    u += j1 - j0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last tile's surplus request lands before the wave ends)
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
      d[0] = t_bar; d[1] = t_mm; d[2] = t_epi; d[3] = t_wait; d[4] = t_pro; d[5] = (long long)__builtin_amdgcn_s_memtime() - t_start; d[6] = u_end - u_begin;
    }
  }
}

int num_cus() {
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
int launch(const A768Params& p, hipStream_t st) {
  const long long units = (long long)ceil_div(p.M, BM) * (p.N / BN);
  static const int per_cu = [] { const char* e = getenv("WVN_A768_WG_PER_CU"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 2 ? v : 2; }();
  const long long cap = (long long)per_cu * num_cus();
  const int grid = (int)(units < cap ? units : cap);
  const int lds = LDS_BYTES;
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(80 * 1024, (const void*)gemm_a768_fp8_kernel<EPI>, (const void*)gemm_a768_fp8_kernel<EPI, true>)) return rc;
  static const bool dbg = getenv("WVN_A768_DEBUG") != nullptr;
  if (dbg) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm_a768_fp8_kernel<EPI>, 256, lds);
    fprintf(stderr, "gemm_a768_fp8<%d>: M %d N %d grid %d lds %d -> %d workgroups per CU\n", EPI, p.M, p.N, grid, lds, nb);
  }
  static const bool timing = getenv("WVN_A768_TIMING") != nullptr;   // (experiment: in-kernel cycle counters of one launch, printed to stderr)
  if (timing) {
    A768Params q = p;
    const size_t n = (size_t)grid * 4 * 8;
    if (hipMalloc((void**)&q.dbg, n * 8) != hipSuccess) return WVN_ERR_ARG;
    (void)hipMemsetAsync(q.dbg, 0, n * 8, st);
    hipLaunchKernelGGL((gemm_a768_fp8_kernel<EPI, true>), dim3(grid), dim3(256), lds, st, q);
    std::vector<long long> h(n);
    (void)hipMemcpyAsync(h.data(), q.dbg, n * 8, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(q.dbg);
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n / 8; ++i) for (int k = 0; k < 7; ++k) a[k] += (double)h[i * 8 + k];
    const double tiles = a[6];
    fprintf(stderr, "gemm_a768_fp8<%d> N %d: per wave and tile: barrier %.0f, MFMA loop %.0f (768 of MFMAs), epilogue %.0f, end wait %.0f, prologue %.0f, total %.0f cycles (%.1f tiles per wave)\n",
            EPI, p.N, a[0] / tiles, a[1] / tiles, a[2] / tiles, a[3] / tiles, a[4] / tiles, a[5] / tiles, tiles / (n / 8));
    return WVN_OK;
  }
  hipLaunchKernelGGL((gemm_a768_fp8_kernel<EPI>), dim3(grid), dim3(256), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

// Eligibility: K == 768 (lda == 768 bytes a row or more), N % 32 == 0 (QKV: (N / 3) % 64 == 0), Wp = backbone.pack_a768_fp8 of the e4m3 weight, 16-byte aligned
// operands; WVN_ERR_ARG otherwise.  Worth it from a few thousand rows on.
int wvn_gemm_a768_fp8_launch(const GemmFp8Params& g, const void* Wp, int epi, hipStream_t st) {
  const bool off = getenv("WVN_NO_A768_FP8") != nullptr;   // (read per launch: tests/test_gpu_fp8.py switches it inside one process)
  if (off || !Wp || g.K != KD || (g.N % BN) != 0 || g.M <= 0 || !g.A || !g.sa || !g.sw || (g.lda % 16) != 0 || (((uintptr_t)g.A | (uintptr_t)Wp) & 15)) return WVN_ERR_ARG;
  if (((uintptr_t)g.sw & 15) || (g.bias && ((uintptr_t)g.bias & 15)) || (g.ls && ((uintptr_t)g.ls & 15))) return WVN_ERR_ARG;
  A768Params p{};
  p.A = g.A; p.lda = g.lda; p.sa = g.sa; p.Wp = (const unsigned char*)Wp; p.sw = g.sw; p.bias = g.bias; p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N; p.ls = g.ls;
  switch (epi) {
    case EPI_BF16:
    case EPI_GELU_BF16:
      if (!g.C || (g.ldc % 8) != 0 || ((uintptr_t)g.C & 15) || (size_t)g.M * g.ldc * 2 >= (1ull << 31)) return WVN_ERR_ARG;
      return epi == EPI_BF16 ? launch<E_BF16>(p, st) : launch<E_GELU>(p, st);
    case EPI_GELU_MX8:
      if (!g.C || !g.c_scales || (g.ldc % 16) != 0 || ((uintptr_t)g.C & 15) || (size_t)g.M * g.ldc >= (1ull << 31)) return WVN_ERR_ARG;
      p.c_scales = g.c_scales;
      return launch<E_GELU_MX8>(p, st);
    case EPI_RESID_F32:
      if (!g.C || (g.ldc % 4) != 0 || ((uintptr_t)g.C & 15) || (size_t)g.M * g.ldc * 4 >= (1ull << 31)) return WVN_ERR_ARG;
      return launch<E_RESID>(p, st);
    case EPI_QKV: {
      if ((g.N % 3) != 0 || ((g.N / 3) % 64) != 0 || g.N / 3 != g.heads * 64 || !g.q || !g.k || !g.vt || (g.ntok_s % 16) || (g.M % 16) || (g.npad % 16)) return WVN_ERR_ARG;
      const uintptr_t lo = std::min({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const uintptr_t hi = std::max({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const size_t one = (size_t)(g.M / (g.ntok_s > 0 ? g.ntok_s : 1)) * g.heads * g.npad * 64 * 2;
      if (hi - lo + one >= (1ull << 31)) return WVN_ERR_ARG;
      p.qkv_base = (bf16_t*)lo; p.q_off = (unsigned)((uintptr_t)g.q - lo); p.k_off = (unsigned)((uintptr_t)g.k - lo); p.v_off = (unsigned)((uintptr_t)g.vt - lo);
      p.qkv_bytes = (unsigned)(hi - lo + one);
      p.heads = g.heads; p.npad = g.npad; p.ntok_s = g.ntok_s; p.q_scale = g.q_scale;
      return launch<E_QKV>(p, st);
    }
    default: return WVN_ERR_ARG;
  }
}
