// fp8 (OCP e4m3) MFMA GEMM for the ViT linears on gfx950 -- BASELINE.json configs[4] (DINOv2 ViT-B/14, 518 x 518, fp8 MFMA
// backbone):  C = epilogue((A_q[M,K] * W_q[N,K]^T) * sa[m] * sw[n] + bias[n]).
//
// Operands are 1-byte e4m3 with one fp32 scale per ROW of A (per token: written by the LayerNorm / row-quantise kernels of
// fp8.hip, amax / 448) and per ROW of W (per output channel, quantised when the model is packed).  The products run on
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 = 127): K = 64 per instruction at 2x the bf16 MFMA rate
// (MI355X_MICROARCH.md: non-scaled fp8 MFMA runs at the bf16 rate; the scaled K = 64 / 128 forms are the only path to the fp8
// peak), fp32 accumulation; the two scale vectors and the bias are applied in the epilogue.
//
// Byte-for-byte the tile geometry of gemm_bf16.hip: 128 (M) x 128 (N) x 128 (K) tile = 128-byte operand rows, 4 waves (2 x 2),
// each wave 64 x 64 = 2 x 2 MFMA tiles, LDS 2 stages x (A + W) x 128 rows x 144 B = 73,728 B (2 workgroups / CU), register
// prefetch of the next K-tile, XCD-aware tile order, epilogue staged through the operand LDS as a row-major image and written
// with 16-byte coalesced stores.  A fragment = the lane's row, bytes [16 hi, 16 hi + 16) and [32 + 16 hi, 32 + 16 hi + 16) of each 64-k step: the hardware's scale block 0 of
// the instruction is registers 0 - 3 of BOTH half-waves, so this is the mapping under which block kb = k [32 kb, 32 kb + 32) (scripts/ubench/mx_formats.hip).
// AMX (round 6): the A operand carries MX block scales -- one E8M0 byte per (row, 32 k), p.a_scales [M][K / 32] -- instead of the per-row fp32 scale sa: what the fc1
// epilogue of gemm_a768_fp8.hip writes (a 32-column tile of the hidden activation IS one scale block of fc2's K), so no row quantiser runs between fc1 and fc2.
// The lane (row, hi) supplies the scale of its row's block hi in byte 0 of the scale operand.
#include "common.h"
#include "wvn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

constexpr int BM = 128, BN = 128, BK = 128;   // BK in fp8 elements = bytes
constexpr int LDS_STRIDE = BK + 16;                      // bytes per LDS row (144 B)
constexpr int STAGE_ELEMS = (BM + BN) * LDS_STRIDE;      // per stage
constexpr int GEMM_LDS_BYTES = 2 * STAGE_ELEMS;          // 73,728 B
constexpr int SCALE_LDS_OFF = GEMM_LDS_BYTES;            // AMX: 2 stages x 128 rows x 4 scale bytes of the K-tile
constexpr int GEMM_LDS_BYTES_AMX = GEMM_LDS_BYTES + 2 * BM * 4;
constexpr int CT_BF16_STRIDE = 128 + 8;                  // output-tile image, bf16 elements per row (272 B)
constexpr int CT_F32_STRIDE = 128 + 4;                   // output-tile image, floats per row (528 B)
static_assert(128 * CT_F32_STRIDE * 4 <= GEMM_LDS_BYTES, "fp32 tile image must fit in the operand LDS");

// erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below bf16 resolution of the output)
__device__ inline float gelu_bf16path(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

template <int EPI>
__device__ inline float activate(float v) {
  if constexpr (EPI == EPI_GELU_BF16) return gelu_bf16path(v);
  if constexpr (EPI == EPI_RELU_BF16) return fmaxf(v, 0.f);
  return v;
}

template <int EPI>
constexpr bool out_is_bf16() {
  return EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || EPI == EPI_QKV;
}

// TR = true : accumulators hold C^T (lane = row m, regs = cols n)  -> LDS image [m][n]
// TR = false: accumulators hold C   (lane = col n, regs = rows m)  -> LDS image [n][m]   (V^T tiles)
template <int EPI, bool TR, bool AMX = false>
__device__ inline void gemm_fp8_tile(const GemmFp8Params& p, int tm, int tn, unsigned char* smem) {
  unsigned char* lds = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;

  // Staging: 4 A chunks + 4 W chunks of 16 B per thread per K-tile, held in one of two register sets (two K-tiles in
  // flight per workgroup); LDS is double-buffered: tile t+1 moves registers -> LDS right after the MFMAs of tile t.
  // Loads are branch-free: rows past M / N are clamped to the last valid row (their results are never stored; an output
  // element depends only on its own A row and its own W row).
  u32x4_t ra[2][4], rb[2][4];
  unsigned rsc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // AMX: the four scale bytes of the K-tile for the thread's four staging rows
  const int srow = tid >> 3, skc = tid & 7;
  const unsigned char* pa[4];
  const unsigned char* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pa[i] = p.A + (size_t)min(m0 + srow + 32 * i, p.M - 1) * p.lda + skc * 16;
    pb[i] = p.W + (size_t)min(n0 + srow + 32 * i, p.N - 1) * p.ldw + skc * 16;
  }
  const int nblk = p.K / 32;   // AMX: scale bytes per row
  auto load_regs = [&](int kt, u32x4_t (&a)[4], u32x4_t (&b)[4], unsigned (&sc)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *(const u32x4_t*)(pa[i] + kt * BK);
      b[i] = *(const u32x4_t*)(pb[i] + kt * BK);
      if constexpr (AMX) sc[i] = *(const unsigned*)(p.a_scales + (size_t)min(m0 + srow + 32 * i, p.M - 1) * nblk + kt * 4);   // (eight threads per row: one request)
    }
  };
  auto store_regs = [&](int stage, const u32x4_t (&a)[4], const u32x4_t (&b)[4], const unsigned (&sc)[4]) {
    unsigned char* As = lds + stage * STAGE_ELEMS;
    unsigned char* Bs = As + BM * LDS_STRIDE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = srow + 32 * i;
      *(u32x4_t*)(As + row * LDS_STRIDE + skc * 16) = a[i];
      *(u32x4_t*)(Bs + row * LDS_STRIDE + skc * 16) = b[i];
      if constexpr (AMX) { if (skc == 0) *(unsigned*)(smem + SCALE_LDS_OFF + (stage * BM + row) * 4) = sc[i]; }
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* As = lds + stage * STAGE_ELEMS;
    const unsigned char* Bs = As + BM * LDS_STRIDE;
    const unsigned char* a_base = As + (wm * 64 + l31) * LDS_STRIDE + hi * 16;
    const unsigned char* b_base = Bs + (wn * 64 + l31) * LDS_STRIDE + hi * 16;
    unsigned scw[2] = {0x7f7f7f7fu, 0x7f7f7f7fu};   // AMX: the K-tile's four scale bytes of the lane's row (per i)
    if constexpr (AMX) {
#pragma unroll
      for (int i = 0; i < 2; ++i) scw[i] = *(const unsigned*)(smem + SCALE_LDS_OFF + (stage * BM + wm * 64 + i * 32 + l31) * 4);
    }
#pragma unroll
    for (int s = 0; s < BK / 64; ++s) {
      i32x8_t af[2], bfr[2];
      int sca[2] = {0x7f7f7f7f, 0x7f7f7f7f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4_t lo = *(const u32x4_t*)(a_base + i * 32 * LDS_STRIDE + s * 64), hi4 = *(const u32x4_t*)(a_base + i * 32 * LDS_STRIDE + s * 64 + 32);
        af[i] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        if constexpr (AMX) sca[i] = (int)(scw[i] >> (8 * (2 * s + hi)));   // byte 0 = the scale of block 2 s + hi of the K-tile (the instruction reads byte 0 of lane (row, hi) for block hi)
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4_t lo = *(const u32x4_t*)(b_base + j * 32 * LDS_STRIDE + s * 64), hi4 = *(const u32x4_t*)(b_base + j * 32 * LDS_STRIDE + s * 64 + 32);
        bfr[j] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // cbsz = blgp = 0: both operands e4m3; block scales 127 = 2^0 (the per-row scales are applied in the epilogue) -- AMX: the activation's E8M0 block scales
          if constexpr (TR)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, sca[i]);
          else
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bfr[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
  };

  // register prefetch two K-tiles deep, branch-free (the K-tile index is clamped, so hipcc counts vmcnt exactly: it waits
  // only for the tile it is about to move to LDS); the loop is unrolled by two so register slots and LDS stages are
  // compile-time.  (Fully unrolled K pipelines as in gemm_bf16.hip spilled 176 VGPRs here: the 8-VGPR fragments.)
  const int nk = p.K / BK;
  auto load_c = [&](int kt, u32x4_t (&a)[4], u32x4_t (&b)[4], unsigned (&sc)[4]) { load_regs(min(kt, nk - 1), a, b, sc); };
  load_c(0, ra[0], rb[0], rsc[0]);
  load_c(1, ra[1], rb[1], rsc[1]);
  store_regs(0, ra[0], rb[0], rsc[0]);
  load_c(2, ra[0], rb[0], rsc[0]);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    compute(0);                              // tile kt
    store_regs(1, ra[1], rb[1], rsc[1]);     // tile kt + 1 (or a clamped re-load of the last tile: then never read)
    load_c(kt + 3, ra[1], rb[1], rsc[1]);
    __syncthreads();
    if (kt + 1 < nk) compute(1);             // tile kt + 1 (block-uniform)
    store_regs(0, ra[0], rb[0], rsc[0]);     // tile kt + 2
    load_c(kt + 4, ra[0], rb[0], rsc[0]);
    __syncthreads();                         // also: after the last K-tile every wave is done with the operand LDS
  }

  // ---------------- epilogue, part 1: registers -> LDS tile image (bias + activation applied) ----------
  // image row = "lane" dimension, image col = "register" dimension (4 consecutive per register group)
  constexpr bool OB = out_is_bf16<EPI>();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lane_dim = TR ? (wm * 64 + i * 32 + l31) : (wn * 64 + j * 32 + l31);
      const int reg_base = TR ? (wn * 64 + j * 32) : (wm * 64 + i * 32);
      float bl = 0.f, sl = 1.f;   // lane-dimension bias / scale: TR: sa of the lane's row; !TR: sw and bias of the lane's column
      if constexpr (!TR) {
        if (n0 + lane_dim < p.N) { bl = p.bias ? p.bias[n0 + lane_dim] : 0.f; sl = p.sw[n0 + lane_dim]; }
      } else if constexpr (!AMX) {
        sl = p.sa[min(m0 + lane_dim, p.M - 1)];
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        int c = reg_base + 8 * g4 + 4 * hi;
        if constexpr (EPI == EPI_QKV && !TR)  // V^T: tokens permuted inside aligned groups of 16 (bits 2 <-> 3), see attention_bf16.hip
          c = reg_base + 16 * (g4 >> 1) + 8 * hi + 4 * (g4 & 1);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float b = bl, sr;      // register-dimension scale: TR: sw of the column; !TR: sa of the row
          if constexpr (TR) {
            b = (p.bias && n0 + c + e < p.N) ? p.bias[n0 + c + e] : 0.f;
            sr = n0 + c + e < p.N ? p.sw[n0 + c + e] : 0.f;
          } else {
            sr = p.sa[min(m0 + 8 * g4 + 4 * hi + e + (wm * 64 + i * 32), p.M - 1)];   // the TRUE row of this register (c is the permuted V^T position)
          }
          v[e] = activate<EPI>(fmaf(acc[i][j][4 * g4 + e], sl * sr, b));
          if constexpr (EPI == EPI_RESID_F32 && TR) {
            if (p.ls) v[e] *= (n0 + c + e < p.N) ? p.ls[n0 + c + e] : 0.f;  // LayerScale (DINOv2): x += ls * (acc + bias)
          }
          if constexpr (EPI == EPI_QKV && TR) {
            if (p.q_scale != 0.f && n0 < p.N / 3) v[e] *= p.q_scale;  // q third (tile-uniform): softmax scale folded in
          }
        }
        if constexpr (OB) {
          u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *(u32x2_t*)((bf16_t*)smem + lane_dim * CT_BF16_STRIDE + c) = o;
        } else {
          f32x4_t o = {v[0], v[1], v[2], v[3]};
          *(f32x4_t*)((float*)smem + lane_dim * CT_F32_STRIDE + c) = o;
        }
      }
    }
  __syncthreads();

  // ---------------- epilogue, part 2: LDS image -> global, 16-byte coalesced --------------------------
  if constexpr (EPI == EPI_QKV) {
    const int D = p.N / 3;
    const int which = n0 / D;  // tile-uniform (D % 128 == 0)
    const int cbase = n0 - which * D;
    if constexpr (TR) {  // q / k : image [m][n]; dst[(b*h + head)*npad + t][d]
      bf16_t* dst = which == 0 ? p.q : p.k;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const int b = m / p.ntok_s, t = m - b * p.ntok_s;
        const int cc = cbase + c8, head = cc >> 6, d = cc & 63;
        const u32x4_t val = *(const u32x4_t*)((const bf16_t*)smem + row * CT_BF16_STRIDE + c8);
        *(u32x4_t*)(dst + (((size_t)b * p.heads + head) * p.npad + t) * 64 + d) = val;
      }
    } else {  // v : image [n = (head, d)][m]; vt[(b*h + head)*64 + d][t], 8 tokens per store
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
        const int m = m0 + c8;
        if (m >= p.M) continue;  // M % 16 == 0 (ntok_s % 16 == 0): a chunk (and its permutation group of 16) is entirely in or out
        const int b = m / p.ntok_s, t = m - b * p.ntok_s;
        const int cc = cbase + row, head = cc >> 6, d = cc & 63;
        const u32x4_t val = *(const u32x4_t*)((const bf16_t*)smem + row * CT_BF16_STRIDE + c8);
        *(u32x4_t*)(p.vt + (((size_t)b * p.heads + head) * 64 + d) * p.npad + t) = val;
      }
    }
  } else if constexpr (OB) {
    bf16_t* C = (bf16_t*)p.C;
    const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)C & 15) == 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m >= p.M || n >= p.N) continue;
      const bf16_t* src = (const bf16_t*)smem + row * CT_BF16_STRIDE + c8;
      if (vec_ok && n + 8 <= p.N) {
        *(u32x4_t*)(C + (size_t)m * p.ldc + n) = *(const u32x4_t*)src;
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) C[(size_t)m * p.ldc + n + e] = src[e];
      }
    }
  } else {
    float* C = (float*)p.C;
    const bool vec_ok = ((p.ldc & 3) == 0) && (((uintptr_t)C & 15) == 0);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int ch = tid + 256 * it, row = ch >> 5, c4 = (ch & 31) * 4;
      const int m = m0 + row, n = n0 + c4;
      if (m >= p.M || n >= p.N) continue;
      f32x4_t v = *(const f32x4_t*)((const float*)smem + row * CT_F32_STRIDE + c4);
      size_t orow = (size_t)m;
      if constexpr (EPI == EPI_PATCH) {
        const int b = m / p.npatch, pp = m - b * p.npatch;
        orow = (size_t)b * p.ntok_s + 1 + pp;
        const f32x4_t pe = *(const f32x4_t*)(p.pos + (size_t)(1 + pp) * p.ldc + n);  // ldc == D, n % 4 == 0
        v += pe;
      }
      float* dst = C + orow * p.ldc + n;
      if (vec_ok && n + 4 <= p.N) {
        if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_ACCUM_F32) v += *(const f32x4_t*)dst;
        *(f32x4_t*)dst = v;
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) {
          float o = v[e];
          if constexpr (EPI == EPI_RESID_F32 || EPI == EPI_ACCUM_F32) o += dst[e];
          dst[e] = o;
        }
      }
    }
  }
}

template <int EPI, bool AMX = false>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(GemmFp8Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if constexpr (EPI == EPI_QKV) {
    if (tn * BN >= 2 * (p.N / 3)) {  // block-uniform: the V third is produced as V^T
      gemm_fp8_tile<EPI, false>(p, tm, tn, smem);
      return;
    }
  }
  gemm_fp8_tile<EPI, true, AMX>(p, tm, tn, smem);
}

template <int EPI, bool AMX = false>
int launch_epi(const GemmFp8Params& p, hipStream_t st) {
  constexpr int LDS = AMX ? GEMM_LDS_BYTES_AMX : GEMM_LDS_BYTES;
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(LDS, (const void*)gemm_fp8_kernel<EPI, AMX>)) return rc;
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI, AMX>), dim3(tiles), dim3(256), LDS, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

int wvn_gemm_fp8_launch(const GemmFp8Params& p, int epi, hipStream_t st) {
  if ((epi == EPI_F32 || epi == EPI_RESID_F32) && p.K >= 1024) {   // long K: the DMA-fed kernel at three workgroups per CU where the shape is its
    const int rc = wvn_gemm_fp8_dma_launch(p, epi, st);
    if (rc != WVN_ERR_ARG) return rc;
  }
  if (p.a_scales) {   // MX block scales on the A operand (what gemm_a768_fp8.hip's GELU epilogue writes): the residual epilogue of fc2
    if (!p.A || !p.W || !p.sw || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.lda % 16) != 0 || (p.ldw % 16) != 0) return WVN_ERR_ARG;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.a_scales & 3)) return WVN_ERR_ARG;
    if (epi == EPI_RESID_F32) return p.C ? launch_epi<EPI_RESID_F32, true>(p, st) : WVN_ERR_ARG;
    if (epi == EPI_F32) return p.C ? launch_epi<EPI_F32, true>(p, st) : WVN_ERR_ARG;
    return WVN_ERR_ARG;
  }
  if (!p.A || !p.W || !p.sa || !p.sw || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.lda % 16) != 0 || (p.ldw % 16) != 0)
    return WVN_ERR_ARG;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return WVN_ERR_ARG;
  switch (epi) {
    case EPI_BF16: return p.C ? launch_epi<EPI_BF16>(p, st) : WVN_ERR_ARG;
    case EPI_GELU_BF16: return p.C ? launch_epi<EPI_GELU_BF16>(p, st) : WVN_ERR_ARG;
    case EPI_F32: return p.C ? launch_epi<EPI_F32>(p, st) : WVN_ERR_ARG;
    case EPI_RESID_F32: return p.C ? launch_epi<EPI_RESID_F32>(p, st) : WVN_ERR_ARG;
    case EPI_QKV:
      if ((p.N % 3) != 0 || ((p.N / 3) % BN) != 0 || !p.q || !p.k || !p.vt || (p.ntok_s % 16) || (p.M % 16) || (p.npad % 16))
        return WVN_ERR_ARG;
      return launch_epi<EPI_QKV>(p, st);
    default: return WVN_ERR_ARG;
  }
}
