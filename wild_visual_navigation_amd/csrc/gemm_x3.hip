// Exact-mode MFMA GEMM for the ViT / STEGO-head linears on gfx950:  C = epilogue(A[M,K] * W[N,K]^T), every operand carried
// as TWO bf16 planes (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and every product formed as
//     hi*hi + hi*lo + lo*hi        (three v_mfma_f32_32x32x16_bf16 per fragment pair, fp32 accumulation);
// the dropped lo*lo term and the plane representation error are both ~2^-17 relative, i.e. the result is fp32-class
// (measured against the fp32 FMA kernels in tests/test_gpu_x3.py) while the work runs on the matrix pipe at one third of
// the bf16 rate instead of on the VALU (157 TFLOP/s peak).  This is the path behind precision "exact": the <= 1e-3 parity
// gate of BASELINE.json's north_star, timed by `bench.py --precision exact`.
//
//   tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles, 24 MFMAs per K-tile
//   LDS: 2 stages x 4 planes (A hi/lo, W hi/lo) x 128 x (32+8) bf16 = 81,920 B  -> 2 workgroups / CU
//   row stride 80 B: the 16-lane groups of a ds_read_b128 fall on 16 distinct 16-byte slots (20 r mod 64 is a
//   permutation of the multiples of 4 over any 16 consecutive rows) -> conflict-free fragment reads.
//   Register prefetch two K-tiles deep, branch-free (the K-tile index is clamped, so hipcc counts vmcnt exactly); the
//   loop is unrolled by two so register slots and LDS stages are compile-time.
//
// Outputs that feed another MFMA (LayerNorm'd activations, q/k/v, the GELU'd hidden layer) leave the epilogue as hi/lo
// planes; the residual stream stays fp32.  GELU is the exact erf form (torch.nn.GELU default, as the oracle), erf to 1.5e-7.
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LSTR = BK + 8;                          // bf16 elements per LDS row (80 B)
constexpr int PLANE = 128 * LSTR;                     // one operand plane of a stage
constexpr int STAGE = 4 * PLANE;                      // A hi | A lo | W hi | W lo
constexpr int X3_LDS_BYTES = 2 * STAGE * 2;           // 81,920 B
constexpr int CT_BF16_STRIDE = 128 + 8;               // output image, bf16 elements per row
constexpr int CT_PLANE = 128 * CT_BF16_STRIDE;        // one output plane image (34,816 B)
constexpr int CT_F32_STRIDE = 128 + 4;
static_assert(2 * CT_PLANE * 2 <= X3_LDS_BYTES, "two bf16 plane images must fit in the operand LDS");
static_assert(128 * CT_F32_STRIDE * 4 <= X3_LDS_BYTES, "fp32 tile image must fit in the operand LDS");

// (a, b) -> packed hi plane word and packed lo plane word
__device__ inline void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
  lo = pack_bf16x2(a - ah, b - bh);
}

// exact erf GELU with erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7: fp32-class, like everything else in this mode)
// instead of libm's erff, whose ~40 VALU per element doubled the fc1 kernel's time
__device__ inline float gelu_as(float x) {
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
  if constexpr (EPI == EPI_GELU_BF16) return gelu_as(v);
  if constexpr (EPI == EPI_RELU_BF16) return fmaxf(v, 0.f);
  return v;
}

template <int EPI>
constexpr bool out_is_planes() {
  return EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || EPI == EPI_QKV;
}

// TR = true : accumulators hold C^T (lane = row m, regs = cols n)  -> LDS image [m][n]
// TR = false: accumulators hold C   (lane = col n, regs = rows m)  -> LDS image [n][m]   (V^T tiles)
template <int EPI, bool TR>
__device__ inline void gemm_x3_tile(const GemmBf16Params& p, int tm, int tn, unsigned char* smem) {
  bf16_t* lds = (bf16_t*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K / BK;  // even (K % 64 == 0)

  // staging: per K-tile and plane 128 rows x 64 B = 512 chunks of 16 B -> 2 chunks per thread; 8 chunks per thread in all
  const int srow = tid >> 2, skc = tid & 3;
  const bf16_t *pah[2], *pal[2], *pbh[2], *pbl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const size_t ra = (size_t)min(m0 + srow + 64 * i, p.M - 1) * p.lda + skc * 8;
    const size_t rb = (size_t)min(n0 + srow + 64 * i, p.N - 1) * p.ldw + skc * 8;
    pah[i] = p.A + ra; pal[i] = p.A_lo + ra;
    pbh[i] = p.W + rb; pbl[i] = p.W_lo + rb;
  }
  u32x4_t r[2][8];
  auto load_regs = [&](int kt, u32x4_t (&x)[8]) {
    const int ko = min(kt, nk - 1) * BK;  // clamped: no control flow around the loads
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      x[i] = *(const u32x4_t*)(pah[i] + ko);
      x[2 + i] = *(const u32x4_t*)(pal[i] + ko);
      x[4 + i] = *(const u32x4_t*)(pbh[i] + ko);
      x[6 + i] = *(const u32x4_t*)(pbl[i] + ko);
    }
  };
  auto store_regs = [&](int stage, const u32x4_t (&x)[8]) {
    bf16_t* base = lds + stage * STAGE + srow * LSTR + skc * 8;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) *(u32x4_t*)(base + pl * PLANE + 64 * i * LSTR) = x[2 * pl + i];
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int stage) {
    const bf16_t* a_base = lds + stage * STAGE + (wm * 64 + l31) * LSTR + hi * 8;
    const bf16_t* b_base = lds + stage * STAGE + 2 * PLANE + (wn * 64 + l31) * LSTR + hi * 8;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *(const bf16x8_t*)(a_base + i * 32 * LSTR + s * 16);
        al[i] = *(const bf16x8_t*)(a_base + PLANE + i * 32 * LSTR + s * 16);
        bh[i] = *(const bf16x8_t*)(b_base + i * 32 * LSTR + s * 16);
        bl[i] = *(const bf16x8_t*)(b_base + PLANE + i * 32 * LSTR + s * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (TR) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  load_regs(0, r[0]);
  load_regs(1, r[1]);
  store_regs(0, r[0]);
  load_regs(2, r[0]);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    compute(0);                 // tile kt
    store_regs(1, r[1]);        // tile kt + 1 (exists: nk is even)
    load_regs(kt + 3, r[1]);
    __syncthreads();
    compute(1);                 // tile kt + 1
    store_regs(0, r[0]);        // tile kt + 2 (a clamped re-load of the last tile on the final trip: never read)
    load_regs(kt + 4, r[0]);
    __syncthreads();            // also: after the last K-tile every wave is done with the operand LDS
  }

  // ---------------- epilogue, part 1: registers -> LDS tile image(s) (bias + activation applied) ----------
  constexpr bool OP = out_is_planes<EPI>();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lane_dim = TR ? (wm * 64 + i * 32 + l31) : (wn * 64 + j * 32 + l31);
      const int reg_base = TR ? (wn * 64 + j * 32) : (wm * 64 + i * 32);
      float bl = 0.f;
      if constexpr (!TR) {
        if (p.bias && n0 + lane_dim < p.N) bl = p.bias[n0 + lane_dim];
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        int c = reg_base + 8 * g4 + 4 * hi;
        if constexpr (EPI == EPI_QKV && !TR)  // V^T: tokens permuted inside aligned groups of 16 (bits 2 <-> 3), see attention_bf16.hip
          c = reg_base + 16 * (g4 >> 1) + 8 * hi + 4 * (g4 & 1);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float b = bl;
          if constexpr (TR) b = (p.bias && n0 + c + e < p.N) ? p.bias[n0 + c + e] : 0.f;
          v[e] = activate<EPI>(acc[i][j][4 * g4 + e] + b);
          if constexpr (EPI == EPI_RESID_F32) {
            if (p.ls) v[e] *= (n0 + c + e < p.N) ? p.ls[n0 + c + e] : 0.f;  // LayerScale (DINOv2): x += ls * (acc + bias)
          }
        }
        if constexpr (EPI == EPI_QKV) {
          if (p.qkv_f16) {   // (uniform) WVN_PREC_MIX: ONE fp16 plane for the fp16 attention kernel; q carries scale * log2(e)
            const float qs = (n0 < p.N / 3 && p.q_scale != 0.f) ? p.q_scale : 1.f;
            if (p.q_lo && n0 < p.N / 3) {   // q as two fp16 planes (attention_bf16.hip, QSPLIT)
              uint32_t h0, l0, h1, l1;
              wvn_split2_f16(v[0] * qs, v[1] * qs, h0, l0);
              wvn_split2_f16(v[2] * qs, v[3] * qs, h1, l1);
              const u32x2_t oh = {h0, h1}, ol = {l0, l1};
              *(u32x2_t*)((bf16_t*)smem + lane_dim * CT_BF16_STRIDE + c) = oh;
              *(u32x2_t*)((bf16_t*)smem + CT_PLANE + lane_dim * CT_BF16_STRIDE + c) = ol;
              continue;
            }
            const u32x2_t oh = {pack_f16x2(v[0] * qs, v[1] * qs), pack_f16x2(v[2] * qs, v[3] * qs)};
            *(u32x2_t*)((bf16_t*)smem + lane_dim * CT_BF16_STRIDE + c) = oh;
            continue;
          }
        }
        if constexpr (OP) {
          uint32_t h0, l0, h1, l1;
          split2(v[0], v[1], h0, l0);
          split2(v[2], v[3], h1, l1);
          const u32x2_t oh = {h0, h1}, ol = {l0, l1};
          *(u32x2_t*)((bf16_t*)smem + lane_dim * CT_BF16_STRIDE + c) = oh;
          *(u32x2_t*)((bf16_t*)smem + CT_PLANE + lane_dim * CT_BF16_STRIDE + c) = ol;
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
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      if (pl == 1 && p.qkv_f16 && !(p.q_lo && which == 0)) break;   // (uniform) single fp16 plane, except a two-plane q
      const bf16_t* img = (const bf16_t*)smem + pl * CT_PLANE;
      if constexpr (TR) {  // q / k : image [m][n]; dst[(b*h + head)*npad + t][d]
        bf16_t* dst = which == 0 ? (pl ? p.q_lo : p.q) : (pl ? p.k_lo : p.k);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
          const int m = m0 + row;
          if (m >= p.M) continue;
          const int b = m / p.ntok_s, t = m - b * p.ntok_s;
          const int cc = cbase + c8, head = cc >> 6, d = cc & 63;
          *(u32x4_t*)(dst + (((size_t)b * p.heads + head) * p.npad + t) * 64 + d) = *(const u32x4_t*)(img + row * CT_BF16_STRIDE + c8);
        }
      } else {  // v : image [n = (head, d)][m]; vt[(b*h + head)*64 + d][t], 8 tokens per store
        bf16_t* dst = pl ? p.vt_lo : p.vt;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
          const int m = m0 + c8;
          if (m >= p.M) continue;  // M % 16 == 0: a chunk (and its permutation group of 16) is entirely in or out
          const int b = m / p.ntok_s, t = m - b * p.ntok_s;
          const int cc = cbase + row, head = cc >> 6, d = cc & 63;
          *(u32x4_t*)(dst + (((size_t)b * p.heads + head) * 64 + d) * p.npad + t) = *(const u32x4_t*)(img + row * CT_BF16_STRIDE + c8);
        }
      }
    }
  } else if constexpr (OP) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      bf16_t* C = pl ? (bf16_t*)p.C_lo : (bf16_t*)p.C;
      const bf16_t* img = (const bf16_t*)smem + pl * CT_PLANE;
      const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)C & 15) == 0);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
        const int m = m0 + row, n = n0 + c8;
        if (m >= p.M || n >= p.N) continue;
        const bf16_t* src = img + row * CT_BF16_STRIDE + c8;
        if (vec_ok && n + 8 <= p.N) {
          *(u32x4_t*)(C + (size_t)m * p.ldc + n) = *(const u32x4_t*)src;
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) C[(size_t)m * p.ldc + n + e] = src[e];
        }
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

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(GemmBf16Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if constexpr (EPI == EPI_QKV) {
    if (tn * BN >= 2 * (p.N / 3)) {  // block-uniform: the V third is produced as V^T
      gemm_x3_tile<EPI, false>(p, tm, tn, smem);
      return;
    }
  }
  gemm_x3_tile<EPI, true>(p, tm, tn, smem);
}

template <int EPI>
int launch(const GemmBf16Params& p, hipStream_t st) {
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(X3_LDS_BYTES, (const void*)gemm_x3_kernel<EPI>)) return rc;
  const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL((gemm_x3_kernel<EPI>), dim3(tiles), dim3(256), X3_LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

int wvn_gemm_x3_launch(const GemmBf16Params& p, int epi, hipStream_t st) {
  if (!p.A || !p.A_lo || !p.W || !p.W_lo || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % 64) != 0 || (p.lda % 8) != 0 ||
      (p.ldw % 8) != 0)
    return WVN_ERR_ARG;
  if ((((uintptr_t)p.A | (uintptr_t)p.A_lo | (uintptr_t)p.W | (uintptr_t)p.W_lo) & 15) != 0) return WVN_ERR_ARG;
  switch (epi) {
    case EPI_BF16: return p.C && p.C_lo ? launch<EPI_BF16>(p, st) : WVN_ERR_ARG;
    case EPI_GELU_BF16: return p.C && p.C_lo ? launch<EPI_GELU_BF16>(p, st) : WVN_ERR_ARG;
    case EPI_RELU_BF16: return p.C && p.C_lo ? launch<EPI_RELU_BF16>(p, st) : WVN_ERR_ARG;
    case EPI_F32: return p.C ? launch<EPI_F32>(p, st) : WVN_ERR_ARG;
    case EPI_RESID_F32: return p.C ? launch<EPI_RESID_F32>(p, st) : WVN_ERR_ARG;
    case EPI_ACCUM_F32: return p.C ? launch<EPI_ACCUM_F32>(p, st) : WVN_ERR_ARG;
    case EPI_PATCH:
      if ((p.ldc & 3) || !p.pos || p.N % 4 || !p.C) return WVN_ERR_ARG;
      return launch<EPI_PATCH>(p, st);
    case EPI_QKV:
      if ((p.N % 3) != 0 || ((p.N / 3) % BN) != 0 || !p.q || !p.k || !p.vt || (!p.qkv_f16 && (!p.q_lo || !p.k_lo || !p.vt_lo)) ||
          (p.ntok_s % 16) || (p.M % 16) || (p.npad % 16))
        return WVN_ERR_ARG;
      return launch<EPI_QKV>(p, st);
    default: return WVN_ERR_ARG;
  }
}
