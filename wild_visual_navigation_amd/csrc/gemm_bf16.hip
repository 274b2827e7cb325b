// bf16 MFMA GEMM for the ViT / STEGO-head linears on gfx950:  C = epilogue(A[M,K] * W[N,K]^T)
//
// A and W are both K-contiguous (W is exactly the torch.nn.Linear [out,in] layout), so the A and B
// MFMA fragments are 16-byte LDS reads.  fp32 accumulation in v_mfma_f32_32x32x16_bf16.
//
//   tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles
//   LDS: 2 stages x (A 128x(64+8) + W 128x(64+8)) bf16 = 73,728 B  -> 2 workgroups / CU
//   row stride 144 B: the 16-lane groups of ds_read_b128 land on 16 distinct 16-B slots (no
//   bank conflicts), and rows stay 16-B aligned for ds_write_b128 staging.
//   global->register prefetch of tile k+1 is issued before the MFMAs of tile k (one barrier per
//   K-tile); tiles are walked N-fastest with an XCD-aware block remap so the A row-panel and the
//   (small, shared) W stay in the XCD's L2.
//
// Epilogue (the ViT GEMMs have K = 384: six K-tiles, so the epilogue IS the kernel): the
// accumulators are produced TRANSPOSED (mfma(Wfrag, Afrag): lane <-> output row m, registers <-> 4
// consecutive output columns n), bias / GELU / ReLU are applied in registers, the tile is staged
// through the (now idle) operand LDS as a row-major image and leaves the CU as 16-byte, fully
// coalesced stores (256 contiguous bytes per 16 lanes).  The V third of the QKV projection uses the
// un-transposed orientation instead, so its LDS image is [d][token] and V^T ([b,h,d,token], what the
// attention kernel's PV MFMA wants) is written with the same wide stores.
//
// Fragment maps (v_mfma_f32_32x32x16_bf16): A: lane l holds row l&31, k-slots (l>>5)*8+j;
// B: lane l holds col l&31, same k-slots; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_STRIDE = BK + 8;                       // bf16 elements per LDS row (144 B)
constexpr int STAGE_ELEMS = (BM + BN) * LDS_STRIDE;      // per stage
constexpr int GEMM_LDS_BYTES = 2 * STAGE_ELEMS * 2;      // 73,728 B
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
template <int EPI, bool TR, int PD, int NK>
__device__ inline void gemm_tile(const GemmBf16Params& p, int tm, int tn, unsigned char* smem) {
  op16_t* lds = (op16_t*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;

  // Staging: 4 A chunks + 4 W chunks of 16 B per thread per K-tile, held in one of PD register sets.
  // The memory system needs ~2-3k cycles to return a K-tile while its MFMAs take ~0.5k, so PD K-tiles are
  // kept in flight per workgroup (register prefetch depth PD; slot of tile t = t % PD).  LDS stays
  // double-buffered: tile t+1 moves registers -> LDS right after the MFMAs of tile t, and the freed
  // register slot is immediately re-issued for tile t+1+PD.  All slot indices are compile-time (the K
  // loop is unrolled by PD) so nothing spills to scratch.
  // Loads are branch-free: rows past M / N are clamped to the last valid row (their results are never
  // stored; an output element depends only on its own A row and its own W row).  With no control flow
  // around the loads and a fully unrolled K loop (NK > 0) hipcc counts its vmcnt waits exactly, i.e. it
  // waits only for the tile it is about to move to LDS while PD-1 younger tiles stay in flight.
  u32x4_t ra[PD][4], rb[PD][4];
  const int srow = tid >> 3, skc = tid & 7;
  const op16_t* pa[4];
  const op16_t* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pa[i] = p.A + (size_t)min(m0 + srow + 32 * i, p.M - 1) * p.lda + skc * 8;
    pb[i] = p.W + (size_t)min(n0 + srow + 32 * i, p.N - 1) * p.ldw + skc * 8;
  }
  auto load_regs = [&](int kt, u32x4_t (&a)[4], u32x4_t (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *(const u32x4_t*)(pa[i] + kt * BK);
      b[i] = *(const u32x4_t*)(pb[i] + kt * BK);
    }
  };
  auto store_regs = [&](int stage, const u32x4_t (&a)[4], const u32x4_t (&b)[4]) {
    op16_t* As = lds + stage * STAGE_ELEMS;
    op16_t* Bs = As + BM * LDS_STRIDE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = srow + 32 * i;
      *(u32x4_t*)(As + row * LDS_STRIDE + skc * 8) = a[i];
      *(u32x4_t*)(Bs + row * LDS_STRIDE + skc * 8) = b[i];
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
    const op16_t* As = lds + stage * STAGE_ELEMS;
    const op16_t* Bs = As + BM * LDS_STRIDE;
    const op16_t* a_base = As + (wm * 64 + l31) * LDS_STRIDE + hi * 8;
    const op16_t* b_base = Bs + (wn * 64 + l31) * LDS_STRIDE + hi * 8;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      opx8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const opx8_t*)(a_base + i * 32 * LDS_STRIDE + s * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const opx8_t*)(b_base + j * 32 * LDS_STRIDE + s * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (TR)
            acc[i][j] = wvn_mfma_32x32x16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = wvn_mfma_32x32x16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  if constexpr (NK > 0) {
    // straight-line software pipeline: every index and every condition below folds at compile time
#pragma unroll
    for (int u = 0; u < PD; ++u)
      if (u < NK) load_regs(u, ra[u], rb[u]);
    store_regs(0, ra[0], rb[0]);
    if (PD < NK) load_regs(PD, ra[0], rb[0]);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      compute(kt & 1);
      if (kt + 1 < NK) {
        store_regs((kt + 1) & 1, ra[(kt + 1) % PD], rb[(kt + 1) % PD]);  // tile kt+1: issued PD tiles ago
        if (kt + 1 + PD < NK) load_regs(kt + 1 + PD, ra[(kt + 1) % PD], rb[(kt + 1) % PD]);
      }
      __syncthreads();  // also: after the last K-tile every wave is done with the operand LDS
    }
  } else {
    // generic K (any multiple of 64): runtime loop, one tile in flight
    const int nk = p.K / BK;
    load_regs(0, ra[0], rb[0]);
    store_regs(0, ra[0], rb[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 1 < nk);
      if (more) load_regs(kt + 1, ra[0], rb[0]);
      compute(kt & 1);
      if (more) store_regs((kt + 1) & 1, ra[0], rb[0]);
      __syncthreads();
    }
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
          if constexpr (EPI == EPI_RESID_F32 && TR) {
            if (p.ls) v[e] *= (n0 + c + e < p.N) ? p.ls[n0 + c + e] : 0.f;  // LayerScale (DINOv2): x += ls * (acc + bias)
          }
          if constexpr (EPI == EPI_QKV && TR) {
            if (p.q_scale != 0.f && n0 < p.N / 3) v[e] *= p.q_scale;  // q third (tile-uniform): softmax scale folded in
          }
        }
        if constexpr (OB) {
          u32x2_t o = {pack_op2(v[0], v[1]), pack_op2(v[2], v[3])};
          *(u32x2_t*)((op16_t*)smem + lane_dim * CT_BF16_STRIDE + c) = o;
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
      op16_t* dst = which == 0 ? p.q : p.k;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const int b = m / p.ntok_s, t = m - b * p.ntok_s;
        const int cc = cbase + c8, head = cc >> 6, d = cc & 63;
        const u32x4_t val = *(const u32x4_t*)((const op16_t*)smem + row * CT_BF16_STRIDE + c8);
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
        const u32x4_t val = *(const u32x4_t*)((const op16_t*)smem + row * CT_BF16_STRIDE + c8);
        *(u32x4_t*)(p.vt + (((size_t)b * p.heads + head) * 64 + d) * p.npad + t) = val;
      }
    }
  } else if constexpr (OB) {
    op16_t* C = (op16_t*)p.C;
    const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)C & 15) == 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int ch = tid + 256 * it, row = ch >> 4, c8 = (ch & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m >= p.M || n >= p.N) continue;
      const op16_t* src = (const op16_t*)smem + row * CT_BF16_STRIDE + c8;
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

template <int EPI, int PD, int NK>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmBf16Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if constexpr (EPI == EPI_QKV) {
    if (tn * BN >= 2 * (p.N / 3)) {  // block-uniform: the V third is produced as V^T
      gemm_tile<EPI, false, PD, NK>(p, tm, tn, smem);
      return;
    }
  }
  gemm_tile<EPI, true, PD, NK>(p, tm, tn, smem);
}

template <int EPI, int PD, int NK>
int launch_v(const GemmBf16Params& p, hipStream_t st) {
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(GEMM_LDS_BYTES, (const void*)gemm_bf16_kernel<EPI, PD, NK>)) return rc;
  int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, PD, NK>), dim3(tiles), dim3(256), GEMM_LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// register prefetch depth of the unrolled K pipelines: 3 K-tiles in flight per workgroup (measured best of 1 / 2 / 3)
template <int EPI, int NK>
int launch_nk(const GemmBf16Params& p, hipStream_t st) {
  return launch_v<EPI, 3, NK>(p, st);
}

// K of the hot-path GEMMs is known: 384 (qkv / proj / fc1 / STEGO hidden), 1536 (fc2), 192 (patch embed),
// 768 (STEGO code).  Those get the unrolled pipelines; anything else the generic loop.
template <int EPI>
int launch(const GemmBf16Params& p, hipStream_t st) {
  const int nk = p.K / BK;
  if (nk == 6) return launch_nk<EPI, 6>(p, st);
  if constexpr (EPI == EPI_RESID_F32) {
    if (nk == 24) return launch_nk<EPI, 24>(p, st);
  }
  if constexpr (EPI == EPI_PATCH) {
    if (nk == 3) return launch_nk<EPI, 3>(p, st);
  }
  if constexpr (EPI == EPI_F32) {
    if (nk == 12) return launch_nk<EPI, 12>(p, st);
  }
  return launch_v<EPI, 1, 0>(p, st);
}

}  // namespace

int WVN_OPSYM(wvn_gemm_bf16_launch)(const GemmBf16Params& p, int epi, hipStream_t st) {
  if (!p.A || !p.W || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.lda % 8) != 0 || (p.ldw % 8) != 0)
    return WVN_ERR_ARG;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return WVN_ERR_ARG;
  // attention projection + residual at sizes that fill the chip: W resident in LDS, built for HBM throughput (gemm_proj.hip)
  if (epi == EPI_RESID_F32 && p.N == 384 && p.K == 384 && p.ldw == 384 && p.M >= 32768 && !p.dbg) {
    const int rc = WVN_OPSYM(wvn_proj_resid_launch)(p.A, p.lda, p.W, p.bias, p.ls, (float*)p.C, p.ldc, p.M, st);
    if (rc != WVN_ERR_ARG) return rc;
  }
  if (p.K == 384 && !p.ls) {  // A-stationary kernel for the K = 384 linears
    const int rc = WVN_OPSYM(wvn_gemm_a384_launch)(p, epi, st);
    if (rc != WVN_ERR_ARG) return rc;
  }
  if (p.N == 384 && p.K > 384 && !p.ls) {  // row-panel kernel for the fc2 residual update
    int done = 0;
    const int rc = WVN_OPSYM(wvn_gemm_n384_launch)(p, epi, st, &done);
    if (rc != WVN_ERR_ARG) {
      if (rc != WVN_OK || done >= p.M) return rc;
      GemmBf16Params rest = p;  // the rows of a thin last round go through the tiled kernel below
      rest.A = p.A + (size_t)done * p.lda;
      rest.C = (epi == EPI_RESID_F32 || epi == EPI_ACCUM_F32) ? (void*)((float*)p.C + (size_t)done * p.ldc) : p.C;
      rest.M = p.M - done;
      return launch<EPI_RESID_F32>(rest, st);
    }
  }
  switch (epi) {
    case EPI_BF16: return launch<EPI_BF16>(p, st);
    case EPI_GELU_BF16: return launch<EPI_GELU_BF16>(p, st);
    case EPI_RELU_BF16: return launch<EPI_RELU_BF16>(p, st);
    case EPI_F32: return launch<EPI_F32>(p, st);
    case EPI_RESID_F32: return launch<EPI_RESID_F32>(p, st);
    case EPI_ACCUM_F32: return launch<EPI_ACCUM_F32>(p, st);
    case EPI_PATCH:
      if ((p.ldc & 3) || !p.pos || p.N % 4) return WVN_ERR_ARG;
      return launch<EPI_PATCH>(p, st);
    case EPI_QKV:
      if ((p.N % 3) != 0 || ((p.N / 3) % BN) != 0 || !p.q || !p.k || !p.vt || (p.ntok_s % 16) || (p.M % 16) ||
          (p.npad % 16))
        return WVN_ERR_ARG;
      return launch<EPI_QKV>(p, st);
    default: return WVN_ERR_ARG;
  }
}
