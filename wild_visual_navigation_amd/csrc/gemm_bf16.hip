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
// Fragment maps (v_mfma_f32_32x32x16_bf16): A: lane l holds row l&31, k-slots (l>>5)*8+j;
// B: lane l holds col l&31, same k-slots; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// SWAP mode issues mfma(Wfrag, Afrag) so the accumulator holds C^T (lane <-> m): used for the V third
// of the QKV projection, which the attention kernel wants transposed ([b,h,d,token], token-contiguous).
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_STRIDE = BK + 8;                       // bf16 elements per LDS row (144 B)
constexpr int STAGE_ELEMS = (BM + BN) * LDS_STRIDE;      // per stage
constexpr int GEMM_LDS_BYTES = 2 * STAGE_ELEMS * 2;      // 73,728 B

template <int EPI>
__device__ inline void epilogue_store(const GemmBf16Params& p, int m, int n, float v) {
  if (m >= p.M || n >= p.N) return;
  if (p.bias) v += p.bias[n];
  if constexpr (EPI == EPI_BF16) {
    ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(v);
  } else if constexpr (EPI == EPI_GELU_BF16) {
    ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(gelu_exact(v));
  } else if constexpr (EPI == EPI_RELU_BF16) {
    ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(fmaxf(v, 0.f));
  } else if constexpr (EPI == EPI_F32) {
    ((float*)p.C)[(size_t)m * p.ldc + n] = v;
  } else if constexpr (EPI == EPI_RESID_F32) {
    float* c = (float*)p.C + (size_t)m * p.ldc + n;
    *c = *c + v;  // in-place residual update: every element is owned by exactly one lane
  } else if constexpr (EPI == EPI_ACCUM_F32) {
    float* c = (float*)p.C + (size_t)m * p.ldc + n;
    *c = *c + v;
  } else if constexpr (EPI == EPI_PATCH) {
    // m indexes patches (b, p); token row = b*ntok + 1 + p ; add the position table row 1+p
    int b = m / p.npatch, pp = m - b * p.npatch;
    ((float*)p.C)[((size_t)b * p.ntok + 1 + pp) * p.ldc + n] = v + p.pos[(size_t)(1 + pp) * p.ldc + n];
  } else if constexpr (EPI == EPI_QKV) {
    // n in [0, 3*D): which = n / D ; head = (n % D) / 64 ; d = n % 64.   m = b*ntok + t
    int D = p.N / 3;
    int which = n / D, c = n - which * D, head = c >> 6, d = c & 63;
    int b = m / p.ntok, t = m - b * p.ntok;
    size_t bh = (size_t)b * p.heads + head;
    bf16_t o = f32_to_bf16(v);
    if (which == 0) p.q[(bh * p.npad + t) * 64 + d] = o;
    else if (which == 1) p.k[(bh * p.npad + t) * 64 + d] = o;
    else p.vt[(bh * 64 + d) * p.npad + t] = o;
  }
}

template <int EPI, bool SWAP>
__device__ inline void gemm_tile(const GemmBf16Params& p, int tm, int tn, bf16_t* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging assignment: 4 A chunks + 4 W chunks of 16 B per thread per K-tile
  // chunk c = tid + 256*i : row = c >> 3 (0..127), kc = c & 7 (8 bf16 each)
  u32x4_t ra[4], rb[4];
  const int srow = tid >> 3, skc = tid & 7;

  auto load_regs = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = srow + 32 * i;
      int gm = m0 + row, gn = n0 + row;
      u32x4_t z = {0u, 0u, 0u, 0u};
      ra[i] = (gm < p.M) ? *(const u32x4_t*)(p.A + (size_t)gm * p.lda + kt * BK + skc * 8) : z;
      rb[i] = (gn < p.N) ? *(const u32x4_t*)(p.W + (size_t)gn * p.ldw + kt * BK + skc * 8) : z;
    }
  };
  auto store_regs = [&](int stage) {
    bf16_t* As = lds + stage * STAGE_ELEMS;
    bf16_t* Bs = As + BM * LDS_STRIDE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = srow + 32 * i;
      *(u32x4_t*)(As + row * LDS_STRIDE + skc * 8) = ra[i];
      *(u32x4_t*)(Bs + row * LDS_STRIDE + skc * 8) = rb[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_regs(0);
  store_regs(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) load_regs(kt + 1);
    const bf16_t* As = lds + (kt & 1) * STAGE_ELEMS;
    const bf16_t* Bs = As + BM * LDS_STRIDE;
    const bf16_t* a_base = As + (wm * 64 + l31) * LDS_STRIDE + hi * 8;
    const bf16_t* b_base = Bs + (wn * 64 + l31) * LDS_STRIDE + hi * 8;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8_t*)(a_base + i * 32 * LDS_STRIDE + s * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8_t*)(b_base + j * 32 * LDS_STRIDE + s * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (SWAP)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    if (more) store_regs((kt + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        int m, n;
        if constexpr (SWAP) {
          n = n0 + wn * 64 + j * 32 + rr;
          m = m0 + wm * 64 + i * 32 + l31;
        } else {
          m = m0 + wm * 64 + i * 32 + rr;
          n = n0 + wn * 64 + j * 32 + l31;
        }
        epilogue_store<EPI>(p, m, n, acc[i][j][r]);
      }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmBf16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* lds = (bf16_t*)smem;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if constexpr (EPI == EPI_QKV) {
    // block-uniform: the V third (n >= 2D) is produced transposed
    if (tn * BN >= 2 * (p.N / 3)) {
      gemm_tile<EPI, true>(p, tm, tn, lds);
      return;
    }
  }
  gemm_tile<EPI, false>(p, tm, tn, lds);
}

template <int EPI>
int launch(const GemmBf16Params& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       GEMM_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles), dim3(256), GEMM_LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

int wvn_gemm_bf16_launch(const GemmBf16Params& p, int epi, hipStream_t st) {
  if (!p.A || !p.W || p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % BK) != 0 || (p.lda % 8) != 0 || (p.ldw % 8) != 0)
    return WVN_ERR_ARG;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return WVN_ERR_ARG;
  switch (epi) {
    case EPI_BF16: return launch<EPI_BF16>(p, st);
    case EPI_GELU_BF16: return launch<EPI_GELU_BF16>(p, st);
    case EPI_RELU_BF16: return launch<EPI_RELU_BF16>(p, st);
    case EPI_F32: return launch<EPI_F32>(p, st);
    case EPI_RESID_F32: return launch<EPI_RESID_F32>(p, st);
    case EPI_ACCUM_F32: return launch<EPI_ACCUM_F32>(p, st);
    case EPI_PATCH: return launch<EPI_PATCH>(p, st);
    case EPI_QKV:
      if ((p.N % 3) != 0 || ((p.N / 3) % BN) != 0 || !p.q || !p.k || !p.vt) return WVN_ERR_ARG;
      return launch<EPI_QKV>(p, st);
    default: return WVN_ERR_ARG;
  }
}
