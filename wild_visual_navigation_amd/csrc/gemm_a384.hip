// A-stationary bf16 MFMA GEMM for the K = 384 linears of the ViT (QKV, attention projection, fc1, STEGO
// hidden layer) on gfx950:  C = epilogue(A[M,384] * W[N,384]^T + bias).
//
// Why a second GEMM kernel: with K = 384 a classic output-tiled GEMM has six K-tiles per tile, so its
// prologue (pipeline fill) and epilogue (bias / GELU / layout change / stores) are as long as its MFMA
// loop.  Here the roles are turned around:
//
//   * a workgroup (8 waves) owns 256 rows of A for the WHOLE N range.  Each wave keeps its 32 rows x 384
//     of A in registers (96 VGPRs, loaded once, already in MFMA operand layout) -- A is never staged
//     through LDS and never re-read;
//   * W streams through a 4-deep LDS ring in slices of 64 (n) x 128 (k) bf16 = 16 KB, written by
//     direct-to-LDS loads (global_load_lds_dwordx4: no VGPR round trip, no ds_write), read by all 8 waves.
//     The ring never drains: one s_barrier per slice, counted s_waitcnt vmcnt so two slices stay in flight
//     across every barrier.  LDS rows are 256 B, 16-byte chunks XOR-swizzled by (row & 15) on the SOURCE
//     address (the DMA writes lane-linear) and on the read -> ds_read_b128 is bank-conflict free;
//   * the output is produced 64 columns at a time (3 slices = 48 MFMAs per wave per column tile).  The
//     epilogue of column tile j-1 (bias, GELU, bf16 pack, wave-private LDS transpose, 16-byte coalesced
//     row stores) is software-pipelined into the MFMA loop of tile j, one third per slice, and the two
//     waves that share a SIMD run the two halves in opposite order ("ping-pong": waves 0-3 MFMA then
//     epilogue, waves 4-7 epilogue then MFMA), so within every slice period one wave feeds the matrix pipe
//     while its partner does the VALU / LDS / store work of the epilogue.
//
// MFMA: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  "TR" orientation mfma(Wfrag, Afrag) leaves lane =
// output row m, registers = 4 consecutive columns n (row-major outputs); the V third of the QKV projection
// uses mfma(Afrag, Wfrag) (lane = n, registers = 4 consecutive tokens) so V^T rows are token-contiguous.
#include <type_traits>

#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int KD = 384;                   // the K this kernel is specialised for
constexpr int BNT = 64;                   // output columns per column tile
constexpr int SLK = 128;                  // k per ring slice
constexpr int NSL = KD / SLK;             // slices per column tile (3)
constexpr int NS = 4;                     // ring depth
constexpr int SLICE_BYTES = BNT * SLK * 2;  // 16 KB
constexpr int RING_BYTES = NS * SLICE_BYTES;
constexpr int STG_BYTES = 8704;           // wave-private epilogue staging (32 x 64 fp32 rows of 272 B)
constexpr int STG_OFF = RING_BYTES;
constexpr int BIAS_OFF = STG_OFF + 8 * STG_BYTES;
constexpr int BM = 256;

enum { A_BF16 = 0, A_GELU = 1, A_RELU = 2, A_RESID = 3, A_QKV = 4 };

struct A384Params {
  const bf16_t* A; int lda;
  const bf16_t* W;     // [N][384]
  const float* bias;   // [N] or nullptr
  void* C; int ldc;    // bf16 or fp32 (A_RESID: in/out)
  int M, N;
  bf16_t* q; bf16_t* k; bf16_t* vt; int heads; int npad; int ntok_s;
};

// tanh-form GELU evaluated as x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): |err| < 5e-4 absolute
// against the exact erf GELU, below the bf16 rounding of the value it is stored as (the fp32 "exact" path
// keeps erff).  5 VALU + 2 transcendental ops per element instead of 14 + 2.
__device__ inline float gelu_fast(float x) {
  const float x2 = x * x;
  const float t = fmaf(x2, -0.1029432f, -2.3022082f);  // -2*sqrt(2/pi)*log2(e) * (1 + 0.044715 x^2)
  const float e = __builtin_amdgcn_exp2f(t * x);       // exp(-2u)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int EPI>
__device__ inline float act(float v) {
  if constexpr (EPI == A_GELU) return gelu_fast(v);
  if constexpr (EPI == A_RELU) return fmaxf(v, 0.f);
  return v;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_a384_kernel(A384Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0w = blockIdx.x * BM + wave * 32;  // first row of this wave
  const int NT = p.N / BNT;
  const int total = NT * NSL;
  const bool epi_first = wave >= 4;
  unsigned char* stg = smem + STG_OFF + wave * STG_BYTES;
  const float* bias_l = (const float*)(smem + BIAS_OFF);

  // ---- W ring producer: 16 wave-instructions of 1 KB (4 rows x 256 B) per slice, 2 per wave ----------
  const bf16_t* wsrc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int inst = wave * 2 + u;
    const int row = inst * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (row & 15);
    wsrc[u] = p.W + (size_t)row * KD + chunk * 8;
  }
  auto issue = [&](int i) {  // slice i = (column tile i / 3, k slice i % 3)
    const int j = i / NSL, ks = i - j * NSL;
    const size_t off = (size_t)j * BNT * KD + ks * SLK;
    unsigned char* dst = smem + (i % NS) * SLICE_BYTES + wave * 2048;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[u] + off),
                                       (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < total) issue(i);

  // ---- bias -> LDS, A rows -> registers (MFMA operand layout: row l31, k = 16 s + 8 hi .. + 7) --------
  for (int i = tid; i < p.N; i += 512) ((float*)(smem + BIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
  bf16x8_t xf[KD / 16];
  {
    const bf16_t* ap = p.A + (size_t)min(m0w + l31, p.M - 1) * p.lda + hi * 8;
#pragma unroll
    for (int s = 0; s < KD / 16; ++s) xf[s] = *(const bf16x8_t*)(ap + s * 16);
  }

  f32x16_t acc[2], prev[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; prev[t][r] = 0.f; }

  const int xorc = l31 & 15;
  const unsigned rd_base = l31 * 256;

  // ---- 16 MFMAs on one ring slice: 4 fragments in flight, one ds_read per MFMA --------------------------
  auto mfma_block = [&](int slot, int ks, auto tr_tag) {
    constexpr bool TR = decltype(tr_tag)::value;
    const unsigned char* base = smem + slot * SLICE_BYTES + rd_base;
    auto rd = [&](int s, int t) {
      return *(const bf16x8_t*)(base + t * 8192 + (((2 * s + hi) ^ xorc) << 4));
    };
    bf16x8_t wf[4];  // two k-steps of fragments in flight
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = rd(i >> 1, i & 1);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (TR)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(s & 1) * 2 + t], xf[ks * 8 + s], acc[t], 0, 0, 0);
        else
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks * 8 + s], wf[(s & 1) * 2 + t], acc[t], 0, 0, 0);
        if (s + 2 < 8) wf[(s & 1) * 2 + t] = rd(s + 2, t);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  };

  // ---- one third of the epilogue of column tile jp (accumulators in prev[]) ------------------------------
  // TR tiles : image [32 rows m][64 cols n]  (bf16 rows of 144 B, fp32 rows of 272 B)
  // !TR tiles: image [64 rows n][32 cols m]  (bf16 rows of 80 B)            -- V^T of the QKV projection
  auto epi_part = [&](int part, int jp, auto tr_tag) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = jp * BNT;
    if (part < 2) {
      const int t = part;
      if constexpr (TR) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 32 * t + 8 * g + 4 * hi;
          const f32x4_t b4 = *(const f32x4_t*)(bias_l + n0 + c);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act<EPI>(prev[t][4 * g + e] + b4[e]);
          if constexpr (EPI == A_RESID) {
            f32x4_t o = {v[0], v[1], v[2], v[3]};
            *(f32x4_t*)(stg + l31 * 272 + c * 4) = o;
          } else {
            u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *(u32x2_t*)(stg + l31 * 144 + c * 2) = o;
          }
        }
      } else {
        const float b = bias_l[n0 + 32 * t + l31];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int mloc = 8 * g + 4 * hi;
          u32x2_t o = {pack_bf16x2(prev[t][4 * g + 0] + b, prev[t][4 * g + 1] + b),
                       pack_bf16x2(prev[t][4 * g + 2] + b, prev[t][4 * g + 3] + b)};
          *(u32x2_t*)(stg + (32 * t + l31) * 80 + mloc * 2) = o;
        }
      }
      return;
    }
    // part 2: wave-private image -> global, 16 bytes per lane, whole rows per 8 (bf16) / 16 (fp32) lanes
    if constexpr (EPI == A_RESID) {
      float* C = (float*)p.C;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int ch = it * 64 + lane, row = ch >> 4, c4 = (ch & 15) * 4;
        const int m = m0w + row;
        f32x4_t v = *(const f32x4_t*)(stg + row * 272 + c4 * 4);
        if (m < p.M) {
          float* dst = C + (size_t)m * p.ldc + n0 + c4;
          v += *(const f32x4_t*)dst;
          *(f32x4_t*)dst = v;
        }
        if (it & 1) __builtin_amdgcn_sched_barrier(0);  // two rows of chunks in flight at a time (VGPR budget)
      }
    } else if constexpr (EPI == A_QKV) {
      const int D = p.heads * 64;
      if constexpr (TR) {  // q / k: one column tile = one head; dst[(b*h + head)*npad + t][0..63]
        const int which = n0 / D, head = (n0 - which * D) >> 6;
        bf16_t* dstb = which == 0 ? p.q : p.k;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int ch = it * 64 + lane, row = ch >> 3, c8 = (ch & 7) * 8;
          const int m = m0w + row;
          const u32x4_t val = *(const u32x4_t*)(stg + row * 144 + c8 * 2);
          if (m < p.M) {
            const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
            *(u32x4_t*)(dstb + (((size_t)b * p.heads + head) * p.npad + tk) * 64 + c8) = val;
          }
          if (it & 1) __builtin_amdgcn_sched_barrier(0);
        }
      } else {  // v: vt[(b*h + head)*64 + d][t], 8 tokens (16 B) per lane
        const int head = (n0 - 2 * D) >> 6;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int ch = it * 64 + lane, d = ch >> 2, m8 = (ch & 3) * 8;
          const int m = m0w + m8;
          const u32x4_t val = *(const u32x4_t*)(stg + d * 80 + m8 * 2);
          if (m < p.M) {  // M % 8 == 0 and ntok_s % 8 == 0: an 8-token chunk is entirely in or out, one frame
            const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
            *(u32x4_t*)(p.vt + (((size_t)b * p.heads + head) * 64 + d) * p.npad + tk) = val;
          }
          if (it & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      bf16_t* C = (bf16_t*)p.C;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ch = it * 64 + lane, row = ch >> 3, c8 = (ch & 7) * 8;
        const int m = m0w + row;
        const u32x4_t val = *(const u32x4_t*)(stg + row * 144 + c8 * 2);
        if (m < p.M) *(u32x4_t*)(C + (size_t)m * p.ldc + n0 + c8) = val;
      }
    }
  };

  // ---- one slice period -----------------------------------------------------------------------------------
  // VM queue at the boundary of slice i (oldest first): DMA(i), [stores of a tile epilogue], DMA(i+1),
  // DMA(i+2) with the stores somewhere behind DMA(i) (they are issued in the ks == 2 period, after that
  // period's DMA).  `allow` = operations that may stay outstanding = everything younger than DMA(i).
  constexpr int ST = EPI == A_RESID ? 16 : 4;  // VM operations of one epilogue part 2, per lane
  auto period = [&](int i, int ks, int j, bool stores_in_window, auto mtr, auto etr, bool do_epi) {
    if (i + 2 < total) {
      if (ks != 2 && stores_in_window) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + ST) : "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (i + NS - 1 < total) issue(i + NS - 1);
    __builtin_amdgcn_sched_barrier(0);
    if (epi_first) {
      if (do_epi) epi_part(ks, j - 1, etr);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(i % NS, ks, mtr);
    } else {
      mfma_block(i % NS, ks, mtr);
      __builtin_amdgcn_sched_barrier(0);
      if (do_epi) epi_part(ks, j - 1, etr);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tile = [&](int j, auto mtr, auto etr, bool do_epi, bool stores_in_window) {
#pragma unroll
    for (int ks = 0; ks < NSL; ++ks) period(j * NSL + ks, ks, j, stores_in_window, mtr, etr, do_epi);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      prev[t] = acc[t];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
  };

  using T = std::true_type;
  using F = std::false_type;
  __syncthreads();  // bias table visible (nothing DMA'd is read before the first period's wait + barrier)
  if constexpr (EPI == A_QKV) {
    const int nqk = 2 * (p.heads * 64) / BNT;  // q and k column tiles (TR); the rest are V tiles
    tile(0, T{}, T{}, false, false);
    tile(1, T{}, T{}, true, false);
    for (int j = 2; j < nqk; ++j) tile(j, T{}, T{}, true, true);
    tile(nqk, F{}, T{}, true, true);
    for (int j = nqk + 1; j < NT; ++j) tile(j, F{}, F{}, true, true);
#pragma unroll
    for (int part = 0; part < 3; ++part) epi_part(part, NT - 1, F{});
  } else {
    tile(0, T{}, T{}, false, false);
    if (NT > 1) tile(1, T{}, T{}, true, false);
    for (int j = 2; j < NT; ++j) tile(j, T{}, T{}, true, true);
#pragma unroll
    for (int part = 0; part < 3; ++part) epi_part(part, NT - 1, T{});
  }
}

constexpr int A384_LDS_MAX = 160 * 1024;

template <int EPI>
int launch(const A384Params& p, hipStream_t st) {
  const int lds = BIAS_OFF + p.N * 4;
  if (lds > A384_LDS_MAX) return WVN_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_a384_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       A384_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_a384_kernel<EPI>), dim3(ceil_div(p.M, BM)), dim3(512), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

// Eligibility: K == 384, N % 64 == 0, 16-byte aligned operands; returns WVN_ERR_ARG otherwise so the
// caller can use the generic tiled kernel.  epi uses the GemmEpilogue codes of wvn_internal.h.
int wvn_gemm_a384_launch(const GemmBf16Params& g, int epi, hipStream_t st) {
  if (g.K != KD || g.ldw != KD || (g.N % BNT) != 0 || g.M <= 0 || !g.A || !g.W || (g.lda % 8) != 0) return WVN_ERR_ARG;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return WVN_ERR_ARG;
  A384Params p{};
  p.A = g.A; p.lda = g.lda; p.W = g.W; p.bias = g.bias; p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N;
  p.q = g.q; p.k = g.k; p.vt = g.vt; p.heads = g.heads; p.npad = g.npad; p.ntok_s = g.ntok_s;
  switch (epi) {
    case EPI_BF16:
    case EPI_GELU_BF16:
    case EPI_RELU_BF16:
      if (!g.C || (g.ldc % 8) != 0 || ((uintptr_t)g.C & 15)) return WVN_ERR_ARG;
      return epi == EPI_BF16 ? launch<A_BF16>(p, st) : epi == EPI_GELU_BF16 ? launch<A_GELU>(p, st) : launch<A_RELU>(p, st);
    case EPI_RESID_F32:
    case EPI_ACCUM_F32:
      if (!g.C || (g.ldc % 4) != 0 || ((uintptr_t)g.C & 15)) return WVN_ERR_ARG;
      return launch<A_RESID>(p, st);
    case EPI_QKV:
      if (g.N != 3 * g.heads * 64 || !g.q || !g.k || !g.vt || (g.ntok_s % 8) || (g.M % 8) || (g.npad % 8)) return WVN_ERR_ARG;
      return launch<A_QKV>(p, st);
    default: return WVN_ERR_ARG;
  }
}
