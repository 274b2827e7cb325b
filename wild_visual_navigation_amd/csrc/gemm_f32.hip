// fp32 GEMM (exact mode + traversability-MLP training math).  C = epilogue(op(A) * op(B)).
//
// Used where results must match the fp32 reference to ~1e-6 relative: the "exact" parity mode of
// the backbone and every GEMM of the online MLP (forward, dX, dW) -- < 1 % of the path's FLOPs.
// 64x64x16 tile, 256 threads, 4x4 register micro-tile per thread, v_fma_f32; k-major LDS tiles so
// both operand reads are conflict-free ds_read_b128.  Supports both storage orders of either
// operand, batching (blockIdx.z) and deterministic split-K (partial slabs + ordered reduce).
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

template <int EPI>
__device__ inline void f32_store(const GemmF32Params& p, float* C, int m, int n, float v) {
  if (m >= p.M || n >= p.N) return;
  if (p.bias) v += p.bias[n];
  if constexpr (EPI == F32_EPI_NONE) {
    C[(size_t)m * p.ldc + n] = v;
  } else if constexpr (EPI == F32_EPI_RELU) {
    C[(size_t)m * p.ldc + n] = fmaxf(v, 0.f);
  } else if constexpr (EPI == F32_EPI_GELU) {
    C[(size_t)m * p.ldc + n] = gelu_exact(v);
  } else if constexpr (EPI == F32_EPI_RESID) {
    float* c = C + (size_t)m * p.ldc + n;
    if (p.ls) v *= p.ls[n];  // LayerScale (DINOv2)
    *c = *c + v;
  } else if constexpr (EPI == F32_EPI_SIGMOID0) {
    C[(size_t)m * p.ldc + n] = (n == 0) ? sigmoid_f(v) : v;
  } else if constexpr (EPI == F32_EPI_RELUMASK) {
    C[(size_t)m * p.ldc + n] = (p.mask[(size_t)m * p.ldmask + n] > 0.f) ? v : 0.f;
  } else if constexpr (EPI == F32_EPI_PATCH) {
    int b = m / p.npatch, pp = m - b * p.npatch;
    C[((size_t)b * p.ntok_s + 1 + pp) * p.ldc + n] = v + p.pos[(size_t)(1 + pp) * p.ldc + n];
  } else if constexpr (EPI == F32_EPI_QKV) {
    int D = p.N / 3;
    int which = n / D, c = n - which * D, head = c >> 6, d = c & 63;
    int b = m / p.ntok_s, t = m - b * p.ntok_s;
    size_t o = (((size_t)b * p.heads + head) * p.npad + t) * 64 + d;
    (which == 0 ? p.q : which == 1 ? p.k : p.v)[o] = v;
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Params p) {
  __shared__ float As[TK][TM + 4];
  __shared__ float Bs[TK][TN + 4];
  const int tid = threadIdx.x;
  const int z = blockIdx.z;
  const int bz = z / p.splitk, sk = z - bz * p.splitk;
  const float* A = p.A + (size_t)bz * p.strideA;
  const float* B = p.B + (size_t)bz * p.strideB;
  float* C = p.C + (size_t)bz * p.strideC + (p.splitk > 1 ? (size_t)sk * p.M * p.ldc : 0);
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  int kchunk = ((p.K + p.splitk - 1) / p.splitk + TK - 1) / TK * TK;
  const int kbeg = sk * kchunk, kend = min(p.K, kbeg + kchunk);

  const int ty = tid >> 4, tx = tid & 15;  // micro-tile rows ty*4.., cols tx*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    // ---- stage A tile (TM x TK) into As[k][m] ----
    if (p.transA == 0) {  // A[m][k], k contiguous: thread -> (row = tid/4, 4 k's)
      int r = tid >> 2, kk = (tid & 3) * 4;
      int gm = m0 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int gk = k0 + kk + e;
        As[kk + e][r] = (gm < p.M && gk < kend) ? A[(size_t)gm * p.lda + gk] : 0.f;
      }
    } else {  // A stored [k][m], m contiguous: thread -> (k = tid/16, 4 m's)
      int kk = tid >> 4, r = (tid & 15) * 4;
      int gk = k0 + kk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int gm = m0 + r + e;
        As[kk][r + e] = (gm < p.M && gk < kend) ? A[(size_t)gk * p.lda + gm] : 0.f;
      }
    }
    // ---- stage B tile (TK x TN) into Bs[k][n] ----
    if (p.transB == 1) {  // B stored [n][k]
      int r = tid >> 2, kk = (tid & 3) * 4;
      int gn = n0 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int gk = k0 + kk + e;
        Bs[kk + e][r] = (gn < p.N && gk < kend) ? B[(size_t)gn * p.ldb + gk] : 0.f;
      }
    } else {  // B[k][n]
      int kk = tid >> 4, r = (tid & 15) * 4;
      int gk = k0 + kk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int gn = n0 + r + e;
        Bs[kk][r + e] = (gn < p.N && gk < kend) ? B[(size_t)gk * p.ldb + gn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      const f32x4_t a = *(const f32x4_t*)&As[kk][ty * 4];
      const f32x4_t b = *(const f32x4_t*)&Bs[kk][tx * 4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) f32_store<EPI>(p, C, m0 + ty * 4 + i, n0 + tx * 4 + j, acc[i][j]);
}

template <int EPI>
int launch(const GemmF32Params& p, hipStream_t st) {
  dim3 grid(ceil_div(p.N, TN), ceil_div(p.M, TM), p.batch * p.splitk);
  hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

__global__ void splitk_reduce_kernel(const float* part, int splitk, size_t n, const float* bias, int ncols,
                                     float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = part[i];
  for (int k = 1; k < splitk; ++k) s += part[(size_t)k * n + i];  // fixed order -> deterministic
  if (bias) s += bias[i % ncols];
  out[i] = s;
}

}  // namespace

int wvn_gemm_f32_launch(const GemmF32Params& pin, int epi, hipStream_t st) {
  GemmF32Params p = pin;
  if (!p.A || !p.B || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0) return WVN_ERR_ARG;
  if (p.batch <= 0) p.batch = 1;
  if (p.splitk <= 0) p.splitk = 1;
  if (p.splitk > 1 && (epi != F32_EPI_NONE || p.bias)) return WVN_ERR_ARG;
  switch (epi) {
    case F32_EPI_NONE: return launch<F32_EPI_NONE>(p, st);
    case F32_EPI_RELU: return launch<F32_EPI_RELU>(p, st);
    case F32_EPI_GELU: return launch<F32_EPI_GELU>(p, st);
    case F32_EPI_RESID: return launch<F32_EPI_RESID>(p, st);
    case F32_EPI_SIGMOID0: return launch<F32_EPI_SIGMOID0>(p, st);
    case F32_EPI_RELUMASK: return p.mask ? launch<F32_EPI_RELUMASK>(p, st) : WVN_ERR_ARG;
    case F32_EPI_PATCH: return launch<F32_EPI_PATCH>(p, st);
    case F32_EPI_QKV: return launch<F32_EPI_QKV>(p, st);
    default: return WVN_ERR_ARG;
  }
}

int wvn_splitk_reduce_launch(const float* part, int splitk, size_t n, const float* bias, int ncols, float* out,
                             hipStream_t st) {
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, splitk, n, bias,
                     ncols, out);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
