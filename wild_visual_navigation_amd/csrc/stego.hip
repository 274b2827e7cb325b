// STEGO-style per-image clustering of the patch-resolution code (stego_interface.py:94-100 as used
// by FeatureExtractor with run_clustering=True): deterministic cosine k-means.
//
// Integer outputs (segment-index maps) must be bit-exact against the oracle given the same fp32
// code, so every floating-point reduction here has a FIXED, documented order and uses the
// correctly-rounded non-fused intrinsics (__fmul_rn/__fadd_rn/__fsqrt_rn/__fdiv_rn):
//   * dot products / norms: strictly sequential over the channel index
//   * centroid sums       : strictly sequential over the point index (ascending)
//   * argmax              : first maximum (lowest cluster id wins ties)
// oracle/interfaces.py::kmeans_cosine_labels mirrors this operation for operation.
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int KM_MAXK = 64;

// xn[p][:] = code[p][:] / max(||code[p]||, 1e-12)
__global__ void normalize_rows_kernel(const float* __restrict__ code, int ldc, float* __restrict__ xn, int rows, int C) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= rows) return;
  const float* r = code + (size_t)p * ldc;
  float n2 = 0.f;
  for (int d = 0; d < C; ++d) n2 = __fadd_rn(n2, __fmul_rn(r[d], r[d]));
  float n = fmaxf(__fsqrt_rn(n2), 1e-12f);
  for (int d = 0; d < C; ++d) xn[(size_t)p * C + d] = __fdiv_rn(r[d], n);
}

template <int C>
__device__ inline int assign_point(const float* __restrict__ xp, const float* cent, int K) {
  float x[C];
#pragma unroll
  for (int d = 0; d < C; ++d) x[d] = xp[d];
  int best = 0;
  float bv = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const float* c = cent + k * C;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < C; ++d) acc = __fadd_rn(acc, __fmul_rn(x[d], c[d]));
    if (acc > bv) { bv = acc; best = k; }
  }
  return best;
}

// one workgroup per image
template <int C>
__global__ __launch_bounds__(1024) void kmeans_kernel(const float* __restrict__ xn, int* __restrict__ labels,
                                                      int* __restrict__ nseg, int P, int K, int iters, int relabel) {
  extern __shared__ unsigned char smem_raw[];
  float* cent = (float*)smem_raw;            // [K][C]
  float* sums = cent + K * C;                // [K][C]
  float* nrm = sums + K * C;                 // [K]
  int* cnt = (int*)(nrm + K);                // [K]
  int* lut = cnt + K;                        // [K]
  unsigned char* lab = (unsigned char*)(lut + K);  // [P]
  const int b = blockIdx.x;
  const float* X = xn + (size_t)b * P * C;
  const int tid = threadIdx.x, nth = blockDim.x;

  for (int i = tid; i < K * C; i += nth) {
    int k = i / C, d = i - k * C;
    int p0 = (int)(((long long)(2 * k + 1) * P) / (2 * K));
    cent[i] = X[(size_t)p0 * C + d];
  }
  __syncthreads();

  for (int it = 0; it <= iters; ++it) {
    for (int p = tid; p < P; p += nth) lab[p] = (unsigned char)assign_point<C>(X + (size_t)p * C, cent, K);
    __syncthreads();
    if (it == iters) break;
    // centroid sums: thread (k,d) walks the points in ascending order
    for (int i = tid; i < K * C; i += nth) {
      const int k = i / C, d = i - k * C;
      float s = 0.f;
      int n = 0;
      for (int p = 0; p < P; ++p)
        if (lab[p] == k) { s = __fadd_rn(s, X[(size_t)p * C + d]); ++n; }
      sums[i] = s;
      if (d == 0) cnt[k] = n;
    }
    __syncthreads();
    if (tid < K) {
      float n2 = 0.f;
      for (int d = 0; d < C; ++d) n2 = __fadd_rn(n2, __fmul_rn(sums[tid * C + d], sums[tid * C + d]));
      nrm[tid] = fmaxf(__fsqrt_rn(n2), 1e-12f);
    }
    __syncthreads();
    for (int i = tid; i < K * C; i += nth) {
      const int k = i / C;
      if (cnt[k] > 0) cent[i] = __fdiv_rn(sums[i], nrm[k]);
    }
    __syncthreads();
  }
  // optional compaction of the used ids to 0..K'-1 in ascending order (feature_extractor.py:245-246)
  if (tid < K) cnt[tid] = 0;
  __syncthreads();
  for (int p = tid; p < P; p += nth) cnt[lab[p]] = 1;  // benign race: all writers store 1
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 0; k < K; ++k) { lut[k] = run; run += cnt[k]; }
    nseg[b] = run;
  }
  __syncthreads();
  for (int p = tid; p < P; p += nth) labels[(size_t)b * P + p] = relabel ? lut[lab[p]] : (int)lab[p];
}

}  // namespace

int wvn_normalize_rows_launch(const float* code, int ldc, float* xn, int rows, int C, hipStream_t st) {
  if (!code || !xn) return WVN_ERR_ARG;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, st, code, ldc, xn, rows, C);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_kmeans_launch(const float* xn, int* labels, int* nseg, int B, int P, int C, int K, int iters, int relabel,
                      hipStream_t st) {
  if (!xn || !labels || !nseg || K <= 0 || K > KM_MAXK || P <= 0 || P > 60000) return WVN_ERR_ARG;
  size_t shm = (size_t)(2 * K * C + K) * sizeof(float) + 2 * K * sizeof(int) + (size_t)P;
  shm = align_up(shm, 16);
  if (shm > 64 * 1024) return WVN_ERR_ARG;
  if (C == 90) hipLaunchKernelGGL((kmeans_kernel<90>), dim3(B), dim3(1024), shm, st, xn, labels, nseg, P, K, iters, relabel);
  else if (C == 64) hipLaunchKernelGGL((kmeans_kernel<64>), dim3(B), dim3(1024), shm, st, xn, labels, nseg, P, K, iters, relabel);
  else if (C == 16) hipLaunchKernelGGL((kmeans_kernel<16>), dim3(B), dim3(1024), shm, st, xn, labels, nseg, P, K, iters, relabel);
  else return WVN_ERR_ARG;
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
