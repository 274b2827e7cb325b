// STEGO-style per-image clustering of the patch-resolution code (stego_interface.py:94-100 as used
// by FeatureExtractor with run_clustering=True): deterministic cosine k-means.
//
// Integer outputs (segment-index maps) must be bit-exact against the oracle given the same fp32
// code, so every floating-point reduction here has a FIXED, documented order and uses the
// correctly-rounded non-fused intrinsics (__fmul_rn/__fadd_rn/__fsqrt_rn/__fdiv_rn; this file is
// compiled with -ffp-contract=off):
//   * dot products / norms : strictly sequential over the channel index
//   * centroid sums        : points are cut into chunks of KM_CHUNK (64) consecutive indices; inside a
//                            chunk the members of a cluster are added in ascending point order starting
//                            from 0, and the chunk partials are then added in ascending chunk order
//   * argmax               : first maximum (lowest cluster id wins ties)
// oracle/interfaces.py::kmeans_cosine_labels mirrors this operation for operation.
//
// The whole GPU works on every Lloyd iteration (3 small launches per iteration, all images at once):
//   assign  : one lane per point, centroids of the image in LDS
//   partial : one workgroup per (chunk, image), lane = channel: LDS table part[K][C] indexed by label
//             (each lane owns its column, so no atomics and a fixed order)
//   update  : one workgroup per image: ordered sum over chunks, sequential norm, divide
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int KM_MAXK = 64;
constexpr int KM_CHUNK = 64;

// Rows of [rows][C] (C <= 128) are staged through LDS so that global traffic is coalesced (a row is 360 B at
// C = 90; one thread walking its own row touches 64 cache lines per load instruction) while each thread still
// reduces ITS row sequentially in index order -- the arithmetic, and therefore every bit of the result, is unchanged.
constexpr int ROWS_PER_BLOCK = 128;
// LDS tile = plain copy of the rows ([r][C], pitch C): for C = 90 thread r reading column d hits bank (26 r + d) % 64,
// distinct for the 32 lanes of a half-wave; the odd-pitch variant is used when rows are not contiguous in memory.
__host__ __device__ inline int row_pitch(int C, bool contiguous) { return contiguous ? C : (C | 1); }

__device__ inline void stage_rows_in(const float* __restrict__ src, int ld, int row0, int rows, int C, float* tile) {
  const int nrow = min(ROWS_PER_BLOCK, rows - row0);
  if (ld == C && ((C * ROWS_PER_BLOCK) & 3) == 0 && (((uintptr_t)(src + (size_t)row0 * C)) & 15) == 0) {
    const int n4 = nrow * C / 4;  // contiguous block: 16-byte coalesced copy (tail elements below)
    const f32x4_t* s4 = (const f32x4_t*)(src + (size_t)row0 * C);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) ((f32x4_t*)tile)[i] = s4[i];
    for (int i = n4 * 4 + threadIdx.x; i < nrow * C; i += blockDim.x) tile[i] = src[(size_t)row0 * C + i];
  } else {
    const int pitch = row_pitch(C, false);
    for (int r = 0; r < nrow; ++r) {
      const int d = threadIdx.x;
      if (d < C) tile[r * pitch + d] = src[(size_t)(row0 + r) * ld + d];
    }
  }
}
__device__ inline bool rows_contiguous(const float* src, int ld, int row0, int C) {
  return ld == C && ((C * ROWS_PER_BLOCK) & 3) == 0 && (((uintptr_t)(src + (size_t)row0 * C)) & 15) == 0;
}

// xn[p][:] = code[p][:] / max(||code[p]||, 1e-12)
__global__ __launch_bounds__(ROWS_PER_BLOCK) void normalize_rows_kernel(const float* __restrict__ code, int ldc,
                                                                        float* __restrict__ xn, int rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [128][pitch]
  const int row0 = blockIdx.x * ROWS_PER_BLOCK;
  const int pitch = row_pitch(C, rows_contiguous(code, ldc, row0, C));
  stage_rows_in(code, ldc, row0, rows, C, tile);
  __syncthreads();
  const int p = row0 + threadIdx.x;
  if (p < rows) {
    float* r = tile + threadIdx.x * pitch;
    float n2 = 0.f;
    for (int d = 0; d < C; ++d) n2 = __fadd_rn(n2, __fmul_rn(r[d], r[d]));
    const float n = fmaxf(__fsqrt_rn(n2), 1e-12f);
    for (int d = 0; d < C; ++d) r[d] = __fdiv_rn(r[d], n);
  }
  __syncthreads();
  const int nrow = min(ROWS_PER_BLOCK, rows - row0);
  if (pitch == C && ((nrow * C) & 3) == 0 && (((uintptr_t)(xn + (size_t)row0 * C)) & 15) == 0) {
    f32x4_t* d4 = (f32x4_t*)(xn + (size_t)row0 * C);
    for (int i = threadIdx.x; i < nrow * C / 4; i += blockDim.x) d4[i] = ((const f32x4_t*)tile)[i];
  } else {
    for (int i = threadIdx.x; i < nrow * C; i += blockDim.x) {
      const int r = i / C, d = i - r * C;
      xn[(size_t)row0 * C + i] = tile[r * pitch + d];
    }
  }
}

// cent[b][k][:] = xn[b][floor((2k+1) P / 2K)][:]
__global__ void km_init_kernel(const float* __restrict__ xn, float* __restrict__ cent, int P, int C, int K) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    int k = i / C, d = i - k * C;
    int p0 = (int)(((long long)(2 * k + 1) * P) / (2 * K));
    cent[(size_t)b * K * C + i] = xn[((size_t)b * P + p0) * C + d];
  }
}

// (points are read straight from global memory, one row per thread: the LDS-staged variant measured 2.7x slower here --
// its 53 KB tile leaves two 128-thread blocks per CU, and the L2 absorbs the row-strided reads of this small array)
template <int C>
__global__ __launch_bounds__(256) void km_assign_kernel(const float* __restrict__ xn, const float* __restrict__ cent,
                                                        int* __restrict__ labels, int P, int K) {
  extern __shared__ float cs[];  // [K][C]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) cs[i] = cent[(size_t)b * K * C + i];
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* xp = xn + ((size_t)b * P + p) * C;
  float x[C];
#pragma unroll
  for (int d = 0; d < C; ++d) x[d] = xp[d];
  int best = 0;
  float bv = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const float* c = cs + k * C;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < C; ++d) acc = __fadd_rn(acc, __fmul_rn(x[d], c[d]));
    if (acc > bv) { bv = acc; best = k; }
  }
  labels[(size_t)b * P + p] = best;
}

// part[b][chunk][k][d] = ordered sum of the chunk's members of cluster k ; pcnt[b][chunk][k] = member count
__global__ void km_partial_kernel(const float* __restrict__ xn, const int* __restrict__ labels,
                                  float* __restrict__ part, int* __restrict__ pcnt, int P, int C, int K, int nchunk) {
  extern __shared__ float tab[];  // [K][C] floats + [K] ints
  int* cn = (int*)(tab + K * C);
  const int chunk = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) tab[i] = 0.f;
  for (int i = threadIdx.x; i < K; i += blockDim.x) cn[i] = 0;
  __syncthreads();
  const int p0 = chunk * KM_CHUNK, p1 = min(P, p0 + KM_CHUNK);
  if (d < C) {
    for (int p = p0; p < p1; ++p) {
      const int k = labels[(size_t)b * P + p];
      tab[k * C + d] = __fadd_rn(tab[k * C + d], xn[((size_t)b * P + p) * C + d]);
    }
  }
  if (d == 0)
    for (int p = p0; p < p1; ++p) cn[labels[(size_t)b * P + p]] += 1;
  __syncthreads();
  float* dst = part + ((size_t)b * nchunk + chunk) * K * C;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) dst[i] = tab[i];
  for (int i = threadIdx.x; i < K; i += blockDim.x) pcnt[((size_t)b * nchunk + chunk) * K + i] = cn[i];
}

// cent[b][k][:] = normalise(sum over chunks, ascending) if the cluster is non-empty.
// One workgroup per (cluster, frame): thread d adds the chunk partials of (k, d) in ascending chunk order (the loads are
// independent, only the adds are chained), thread 0 forms the squared norm over d in index order -- the same arithmetic,
// in the same order, as a single workgroup per frame would do, on K times as many CUs.
__global__ __launch_bounds__(128) void km_update_kernel(const float* __restrict__ part, const int* __restrict__ pcnt,
                                                        float* __restrict__ cent, int C, int K, int nchunk) {
  __shared__ float sums[128];
  __shared__ float nrm_s;
  __shared__ int cnt_s;
  const int k = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  if (d < C) {
    float s = 0.f;
#pragma unroll 7
    for (int c = 0; c < nchunk; ++c) s = __fadd_rn(s, part[(((size_t)b * nchunk + c) * K + k) * C + d]);
    sums[d] = s;
  }
  if (d == 127) {
    int n = 0;
    for (int c = 0; c < nchunk; ++c) n += pcnt[((size_t)b * nchunk + c) * K + k];
    cnt_s = n;
  }
  __syncthreads();
  if (d == 0) {
    float n2 = 0.f;
    for (int i = 0; i < C; ++i) n2 = __fadd_rn(n2, __fmul_rn(sums[i], sums[i]));
    nrm_s = fmaxf(__fsqrt_rn(n2), 1e-12f);
  }
  __syncthreads();
  if (d < C && cnt_s > 0) cent[((size_t)b * K + k) * C + d] = __fdiv_rn(sums[d], nrm_s);
}

// compaction of the used ids to 0..K'-1 in ascending order (feature_extractor.py:245-246) + distinct count
__global__ __launch_bounds__(1024) void km_relabel_kernel(int* __restrict__ labels, int* __restrict__ nseg, int P, int K,
                                                          int relabel) {
  __shared__ int used[KM_MAXK];
  __shared__ int lut[KM_MAXK];
  const int b = blockIdx.x;
  int* lab = labels + (size_t)b * P;
  if (threadIdx.x < K) used[threadIdx.x] = 0;
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += blockDim.x) used[lab[p]] = 1;  // benign race: all writers store 1
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < K; ++k) { lut[k] = run; run += used[k]; }
    nseg[b] = run;
  }
  __syncthreads();
  if (relabel)
    for (int p = threadIdx.x; p < P; p += blockDim.x) lab[p] = lut[lab[p]];
}

template <int C>
int run_kmeans(const float* xn, int* labels, int* nseg, float* scratch, int B, int P, int K, int iters, int relabel,
               hipStream_t st) {
  const int nchunk = ceil_div(P, KM_CHUNK);
  float* cent = scratch;                                   // [B][K][C]
  float* part = cent + (size_t)B * K * C;                  // [B][nchunk][K][C]
  int* pcnt = (int*)(part + (size_t)B * nchunk * K * C);   // [B][nchunk][K]
  const size_t shm_kc = (size_t)K * C * sizeof(float);
  hipLaunchKernelGGL(km_init_kernel, dim3(B), dim3(256), 0, st, xn, cent, P, C, K);
  WVN_LAUNCH_CHECK();
  const int threads_c = ((C + 63) / 64) * 64;
  for (int it = 0; it <= iters; ++it) {
    hipLaunchKernelGGL((km_assign_kernel<C>), dim3(ceil_div(P, 256), B), dim3(256), shm_kc, st, xn, cent, labels, P, K);
    WVN_LAUNCH_CHECK();
    if (it == iters) break;
    hipLaunchKernelGGL(km_partial_kernel, dim3(nchunk, B), dim3(threads_c), shm_kc + K * sizeof(int), st, xn, labels,
                       part, pcnt, P, C, K, nchunk);
    WVN_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_update_kernel, dim3(K, B), dim3(128), 0, st, part, pcnt, cent, C, K, nchunk);
    WVN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(km_relabel_kernel, dim3(B), dim3(1024), 0, st, labels, nseg, P, K, relabel);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// out[r] = argmax_c x[r][c], lowest index wins ties (torch.argmax semantics on finite rows): the label maps of the STEGO
// cluster probe (cosine similarity against learned centroids) and linear probe (stego_interface.py:94-100)
__global__ void argmax_rows_kernel(const float* __restrict__ x, int ld, int rows, int cols, int* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* xr = x + (size_t)r * ld;
  float best = xr[0];
  int bi = 0;
  for (int c = 1; c < cols; ++c) {
    const float v = xr[c];
    if (v > best) { best = v; bi = c; }
  }
  out[r] = bi;
}

}  // namespace

int wvn_argmax_rows_launch(const float* x, int ld, int rows, int cols, int* out, hipStream_t st) {
  if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return WVN_ERR_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, st, x, ld, rows, cols, out);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_normalize_rows_launch(const float* code, int ldc, float* xn, int rows, int C, hipStream_t st) {
  if (!code || !xn) return WVN_ERR_ARG;
  if (C > ROWS_PER_BLOCK) return WVN_ERR_ARG;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(ceil_div(rows, ROWS_PER_BLOCK)), dim3(ROWS_PER_BLOCK),
                     ROWS_PER_BLOCK * (C | 1) * sizeof(float), st, code, ldc, xn, rows, C);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

size_t wvn_kmeans_scratch_floats(int B, int P, int C, int K) {
  const size_t nchunk = (size_t)ceil_div(P, KM_CHUNK);
  return (size_t)B * K * C + (size_t)B * nchunk * K * C + (size_t)B * nchunk * K;
}

int wvn_kmeans_launch(const float* xn, int* labels, int* nseg, float* scratch, int B, int P, int C, int K, int iters,
                      int relabel, hipStream_t st) {
  if (!xn || !labels || !nseg || !scratch || K <= 0 || K > KM_MAXK || P <= 0) return WVN_ERR_ARG;
  if ((size_t)K * C * sizeof(float) + 2 * K * sizeof(float) > 60 * 1024) return WVN_ERR_ARG;
  if (C == 90) return run_kmeans<90>(xn, labels, nseg, scratch, B, P, K, iters, relabel, st);
  if (C == 64) return run_kmeans<64>(xn, labels, nseg, scratch, B, P, K, iters, relabel, st);
  if (C == 16) return run_kmeans<16>(xn, labels, nseg, scratch, B, P, K, iters, relabel, st);
  return WVN_ERR_ARG;
}
