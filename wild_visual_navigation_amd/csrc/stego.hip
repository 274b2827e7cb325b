// STEGO-style per-image clustering of the patch-resolution code (stego_interface.py:94-100 as used
// by FeatureExtractor with run_clustering=True): deterministic cosine k-means.
//
// Integer outputs (segment-index maps) must be bit-exact against the oracle given the same fp32
// code, so every floating-point reduction here has a FIXED, documented order and uses the
// explicitly fused (__fmaf_rn) or explicitly separate (__fmul_rn / __fadd_rn are plain operators: this file is compiled
// with -ffp-contract=off) operations, and square roots / reciprocals formed in fp64 (rinv_norm below):
//   * dot products / norms : fused multiply-add chains over the channel index, acc = fma(a_d, b_d, acc) from +0
//   * centroid sums        : points are cut into chunks of KM_CHUNK (64) consecutive indices; inside a
//                            chunk the members of a cluster are added in ascending point order starting
//                            from 0; the chunk partials are added in ascending chunk order inside groups
//                            of km_super(P) (8; 16 above 8192 points) consecutive chunks, and the group partials in ascending group
//                            order (the three-level order lets the pixel-resolution form keep a group's
//                            running sums on chip; at every level an addition chain starts from +0)
//   * argmax               : first maximum (lowest cluster id wins ties)
// oracle/interfaces.py::kmeans_cosine_labels mirrors this operation for operation.
//
// The whole GPU works on every Lloyd iteration (3 small launches per iteration, all images at once):
//   assign  : one lane per point, centroids of the image in LDS
//   partial : one workgroup per (chunk, image), lane = channel: LDS table part[K][C] indexed by label
//             (each lane owns its column, so no atomics and a fixed order)
//   update  : one workgroup per image: ordered sum over chunks, sequential norm, divide
#include <stdlib.h>

#include "common.h"
#include "wvn_internal.h"

namespace {

// 1 / max(sqrt(n2), 1e-12) with BOTH operations correctly rounded, through fp64: a 53-bit square root / quotient of fp32
// operands rounds to the correctly rounded fp32 result (53 >= 2 * 24 + 2).  Not __fsqrt_rn / __fdiv_rn: hipcc lowered
// __fsqrt_rn to the bare v_sqrt_f32 (1 ulp) in one kernel of this file and to the refined sequence in another -- the
// reciprocal norms of 13 % of the pixels came out one ulp apart between the two k-means forms.
__device__ inline float rinv_norm(float n2) {
  const float n = fmaxf((float)sqrt((double)n2), 1e-12f);
  return (float)(1.0 / (double)n);
}

typedef __attribute__((ext_vector_type(2))) float f32x2v_t;
constexpr int KM_MAXK = 64;
constexpr int KM_CHUNK = 64;
// chunk partials are folded in groups of km_super(P) consecutive chunks: 8 (512 points) for up to 8192 points, 16 (1024 points) for
// more -- the pixel-resolution clustering of a 448 x 448 frame then folds 196 group partials per centroid value instead of 392
#ifndef WVN_KM_SUPER_BIG
#define WVN_KM_SUPER_BIG 16   // (timing experiments only: the oracles state 16; measured 8 / 16 / 24 / 32 / 48: 20.6 / 19.6 / 19.6 / 20.0 / 20.7 ms per 64-frame k-means)
#endif
__host__ __device__ inline int km_super(long long P) { return P > 8192 ? WVN_KM_SUPER_BIG : 8; }

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

// xn[p][:] = code[p][:] * (1 / max(||code[p]||, 1e-12)): ONE correctly rounded reciprocal per row, then a multiply per element
// (the pixel-resolution k-means below re-creates every row on the fly in every pass: 90 multiplies instead of 90 divisions)
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
    for (int d = 0; d < C; ++d) n2 = __fmaf_rn(r[d], r[d], n2);
    const float rinv = rinv_norm(n2);
    for (int d = 0; d < C; ++d) r[d] = __fmul_rn(r[d], rinv);
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
    for (int d = 0; d < C; ++d) acc = __fmaf_rn(x[d], c[d], acc);
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

// The pair-interleaved copy of a frame's centroids that km_pix_assign_kernel's PACKED form reads (K <= 20): for a block of 16 channels
// and a PAIR of clusters (2 kp, 2 kp + 1) the 32 floats {c_2kp[d], c_2kp+1[d]}, d ascending -- two s_load_dwordx16, every aligned
// scalar register pair of which is the second operand of one v_pk_fma_f32.  cpk[blk][kp (KM_PK_PAIRS)][j (16)][2]
constexpr int KM_PK_PAIRS = 10;
__host__ __device__ inline size_t km_pk_floats(int C) { return (size_t)((C + 15) / 16) * KM_PK_PAIRS * 32; }
__host__ __device__ inline size_t km_pk_index(int k, int d) { return ((size_t)((d >> 4) * KM_PK_PAIRS + (k >> 1)) * 16 + (d & 15)) * 2 + (k & 1); }

// cent[b][k][:] = normalise(sum over chunks, ascending) if the cluster is non-empty.
// One workgroup per (cluster, frame): thread d adds the chunk partials of (k, d) in ascending chunk order (the loads are
// independent, only the adds are chained), thread 0 forms the squared norm over d in index order -- the same arithmetic,
// in the same order, as a single workgroup per frame would do, on K times as many CUs.
__global__ __launch_bounds__(128) void km_update_kernel(const float* __restrict__ part, const int* __restrict__ pcnt,
                                                        float* __restrict__ cent, int C, int K, int nchunk, int group,
                                                        float* __restrict__ cpk = nullptr) {
  __shared__ float sums[128];
  __shared__ float nrm_s;
  __shared__ int cnt_s;
  const int k = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  if (d < C) {
    float s = 0.f;
    if (group == 1) {
      // `part` holds group partials (the pixel-resolution form: 196 of them per centroid value at 448^2): the same chain s = s + part[c],
      // c ascending, with the loads issued SIXTEEN at a time -- left as a load-add loop the kernel took 112 us per pass (one memory
      // round trip per addition) for 90 MB that stream in 25.  (0 + x in the general form below is x: the chains are identical.)
      constexpr int UB = 16;
      const float* src = part + ((size_t)b * nchunk * K + k) * C + d;
      const size_t step = (size_t)K * C;
      for (int c0 = 0; c0 < nchunk; c0 += UB) {
        float v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = src[(size_t)min(c0 + u, nchunk - 1) * step];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const float t = __fadd_rn(s, __fadd_rn(0.f, v[u]));
          s = c0 + u < nchunk ? t : s;
        }
      }
    } else {
      for (int c0 = 0; c0 < nchunk; c0 += group) {   // group = km_super(P) for chunk partials
        float gsum = 0.f;
        const int c1 = min(nchunk, c0 + group);
        for (int c = c0; c < c1; ++c) gsum = __fadd_rn(gsum, part[(((size_t)b * nchunk + c) * K + k) * C + d]);
        s = __fadd_rn(s, gsum);
      }
    }
    sums[d] = s;
  }
  if (d >= 96) {   // (the last wave of the workgroup is idle above: its 32 lanes count the members)
    int n = 0;
    for (int c = d - 96; c < nchunk; c += 32) n += pcnt[((size_t)b * nchunk + c) * K + k];
    for (int o = 16; o > 0; o >>= 1) n += __shfl_xor(n, o, 32);
    if (d == 127) cnt_s = n;
  }
  __syncthreads();
  if (d == 0) {
    float n2 = 0.f;
    for (int i = 0; i < C; ++i) n2 = __fmaf_rn(sums[i], sums[i], n2);
    nrm_s = rinv_norm(n2);
  }
  __syncthreads();
  if (d < C && cnt_s > 0) {
    const float v = __fmul_rn(sums[d], nrm_s);
    cent[((size_t)b * K + k) * C + d] = v;
    if (cpk) {   // the pair-interleaved copy the packed assign kernel reads; its unused slots mirror centroid 0
      cpk[(size_t)b * km_pk_floats(C) + km_pk_index(k, d)] = v;
      if (k == 0)
        for (int kk = K; kk < 2 * KM_PK_PAIRS; ++kk) cpk[(size_t)b * km_pk_floats(C) + km_pk_index(kk, d)] = v;
    }
  }
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
    hipLaunchKernelGGL(km_update_kernel, dim3(K, B), dim3(128), 0, st, part, pcnt, cent, C, K, nchunk, km_super(P));
    WVN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(km_relabel_kernel, dim3(B), dim3(1024), 0, st, labels, nseg, P, K, relabel);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Pixel-resolution clustering (the reading of the absent STEGO package in which postprocess() clusters the CODE PIXELS,
// stego_interface.py:94-109): the points are the H x H bilinearly up-sampled (align_corners=True), normalised code rows.
// That array is 4.6 GB per 64 frames at 448^2 and the k-means passes over it 21 times; here it never exists: every pass
// re-creates its rows from the G x G patch codes (1.1 MB per frame, L2-resident) with the same explicitly rounded
// interpolation the up-sampling kernel uses (common.h: lerp_tap / bilerp_fixed), multiplied by the row's reciprocal norm
// (computed once per pixel, 4 B).  Same algorithm, same summation orders, same bits as run_kmeans on the materialised rows.
//   rinv    : lane = pixel; the block's two source code rows in LDS
//   assign  : lane = pixel, x[C] in registers, centroids through the scalar cache (uniform addresses), four independent
//             dot-product chains in flight
//   partial : one wave per group of km_super(P) chunks, a lane owns two channels; the code values of a pixel's cell stay in
//             registers while consecutive pixels share it (the next cell's are prefetched); a cluster's running sums stay in
//             registers while the label repeats -- the addition order is exactly the sequential one.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PIX_RPB = 4;   // image rows per workgroup of the lane-per-pixel kernels (the two staged code rows serve P of them)
// Workgroup -> (part ix of the frame, frame b) for 1-D grids of nx * B workgroups.  Consecutive workgroup ids land on consecutive
// XCDs (guide: block i runs on XCD i mod 8), each with its own 4 MB L2: with whole multiples of 8 frames, every frame is worked on
// by ONE XCD (frame b on XCD b mod 8), so its 1.1 MB of patch codes, its labels and its reciprocal norms are fetched into one L2
// instead of eight (round 3 PMC: the partial-sum kernel read 1376 MB per launch against 175 MB algorithmic, the assign kernel 467).
__host__ __device__ inline int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
__device__ inline void km_frame_map(int id, int nx, int B, int& ix, int& b) {
  if ((B & 7) == 0) {
    const int xcd = id & 7, s = id >> 3, f = s / nx;
    b = f * 8 + xcd;
    ix = s - f * nx;
  } else {
    b = id / nx;
    ix = id - b * nx;
  }
}
// stage code rows y0 / y1 of frame b ([G][C] each) into LDS: rows[0][G*C], rows[1][G*C]
__device__ inline void pix_stage_rows(const float* __restrict__ code, int b, int G, int C, int y0, int y1, float* rows) {
  const int n = G * C;   // contiguous in memory
  const float* r0 = code + ((size_t)b * G * G + (size_t)y0 * G) * C;
  const float* r1 = code + ((size_t)b * G * G + (size_t)y1 * G) * C;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { rows[i] = r0[i]; rows[n + i] = r1[i]; }
}

template <int C>
__device__ inline void pix_row(const float* rows, int G, const LerpTap& tx, const LerpTap& ty, float* v) {
  const float* a0 = rows + tx.i0 * C;
  const float* a1 = rows + tx.i1 * C;
  const float* b0 = rows + G * C + tx.i0 * C;
  const float* b1 = rows + G * C + tx.i1 * C;
#pragma unroll
  for (int d = 0; d < C; ++d) {
    v[d] = bilerp_fixed(a0[d], a1[d], b0[d], b1[d], tx.w0, tx.w1, ty.w0, ty.w1);
  }
}

// rinv[b][y*H + x] = 1 / max(||v||, 1e-12), v = the interpolated code row of pixel (y, x).  The squared norm is one fma chain in
// channel order; the row is interpolated sixteen channels at a time (nothing but the chain's accumulator lives across the blocks:
// ~30 registers, and with one wave per 64 pixels of an image row the kernel fills the CU like the assign kernel does).
template <int C>
__global__ __launch_bounds__(512) void km_pix_rinv_kernel(const float* __restrict__ code, float* __restrict__ rinv, int G, int H, int B, int ac) {
  extern __shared__ float rows[];  // [2][G][C]
  int bx, b;
  km_frame_map(blockIdx.x, ceil_div_dev(H, PIX_RPB), B, bx, b);
  const float scale = lerp_scale(G, H, ac);
  int s0 = -1, s1 = -1;            // the code rows staged in LDS
  for (int y = bx * PIX_RPB; y < min(H, (bx + 1) * PIX_RPB); ++y) {   // consecutive image rows mostly share them
    const LerpTap ty = lerp_tap_ac(y, G, scale);
    if (ty.i0 != s0 || ty.i1 != s1) {
      __syncthreads();
      pix_stage_rows(code, b, G, C, ty.i0, ty.i1, rows);
      __syncthreads();
      s0 = ty.i0; s1 = ty.i1;
    }
    for (int x = threadIdx.x; x < H; x += blockDim.x) {
      const LerpTap tx = lerp_tap_ac(x, G, scale);
      const float* a0 = rows + tx.i0 * C;
      const float* a1 = rows + tx.i1 * C;
      const float* b0 = rows + G * C + tx.i0 * C;
      const float* b1 = rows + G * C + tx.i1 * C;
      float n2 = 0.f;
#pragma unroll
      for (int d0 = 0; d0 < C; d0 += 16) {
        asm volatile("" ::: "memory");   // (one block's taps in flight at a time)
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (d0 + j < C) {
            const float v = bilerp_fixed(a0[d0 + j], a1[d0 + j], b0[d0 + j], b1[d0 + j], tx.w0, tx.w1, ty.w0, ty.w1);
            n2 = __fmaf_rn(v, v, n2);
          }
      }
      rinv[(size_t)b * H * H + (size_t)y * H + x] = rinv_norm(n2);
    }
  }
}

// cent[b][k][:] = x at pixel floor((2k+1) P / 2K), P = H*H
__global__ void km_pix_init_kernel(const float* __restrict__ code, const float* __restrict__ rinv, float* __restrict__ cent, int G,
                                   int H, int C, int K, float* __restrict__ cpk, int ac = 1) {
  const int b = blockIdx.x;
  const long long P = (long long)H * H;
  const float scale = lerp_scale(G, H, ac);
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    const int k = i / C, d = i - k * C;
    const long long p0 = ((long long)(2 * k + 1) * P) / (2 * K);
    const int y = (int)(p0 / H), x = (int)(p0 - (long long)y * H);
    const LerpTap ty = lerp_tap_ac(y, G, scale), tx = lerp_tap_ac(x, G, scale);
    const float* cb = code + (size_t)b * G * G * C;
    const float v = bilerp_fixed(cb[((size_t)ty.i0 * G + tx.i0) * C + d], cb[((size_t)ty.i0 * G + tx.i1) * C + d],
                                 cb[((size_t)ty.i1 * G + tx.i0) * C + d], cb[((size_t)ty.i1 * G + tx.i1) * C + d], tx.w0, tx.w1,
                                 ty.w0, ty.w1);
    const float c = __fmul_rn(v, rinv[(size_t)b * P + p0]);
    cent[(size_t)b * K * C + i] = c;
    if (cpk) {
      cpk[(size_t)b * km_pk_floats(C) + km_pk_index(k, d)] = c;
      if (k == 0)   // the unused slots mirror centroid 0 (km_pix_assign_pk_kernel)
        for (int kk = K; kk < 2 * KM_PK_PAIRS; ++kk) cpk[(size_t)b * km_pk_floats(C) + km_pk_index(kk, d)] = c;
    }
  }
}

// Lane = pixel.  The K dot products of a pixel advance TOGETHER, sixteen channels at a time: K accumulators in registers, the
// pixel's interpolated values only for the current channel block (each dot product is still one fma chain strictly in channel
// order: blocks ascending, channels ascending within a block -- the bits of the sequential definition).  ~60 registers instead of
// the 138 of the pixel-vector-resident form, i.e. 8 waves per SIMD instead of 3: the centroid values come through the scalar
// cache (uniform addresses), and it takes that many waves to cover their latency -- the old form ran at a third of the fp32 rate
// this chip sustains (scripts/ubench/valu_rate.hip: 147 TFLOP/s of plain v_fma_f32 at 8 waves per SIMD, 95 at one).
// (Measured and not kept: the interpolation on channel PAIRS with v_pk_mul / v_pk_fma -- half its instructions, 70 registers, 21
// spilled scalars -- 17.6 ms per 64-frame k-means against 16.8; two pixels per lane -- (y, x) and (y + 1, x), three staged code rows, every scalar load feeding 32 fmas per
// lane -- 21.0 ms per 64-frame k-means against 20.4 for this form: four waves per SIMD cover less than six; earlier, on the
// pixel-vector-resident form: packed pairs 24.2 ms against 22.5, centroids in LDS 33.5 ms.)
__device__ inline f32x16_t km_sload16(const float* p) {   // 16 consecutive floats at a wave-uniform address -> SGPRs
  f32x16_t r;
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p));
  return r;
}
// the same, issued BEFORE the users of `busy` (the previous request's result): without the tie hipcc moves the sixteen fmas that read
// `busy` above the request, into the same scalar registers, and the wave waits out every request's full latency
__device__ inline f32x16_t km_sload16_before(const float* p, f32x16_t& busy) {
  f32x16_t r;
  asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=&s"(r), "+s"(busy) : "s"(p));
  return r;
}
__device__ inline void km_swait(f32x16_t& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }   // (ties the users of r to the wait)

template <int C, int KMAX, bool EXACTK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(7, 8))) void km_pix_assign_kernel(const float* __restrict__ code, const float* __restrict__ rinv,
                                                            const float* __restrict__ cent, int* __restrict__ labels, int G, int H,
                                                            int K, int B) {
  // LDS row pitch = C, unpadded: at C = 90 and G = 56 the two staged code rows are 40,320 bytes -- FOUR workgroups per CU (padding
  // the rows to 16-byte multiples for ds_read_b128 taps costs 41,216: three).  With one wave per 64 pixels of an image row (448
  // threads at 448 pixels: seven waves, none idle) and <= 72 registers that is 28 waves per CU against 21, and 7168 workgroups
  // are exactly seven rounds of 1024.
  constexpr int CP = C;
  constexpr int DB = 16;             // channels per block
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [2][G][CP]
  int bx, b;
  km_frame_map(blockIdx.x, ceil_div_dev(H, PIX_RPB), B, bx, b);
  const float scale = lerp_scale(G, H);
  const float* __restrict__ cb = cent + (size_t)b * K * C;   // uniform addresses: served by the scalar cache
  int s0 = -1, s1 = -1;            // the code rows staged in LDS
  for (int y = bx * PIX_RPB; y < min(H, (bx + 1) * PIX_RPB); ++y) {   // consecutive image rows mostly share them
    const LerpTap ty = lerp_tap(y, G, scale);
    if (ty.i0 != s0 || ty.i1 != s1) {
      __syncthreads();
      const float* r0 = code + ((size_t)b * G * G + (size_t)ty.i0 * G) * C;
      const float* r1 = code + ((size_t)b * G * G + (size_t)ty.i1 * G) * C;
      for (int i = threadIdx.x; i < G * C; i += blockDim.x) {
        const int g = i / C, d = i - g * C;
        rows[g * CP + d] = r0[i];
        rows[(G + g) * CP + d] = r1[i];
      }
      __syncthreads();
      s0 = ty.i0; s1 = ty.i1;
    }
    for (int x = threadIdx.x; x < H; x += blockDim.x) {
      const LerpTap tx = lerp_tap(x, G, scale);
      const size_t p = (size_t)b * H * H + (size_t)y * H + x;
      const float ri = rinv[p];
      const float* a0 = rows + tx.i0 * CP;
      const float* a1 = rows + tx.i1 * CP;
      const float* b0 = rows + (G + tx.i0) * CP;
      const float* b1 = rows + (G + tx.i1) * CP;
      float acc[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
#pragma unroll
      for (int d0 = 0; d0 < C; d0 += DB) {
        constexpr int dummy = 0; (void)dummy;
        const int n = C - d0 < DB ? C - d0 : DB;   // (compile-time after unrolling)
        float v[DB];
#pragma unroll
        for (int j = 0; j < DB; ++j)
          if (j < n)
            v[j] = __fmul_rn(bilerp_fixed(a0[d0 + j], a1[d0 + j], b0[d0 + j], b1[d0 + j], tx.w0, tx.w1, ty.w0, ty.w1), ri);
        // The centroid values are wave-uniform: sixteen of them arrive in SGPRs by ONE scalar load, issued by hand -- centroid
        // k + 1's while centroid k's are being used (left to hipcc, all K x 16 loads of a block are requested at once and 300 - 1400
        // scalar registers spilled, whatever the source order).  Scalar loads return out of order, so each wait is for all of them:
        // the one in flight has had sixteen fmas (x 8 waves) of cover by then.  (The last block of a 90-channel row reads 6 floats
        // past it: the next centroid's, or -- for the last centroid -- the first bytes of the partial-sum area behind `cent`.)
        // (Pairs of centroids -- two requests in flight, one wait per 32 fmas -- measured the same: 19.70 against 19.64 ms.)
        f32x16_t cq[2];
        const float* pk = cb + d0;   // ONE running address (120 precomputed ones cost more scalar registers than there are)
        cq[0] = km_sload16(pk);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          {   // (every k < KMAX is computed: for K < KMAX the extra rows read whatever follows the centroids -- finite or not, they are
              //  never compared -- which keeps the loop free of branches and the scalar registers of merges)
            km_swait(cq[k & 1]);
            if (k + 1 < KMAX) {
              pk += C;
              asm volatile("" : "+s"(pk));
              cq[(k + 1) & 1] = km_sload16_before(pk, cq[k & 1]);
            }
#pragma unroll
            for (int j = 0; j < DB; ++j)
              if (j < n) acc[k] = __fmaf_rn(v[j], cq[k & 1][j], acc[k]);
            asm volatile("" : "+v"(acc[k]));   // (the chain is finished here: its sixteen scalars are dead before the next request)
          }
        }
      }
      int best = 0;
      float bv = -INFINITY;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if ((EXACTK || k < K) && acc[k] > bv) { bv = acc[k]; best = k; }
      labels[p] = best;
    }
  }
}


// ---- the same kernel with PACKED dot products (round 4) ---------------------------------------------------------------------------
// The kernel above runs at the issue rate of plain v_fma_f32: 2440 VALU instructions per 64 pixels x 4 cycles = 0.80 ms per 64-frame
// pass, measured 0.78 -- the "147 TFLOP/s" it was sized against in round 3 is the rate of v_pk_fma_f32 (scripts/ubench/valu_rate.hip's
// plain loop had been packed by hipcc's SLP vectoriser).  A packed fma is two independent, correctly rounded fmas in one issue slot:
// here the two halves are the chains of TWO CLUSTERS over the same channel,
//     (acc_2kp, acc_2kp+1) = fma((x_d, x_d), (c_2kp[d], c_2kp+1[d]), (acc_2kp, acc_2kp+1))
// x_d broadcast by op_sel, the centroid pair an aligned scalar register pair of the pair-interleaved copy (km_pk_index) -- every chain
// still runs strictly in channel order from +0: the bits of the kernel above, half its dot-product instructions.  PKI: the
// interpolation on channel pairs as well (v_pk_mul / v_pk_fma on ds_read_b64 taps).  K < 20: the unused slots of the packed copy hold
// copies of centroid 0 (km_pix_init_kernel / km_update_kernel keep them current), so the kernel has no K at all -- twenty runtime
// masks cost twenty scalar register pairs, and under that pressure hipcc's allocator satisfied the "+s" tie of a scalar-load wait with
// COPIES of the still in-flight registers in front of the wait (first version: 203 of 44,100 labels wrong at K = 17;
// tests/test_isa_hazards.py now screens the library for reads of in-flight scalar-load destinations).
// Timing experiments (scripts/build_variant.sh ... -DWVN_KM_ASSIGN_ABL=<bits>; results are WRONG with any bit set): 2 no scalar loads after a
// pixel's first pair, 4 no LDS taps / interpolation (lane constants).  Measured (whole k-means call, 64 frames, 12.41 ms): 12.18 / 9.82 /
// both 9.65 -- the scalar loads are free, the taps + interpolation cost 235 us of a 618-us pass (their VALU share is ~105: the rest is the
// LDS pipe, 1440 bytes per pixel and pass), the dot products run at 80 % of the packed issue rate.
#ifndef WVN_KM_ASSIGN_ABL
#define WVN_KM_ASSIGN_ABL 0
#endif
template <int C, bool PKI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(7, 8))) void km_pix_assign_pk_kernel(const float* __restrict__ code, const float* __restrict__ rinv,
                                                            const float* __restrict__ cpk, int* __restrict__ labels, int G, int H,
                                                            int K, int B) {
  constexpr int CP = C;
  constexpr int DB = 16;             // channels per block
  constexpr int KP = KM_PK_PAIRS;    // cluster pairs
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [2][G][CP]
  int bx, b;
  km_frame_map(blockIdx.x, ceil_div_dev(H, PIX_RPB), B, bx, b);
  const float scale = lerp_scale(G, H);
  const float* __restrict__ cb = cpk + (size_t)b * km_pk_floats(C);   // uniform addresses: served by the scalar cache
  int s0 = -1, s1 = -1;            // the code rows staged in LDS
  for (int y = bx * PIX_RPB; y < min(H, (bx + 1) * PIX_RPB); ++y) {   // consecutive image rows mostly share them
    const LerpTap ty = lerp_tap(y, G, scale);
    if (ty.i0 != s0 || ty.i1 != s1) {
      __syncthreads();
      const float* r0 = code + ((size_t)b * G * G + (size_t)ty.i0 * G) * C;
      const float* r1 = code + ((size_t)b * G * G + (size_t)ty.i1 * G) * C;
      for (int i = threadIdx.x; i < G * C; i += blockDim.x) {
        const int g = i / C, d = i - g * C;
        rows[g * CP + d] = r0[i];
        rows[(G + g) * CP + d] = r1[i];
      }
      __syncthreads();
      s0 = ty.i0; s1 = ty.i1;
    }
    for (int x = threadIdx.x; x < H; x += blockDim.x) {
      const LerpTap tx = lerp_tap(x, G, scale);
      const size_t p = (size_t)b * H * H + (size_t)y * H + x;
      const float ri = rinv[p];
      const float* a0 = rows + tx.i0 * CP;
      const float* a1 = rows + tx.i1 * CP;
      const float* b0 = rows + (G + tx.i0) * CP;
      const float* b1 = rows + (G + tx.i1) * CP;
      f32x2v_t acc[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[k] = f32x2v_t{0.f, 0.f};
      const float* pk = cb;          // ONE running address through the frame's packed centroids: [block][pair][2 x 16 floats]
      f32x16_t cq[2][2];
#pragma unroll
      for (int d0 = 0; d0 < C; d0 += DB) {
        const int n = C - d0 < DB ? C - d0 : DB;   // (compile-time after unrolling)
        f32x2v_t v[DB / 2];                          // the block's interpolated, normalised values, channel pairs
        if constexpr ((WVN_KM_ASSIGN_ABL & 4) != 0) {
#pragma unroll
          for (int j = 0; j < DB / 2; ++j) v[j] = f32x2v_t{ri + (float)(d0 + j), tx.w0};
        } else if constexpr (PKI) {
          const f32x2v_t X0 = {tx.w0, tx.w0}, X1 = {tx.w1, tx.w1}, Y0 = {ty.w0, ty.w0}, Y1 = {ty.w1, ty.w1}, R = {ri, ri};
#pragma unroll
          for (int j = 0; j < DB / 2; ++j)
            if (2 * j < n) {   // (C even: pairs never straddle the end)
              const f32x2v_t q00 = *(const f32x2v_t*)(a0 + d0 + 2 * j), q01 = *(const f32x2v_t*)(a1 + d0 + 2 * j);
              const f32x2v_t q10 = *(const f32x2v_t*)(b0 + d0 + 2 * j), q11 = *(const f32x2v_t*)(b1 + d0 + 2 * j);
              const f32x2v_t t0 = __builtin_elementwise_fma(X1, q01, X0 * q00);
              const f32x2v_t t1 = __builtin_elementwise_fma(X1, q11, X0 * q10);
              v[j] = __builtin_elementwise_fma(Y1, t1, Y0 * t0) * R;      // (the bits of bilerp_fixed * ri, two channels at once)
            }
        } else {
#pragma unroll
          for (int j = 0; j < DB; ++j)
            if (j < n)
              v[j >> 1][j & 1] = __fmul_rn(bilerp_fixed(a0[d0 + j], a1[d0 + j], b0[d0 + j], b1[d0 + j], tx.w0, tx.w1, ty.w0, ty.w1), ri);
        }
        // a pair's 32 floats arrive by TWO scalar loads, requested while the previous pair's sixteen packed fmas run (one wait per pair:
        // scalar loads return out of order, so a wait is for all of them)
        if (d0 == 0) { cq[0][0] = km_sload16(pk); cq[0][1] = km_sload16(pk + 16); }
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
          const int cur = (d0 / DB * KP + kp) & 1;
          km_swait(cq[cur][0]); km_swait(cq[cur][1]);
          if ((d0 + DB < C || kp + 1 < KP) && !((WVN_KM_ASSIGN_ABL & 2) && (d0 > 0 || kp > 0))) {          // the next pair (of this block or the first of the next)
            pk += 32;
            asm volatile("" : "+s"(pk));
            cq[cur ^ 1][0] = km_sload16_before(pk, cq[cur][0]);
            cq[cur ^ 1][1] = km_sload16_before(pk + 16, cq[cur][1]);
          }
#pragma unroll
          for (int j = 0; j < DB; ++j)
            if (j < n) {
              const float xv = v[j >> 1][j & 1];
              const f32x2v_t xb = {xv, xv};
              const f32x2v_t cc = {cq[cur][j >> 3][2 * (j & 7)], cq[cur][j >> 3][2 * (j & 7) + 1]};
              acc[kp] = __builtin_elementwise_fma(xb, cc, acc[kp]);
            }
          asm volatile("" : "+v"(acc[kp]));   // (the chains are finished here: their scalars are dead before the next request)
        }
      }
      int best = 0;
      float bv = -INFINITY;
#pragma unroll
      for (int k = 0; k < 2 * KP; ++k)   // (no k < K: the slots K .. 2 KP - 1 hold COPIES of centroid 0 -- their similarity equals cluster 0's
        if (acc[k >> 1][k & 1] > bv) { bv = acc[k >> 1][k & 1]; best = k; }   //  bit for bit and a strict > never prefers them)
      labels[p] = best;
    }
  }
}


// ---- the assignment pass on the matrix pipe (round 4): SCREENED argmax -------------------------------------------------------
// Labels are an integer function of fp32 similarities that the definition fixes bit for bit (fmaf chains in channel order).  The fp32
// MFMA forms reproduce such a chain exactly but do not outrun the VALU here (profiles/r04b_kmeans_assign_forms.md: the 4x4x1 form issues
// at the v_fma_f32 rate, the 32-slot forms waste 37 % on K = 20).  What the 16x faster bf16 MFMA can do is DECIDE almost every pixel:
//   a_k = sum_ch (x_hi c_hi + x_hi c_lo + x_lo c_hi)      x, c_k unit vectors as hi + lo bf16 planes, fp32 accumulation
// differs from the real dot product by at most (2^-17 + 2^-17 + 2^-18) sum|x c| + 98 * 2^-24 <= 2.6e-5, the definition's chain e_k
// by at most 90 * 2^-24 <= 5.4e-6 (sum|x c| <= 1): |a_k - e_k| <= 3.2e-5 for every k.  So where the screened best beats the screened
// runner-up by more than 2 * 3.2e-5, argmax_k e_k = argmax_k a_k (and no tie is involved): the label is PROVEN without evaluating e.
// The kernel takes tau = 1.5e-4; the pixels inside the band (about 1 in 10^4: the margin density near zero is ~0.5 per unit) are
// re-done by the definition itself -- a wave re-runs an image row of 64 pixels with the exact fmaf chains when any of its lanes asks.
// Net: 3 MFMAs per 16 channels and 32 pixels instead of 20 x 16 v_fma_f32 per pixel; what remains is the interpolation (shared between
// the two image rows of an item: the horizontal half is row-independent) and the plane split.  Labels and centroids stay bit-identical
// to the VALU form by construction AND by test (tests/test_gpu_stego_pixels.py runs both forms on every shape; a forced-fallback
// switch, wvn_debug_kmeans_assign_form(2), runs the exact path for every row).
// OPT-IN (wvn_debug_kmeans_assign_form(1)), not the default: what it buys depends on how many 64-pixel row groups hold a pixel inside
// the band.  On code maps with cluster structure (scripts/bench_pixel_kmeans.py: 4.4 % of the row groups) a k-means call takes 14.2 ms
// against 15.8; on the bench's SYNTHETIC-weight code, which has none -- twenty near-parallel centroids -- 85 % of the row groups
// ask for the exact path and the headline drops from 1007 to 846 frames/s (profiles/r04b_kmeans_assign_forms.md).  A released STEGO
// checkpoint is the first case; it cannot be measured here (no network), so the VALU form stays the default.
constexpr float SCR_TAU = 1.5e-4f;   // > 2 * (3.2e-5 + 2e-6: the index tag in the five low significand bits of the screened values)
__device__ inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ unsigned long long g_scr_exact_rows[2];   // [0] 64-pixel row groups sent down the exact path, [1] row groups seen (statistics)
__host__ __device__ constexpr int scr_cp(int C) { return (C + 15) / 16 * 16; }
__host__ inline size_t scr_lds_bytes(int G, int C) { return (size_t)2 * G * scr_cp(C) * sizeof(float); }

// the definition for one pixel: K fmaf chains in channel order over the interpolated, normalised row (rows: [2][G][CP] staged code rows)
template <int C, int KMAX>
__device__ inline int km_pix_exact_label(const float* rows, int G, int CP, const LerpTap& tx, const LerpTap& ty, float ri, const float* __restrict__ cb, int K) {
  const float* a0 = rows + tx.i0 * CP;
  const float* a1 = rows + tx.i1 * CP;
  const float* b0 = rows + (G + tx.i0) * CP;
  const float* b1 = rows + (G + tx.i1) * CP;
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  constexpr int DB = 10;
  for (int d0 = 0; d0 < C; d0 += DB) {   // (a run-time loop: this path runs for about one wave in three hundred)
    float v[DB];
#pragma unroll
    for (int j = 0; j < DB; ++j) {
      const int d = min(d0 + j, C - 1);
      v[j] = __fmul_rn(bilerp_fixed(a0[d], a1[d], b0[d], b1[d], tx.w0, tx.w1, ty.w0, ty.w1), ri);
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const float* ck = cb + (size_t)min(k, K - 1) * C + d0;
#pragma unroll
      for (int j = 0; j < DB; ++j)
        if (d0 + j < C) acc[k] = __fmaf_rn(v[j], ck[j], acc[k]);
    }
  }
  int best = 0;
  float bv = -INFINITY;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K && acc[k] > bv) { bv = acc[k]; best = k; }
  return best;
}

template <int C>
__global__ __launch_bounds__(256) void km_pix_assign_screen_kernel(const float* __restrict__ code, const float* __restrict__ rinv,
                                                                   const float* __restrict__ cent, int* __restrict__ labels, int G, int H,
                                                                   int K, int B, int force_exact) {
  constexpr int CP = scr_cp(C), NS = CP / 16;
  static_assert(C % 10 == 0 || C == 16, "km_pix_exact_label walks the channels in blocks of 10");
  extern __shared__ __attribute__((aligned(16))) float rows[];   // [2][G][CP]: source rows i0 and min(i0 + 1, G - 1), zero-padded channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int i0, b;
  km_frame_map(blockIdx.x, G, B, i0, b);
  const float scale = lerp_scale(G, H);
  int yb = scale > 0.f ? max(0, (int)((float)i0 / scale) - 2) : 0;   // the run of image rows whose upper source row is i0: [yb, ye)
  while (yb < H && lerp_tap(yb, G, scale).i0 < i0) ++yb;
  int ye = yb;
  while (ye < H && lerp_tap(ye, G, scale).i0 == i0) ++ye;
  if (ye == yb) return;
  const float* cb = cent + (size_t)b * K * C;
  {
    const float* src0 = code + (size_t)b * G * G * C;
    for (int rg = wave; rg < 2 * G; rg += 4) {
      const int r = rg / G, g = rg - r * G;
      const float* src = src0 + ((size_t)min(i0 + r, G - 1) * G + g) * C;
      float* dst = rows + (size_t)rg * CP;
#pragma unroll
      for (int d = lane; d < CP; d += 64) dst[d] = d < C ? src[d] : 0.f;
    }
  }
  // centroid fragments (MFMA A operand): lane (c = l31, hi) holds channels 16 s + 8 hi + j of centroid c as hi / lo bf16 planes
  bf16x8_t ch_[NS], cl_[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    u32x4_t uh, ul;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int d = 16 * s + 8 * hi + 2 * e;
      const float v0 = (l31 < K && d < C) ? cb[(size_t)l31 * C + d] : 0.f, v1 = (l31 < K && d + 1 < C) ? cb[(size_t)l31 * C + d + 1] : 0.f;
      const uint32_t h = pack_bf16x2(v0, v1);
      uh[e] = h;
      ul[e] = pack_bf16x2(v0 - __uint_as_float(h << 16), v1 - __uint_as_float(h & 0xffff0000u));
    }
    ch_[s] = __builtin_bit_cast(bf16x8_t, uh);
    cl_[s] = __builtin_bit_cast(bf16x8_t, ul);
  }
  __syncthreads();
  const int ngx = ceil_div_dev(H, 64), npair = (ye - yb + 1) / 2;
  for (int item = wave; item < npair * ngx; item += 4) {
    const int rp = item / ngx, gx = item - rp * ngx;
    const int y0 = yb + 2 * rp, y1 = min(y0 + 1, ye - 1);
    const LerpTap t0 = lerp_tap(y0, G, scale), t1 = lerp_tap(y1, G, scale);
    unsigned need_exact = force_exact > 0 ? 3u : 0u;          // bit r: image row r of the item needs the definition
    int lab[2][2];
    if (force_exact <= 0) {
      f32x16_t acc[2][2];   // [half][row]
      LerpTap tx[2];
      float ri[2][2];
      const float* pt[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xc = min(gx * 64 + 32 * h + l31, H - 1);
        tx[h] = lerp_tap(xc, G, scale);
        ri[h][0] = rinv[(size_t)b * H * H + (size_t)y0 * H + xc];
        ri[h][1] = rinv[(size_t)b * H * H + (size_t)y1 * H + xc];
        pt[h] = rows + 8 * hi;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[h][r][e] = 0.f;
      }
      // (half, k-step) steps, software-pipelined by hand: the eight tap reads of step n + 1 are requested before step n's arithmetic --
      // two waves per SIMD do not cover an LDS round trip per step on their own (measured: 646 us per pass without the prefetch)
      struct Taps { f32x4_t a00[2], a01[2], a10[2], a11[2]; };
      auto fetch = [&](int n, Taps& t) {
        const int s = n >> 1, h = n & 1;
        const float* q00 = pt[h] + tx[h].i0 * CP + 16 * s;
        const float* q01 = pt[h] + tx[h].i1 * CP + 16 * s;
        const float* q10 = pt[h] + (G + tx[h].i0) * CP + 16 * s;
        const float* q11 = pt[h] + (G + tx[h].i1) * CP + 16 * s;
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
          t.a00[q4] = *(const f32x4_t*)(q00 + 4 * q4); t.a01[q4] = *(const f32x4_t*)(q01 + 4 * q4);
          t.a10[q4] = *(const f32x4_t*)(q10 + 4 * q4); t.a11[q4] = *(const f32x4_t*)(q11 + 4 * q4);
        }
      };
      Taps cur, nxt;
      fetch(0, cur);
#pragma unroll
      for (int n = 0; n < 2 * NS; ++n) {
        const int s = n >> 1, h = n & 1;
        if (n + 1 < 2 * NS) fetch(n + 1, nxt);
        float v0[8], v1[8];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4)
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // bilerp_fixed with the row-independent half hoisted (the same operations: the same bits)
            const float h0 = __fmaf_rn(tx[h].w1, cur.a01[q4][e], __fmul_rn(tx[h].w0, cur.a00[q4][e]));
            const float h1 = __fmaf_rn(tx[h].w1, cur.a11[q4][e], __fmul_rn(tx[h].w0, cur.a10[q4][e]));
            v0[4 * q4 + e] = __fmul_rn(__fmaf_rn(t0.w1, h1, __fmul_rn(t0.w0, h0)), ri[h][0]);
            v1[4 * q4 + e] = __fmul_rn(__fmaf_rn(t1.w1, h1, __fmul_rn(t1.w0, h0)), ri[h][1]);
          }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float* v = r ? v1 : v0;
          u32x4_t uh, ul;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t hh = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            uh[e] = hh;
            ul[e] = pack_bf16x2(v[2 * e] - __uint_as_float(hh << 16), v[2 * e + 1] - __uint_as_float(hh & 0xffff0000u));
          }
          const bf16x8_t xh = __builtin_bit_cast(bf16x8_t, uh), xl = __builtin_bit_cast(bf16x8_t, ul);
          acc[h][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch_[s], xl, acc[h][r], 0, 0, 0);
          acc[h][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl_[s], xh, acc[h][r], 0, 0, 0);
          acc[h][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch_[s], xh, acc[h][r], 0, 0, 0);
        }
        if (n + 1 < 2 * NS) cur = nxt;
      }
      // screened best / runner-up per pixel: the lane's own 16 centroid slots (slot = (e & 3) + 8 (e >> 2) + 4 hi), then the partner's
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          // the centroid index rides in the five low significand bits (2^-19 of a value <= 1: far inside the band's safety factor), so
          // best and runner-up are plain maxima: max3 chains, the best masked out for the second pass
          float key[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int k = (e & 3) + 8 * (e >> 2) + 4 * hi;
            key[e] = k < K ? __uint_as_float((__float_as_uint(acc[h][r][e]) & ~31u) | (unsigned)k) : -INFINITY;
          }
          auto max16 = [&](const float (&a)[16]) {
            float m0 = max3f(a[0], a[1], a[2]), m1 = max3f(a[3], a[4], a[5]);
            m0 = max3f(m0, a[6], a[7]); m1 = max3f(m1, a[8], a[9]);
            m0 = max3f(m0, a[10], a[11]); m1 = max3f(m1, a[12], a[13]);
            return max3f(max3f(m0, a[14], a[15]), m1, m1);
          };
          const float bv = max16(key);
          float rest[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) rest[e] = key[e] == bv ? -INFINITY : key[e];   // (keys are distinct: the index is part of them)
          const float sv = max16(rest);
          const float obv = __shfl_xor(bv, 32, 64), osv = __shfl_xor(sv, 32, 64);
          const float best = fmaxf(bv, obv), second = fmaxf(fminf(bv, obv), fmaxf(sv, osv));
          lab[h][r] = (int)(__float_as_uint(best) & 31u);    // (negative similarities order the tagged keys the other way round below
                                                             //  2^-19: inside the band, where the exact path decides)
          if (!(best - second > SCR_TAU)) need_exact |= 1u << r;   // (NaN-safe: anything not provably clear asks)
        }
      need_exact = (__any(need_exact & 1u) ? 1u : 0u) | (__any(need_exact & 2u) ? 2u : 0u);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int x = gx * 64 + 32 * h + l31;
        if (hi == 0 && x < H) {
          if (!(need_exact & 1u)) labels[(size_t)b * H * H + (size_t)y0 * H + x] = lab[h][0];
          if (!(need_exact & 2u) && y1 != y0) labels[(size_t)b * H * H + (size_t)y1 * H + x] = lab[h][1];
        }
      }
    }
    if (force_exact < 0 && lane == 0) {   // (statistics run only: wvn_debug_kmeans_assign_form(3))
      atomicAdd(&g_scr_exact_rows[1], (unsigned long long)(y1 != y0 ? 2 : 1));
      if (need_exact) atomicAdd(&g_scr_exact_rows[0], (unsigned long long)(((need_exact & 1u) ? 1 : 0) + (((need_exact & 2u) && y1 != y0) ? 1 : 0)));
    }
    if (need_exact) {   // (wave-uniform) the definition, lane = pixel of the 64-pixel group
      const int x = gx * 64 + lane, xc = min(x, H - 1);
      const LerpTap txe = lerp_tap(xc, G, scale);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int y = r ? y1 : y0;
        if (!(need_exact & (1u << r)) || (r == 1 && y1 == y0)) continue;
        const size_t p = (size_t)b * H * H + (size_t)y * H + xc;
        const int best = km_pix_exact_label<C, 20>(rows, G, CP, txe, r ? t1 : t0, rinv[p], cb, K);
        if (x < H) labels[p] = best;
      }
    }
  }
}

// The pixel-vector-resident form (x[C] in registers, four dot-product chains in flight, centroids through the scalar cache at
// hipcc's discretion): any K; what runs for K > 32, where the accumulator-resident form above has no registers left.
template <int C>
__global__ __launch_bounds__(256) void km_pix_assign_wide_kernel(const float* __restrict__ code, const float* __restrict__ rinv,
                                                            const float* __restrict__ cent, int* __restrict__ labels, int G, int H,
                                                            int K, int B) {
  extern __shared__ float rows[];  // [2][G][C]
  int bx, b;
  km_frame_map(blockIdx.x, ceil_div_dev(H, PIX_RPB), B, bx, b);
  const float scale = lerp_scale(G, H);
  const float* __restrict__ cb = cent + (size_t)b * K * C;   // uniform addresses: served by the scalar cache
  int s0 = -1, s1 = -1;            // the code rows staged in LDS
  for (int y = bx * PIX_RPB; y < min(H, (bx + 1) * PIX_RPB); ++y) {   // consecutive image rows mostly share them
  const LerpTap ty = lerp_tap(y, G, scale);
  if (ty.i0 != s0 || ty.i1 != s1) {
    __syncthreads();
    pix_stage_rows(code, b, G, C, ty.i0, ty.i1, rows);
    __syncthreads();
    s0 = ty.i0; s1 = ty.i1;
  }
  for (int x = threadIdx.x; x < H; x += blockDim.x) {
    const LerpTap tx = lerp_tap(x, G, scale);
    const size_t p = (size_t)b * H * H + (size_t)y * H + x;
    float v[C];
    pix_row<C>(rows, G, tx, ty, v);
    const float ri = rinv[p];
#pragma unroll
    for (int d = 0; d < C; ++d) v[d] = __fmul_rn(v[d], ri);
    int best = 0;
    float bv = -INFINITY;
    int k = 0;
    for (; k + 4 <= K; k += 4) {   // four independent chains; each dot product still strictly in index order
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const float* c0 = cb + (size_t)k * C;
#pragma unroll
      for (int d = 0; d < C; ++d) {
        a0 = __fmaf_rn(v[d], c0[d], a0);
        a1 = __fmaf_rn(v[d], c0[C + d], a1);
        a2 = __fmaf_rn(v[d], c0[2 * C + d], a2);
        a3 = __fmaf_rn(v[d], c0[3 * C + d], a3);
      }
      if (a0 > bv) { bv = a0; best = k; }
      if (a1 > bv) { bv = a1; best = k + 1; }
      if (a2 > bv) { bv = a2; best = k + 2; }
      if (a3 > bv) { bv = a3; best = k + 3; }
    }
    for (; k < K; ++k) {
      float a0 = 0.f;
      const float* c0 = cb + (size_t)k * C;
#pragma unroll
      for (int d = 0; d < C; ++d) a0 = __fmaf_rn(v[d], c0[d], a0);
      if (a0 > bv) { bv = a0; best = k; }
    }
    labels[p] = best;
  }
  }
}

// part[b][group][k][d] = the group's partial (chunk partials added in ascending chunk order), pcnt[b][group][k] = member count
//
// ONE wave per group; lane l owns channels 2 l and 2 l + 1 (C even, <= 128).  The kernel is bound by instruction issue, not by
// memory: per pixel there is a handful of arithmetic and a dozen wave-uniform moves, so everything that is per WAVE is paid once
// (two waves with lane = channel paid it twice, the second for 26 of its 64 lanes).  Consecutive pixels of an image row share their
// four source patches for about P pixels (8.1 at 448 / 56): the chunk is walked cell by cell -- a cell = a run of pixels with the
// same four taps -- with the NEXT cell's code values already on their way while the current cell's pixels are added (fetching the
// taps of every pixel: L1-throughput-bound, 2.1 ms per launch; one pixel at a time with dependent scalar loads: three memory round
// trips per pixel, 4.5 ms).  Per-pixel parameters (tap offsets and weights, label, reciprocal norm) are computed once per chunk,
// pixel j by lane j, and handed to all lanes through v_readlane.  A cluster's running sum of the current chunk stays in registers
// while consecutive pixels carry the same label (the usual case: a dependent chain of one v_add per pixel) and is parked in the
// LDS table tab[k][c] when the label changes -- the addition order per (cluster, channel) is exactly the pixel order.  The group
// partial lives in registers (KMAX x 2 per lane); member counts come from ballots.
//
// Round 4, after an ablation of the kernel (profiles/r04e_kmeans_packed_assign.md: of 110 shader cycles per pixel and SIMD, 52 were the
// eleven per-pixel VALU instructions -- a wave64 VALU instruction costs 4 cycles whatever it is --, 33 the per-chunk / per-cell skeleton,
// 17 the label runs, 11 the tap loads):
//   * the per-pixel weights {wx0, wx1} and 1 / norm come from LDS records the chunk's lanes wrote once (wave-uniform ds_read: an LDS
//     instruction, not three v_readlane on the VALU), four pixels of a run at a time (one address move per four pixels);
//   * member counts loop over the labels PRESENT in the chunk (1 - 4 with coherent labels) instead of all K;
//   * the chunk -> group fold touches only the clusters the chunk touched (a scalar bit mask; the others would add +0, exactly nothing);
//   * the taps are buffer loads with a SCALAR offset (the cell's, from v_readlane) and a lane-constant one: no 64-bit VALU address.
// Same additions, same order, same bits (tests/test_gpu_stego_pixels.py: against the materialised route and the CPU oracle).
#ifndef WVN_KM_PARTIAL_V2
#define WVN_KM_PARTIAL_V2 1   // 0: the round-3 kernel body (scripts/build_variant.sh A/B)
#endif
__host__ inline size_t km_pix_partial_lds(int K, int C) { return (size_t)K * C * sizeof(float) + (WVN_KM_PARTIAL_V2 ? KM_CHUNK * 12 : 0); }
template <int KMAX>
__global__ __launch_bounds__(64) void km_pix_partial_kernel(const float* __restrict__ code, const float* __restrict__ rinv,
                                                            const int* __restrict__ labels, float* __restrict__ part,
                                                            int* __restrict__ pcnt, int G, int H, int C, int K, int ngroup, int nsup, int B) {
  extern __shared__ __attribute__((aligned(8))) float tab[];   // [K][C]: the current chunk's parked sums (+ V2: the chunk's pixel records)
  int g, b;
  km_frame_map(blockIdx.x, ngroup, B, g, b);
  const int lane = threadIdx.x;
  const long long P = (long long)H * H;
  const bool act = 2 * lane < C;
  const int c2 = act ? 2 * lane : 0;   // (inactive lanes read channels 0, 1 and drop the result)
  for (int i = lane; i < K * C; i += 64) tab[i] = 0.f;
  f32x2v_t grp[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) grp[k] = f32x2v_t{0.f, 0.f};
  const float* __restrict__ cb = code + (size_t)b * G * G * C;
  const int* __restrict__ lab = labels + (size_t)b * P;
  const float* __restrict__ rv = rinv + (size_t)b * P;
  const long long g0 = (long long)g * nsup * KM_CHUNK;
  const float scale = lerp_scale(G, H);
  int mycnt = 0;                       // lane k: members of cluster k in this group
#if WVN_KM_PARTIAL_V2
  f32x2v_t* rec_w = (f32x2v_t*)(tab + K * C);            // [KM_CHUNK] {wx0, wx1}   (K * C even: 8-byte aligned)
  float* rec_r = (float*)(rec_w + KM_CHUNK);             // [KM_CHUNK] 1 / norm
  const __amdgpu_buffer_rsrc_t rs_code = __builtin_amdgcn_make_buffer_rsrc((void*)cb, 0, (int)((size_t)G * G * C * sizeof(float)), 0x00020000);
  auto tap = [&](int off) -> f32x2v_t {                  // off: wave-uniform element offset of the tap's channel 0
    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rs_code, c2 * 4, __builtin_amdgcn_readfirstlane(off) * 4, 0);
    return f32x2v_t{__uint_as_float(r[0]), __uint_as_float(r[1])};
  };
#else
  auto tap = [&](int off) -> f32x2v_t { return *(const f32x2v_t*)(cb + off + c2); };
#endif
  for (int c = 0; c < nsup; ++c) {
    const long long p0 = g0 + (long long)c * KM_CHUNK;
    if (p0 >= P) break;                                  // (uniform)
    const int n = (int)min((long long)KM_CHUNK, P - p0);
    const unsigned long long valid = n == 64 ? ~0ull : ((1ull << n) - 1);
    // ---- per-pixel parameters: lane j <-> pixel p0 + j ----
    const int pj = (int)min(p0 + lane, P - 1);
    const int yj = pj / H, xj = pj - yj * H;
    const LerpTap ty = lerp_tap(yj, G, scale), tx = lerp_tap(xj, G, scale);
    const int o00 = (ty.i0 * G + tx.i0) * C, o01 = (ty.i0 * G + tx.i1) * C, o10 = (ty.i1 * G + tx.i0) * C, o11 = (ty.i1 * G + tx.i1) * C;
    const float rj = rv[pj];
    const int kj = lab[pj];
    unsigned touched = 0;                                // (uniform) bit k: cluster k has members in this chunk
#if WVN_KM_PARTIAL_V2
    rec_w[lane] = f32x2v_t{tx.w0, tx.w1};
    rec_r[lane] = rj;
    for (unsigned long long todo = valid; todo;) {       // (uniform loop over the labels PRESENT in the chunk) member counts by ballot
      const int k = __builtin_amdgcn_readlane(kj, (int)__builtin_ctzll(todo));
      const unsigned long long m = __ballot(kj == k) & valid;
      if (lane == k) mycnt += __builtin_popcountll(m);
      touched |= 1u << (k & 31);
      todo &= ~m;
    }
#else
    for (int k = 0; k < K; ++k) {                        // (uniform loop) member counts by ballot
      const int m = __builtin_popcountll(__ballot(kj == k) & valid);
      if (lane == k) mycnt += m;
    }
#endif
    // ---- cells ----
    int j = 0;
    f32x2v_t n00 = tap(__builtin_amdgcn_readlane(o00, 0)), n01 = tap(__builtin_amdgcn_readlane(o01, 0));
    f32x2v_t n10 = tap(__builtin_amdgcn_readlane(o10, 0)), n11 = tap(__builtin_amdgcn_readlane(o11, 0));
    int kcur = -1;     // the cluster whose running sums of this chunk are in `acc` (the others are parked in tab)
    f32x2v_t acc = {0.f, 0.f};
    while (j < n) {                                      // (uniform)
      const f32x2v_t v00 = n00, v01 = n01, v10 = n10, v11 = n11;
      const int a00 = __builtin_amdgcn_readlane(o00, j), a10 = __builtin_amdgcn_readlane(o10, j);
      const unsigned long long same = __ballot(o00 == a00 && o10 == a10) & valid;
      const unsigned long long rest = ~same & valid & ~((2ull << j) - 1);   // pixels after j that are not in j's cell
      const int jend = rest ? __builtin_ctzll(rest) : n;                    // cells are contiguous runs along x
      if (jend < n) {                                    // the next cell's values: in flight during this cell's additions
        n00 = tap(__builtin_amdgcn_readlane(o00, jend)); n01 = tap(__builtin_amdgcn_readlane(o01, jend));
        n10 = tap(__builtin_amdgcn_readlane(o10, jend)); n11 = tap(__builtin_amdgcn_readlane(o11, jend));
      }
      // (a cell lies in one image row: its row weights are fetched once; the lane's two channels go through the interpolation as
      //  one packed pair -- v_pk_mul / v_pk_fma are two independent correctly rounded operations, the bits of bilerp_fixed)
      const float wy0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ty.w0), j));
      const float wy1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ty.w1), j));
      const f32x2v_t Y0 = {wy0, wy0}, Y1 = {wy1, wy1};
      auto pixel = [&](f32x2v_t w, float ri) -> f32x2v_t {   // the interpolated, normalised value of one pixel of the cell (2 channels)
        const f32x2v_t X0 = {w[0], w[0]}, X1 = {w[1], w[1]}, R = {ri, ri};
        const f32x2v_t t0 = __builtin_elementwise_fma(X1, v01, X0 * v00);
        const f32x2v_t t1 = __builtin_elementwise_fma(X1, v11, X0 * v10);
        return __builtin_elementwise_fma(Y1, t1, Y0 * t0) * R;
      };
      int jj = j;
      while (jj < jend) {                                // runs of one label inside the cell (labels are spatially coherent: usually one)
        const int k = __builtin_amdgcn_readlane(kj, jj);
        const unsigned long long other = __ballot(kj != k) & valid & ~((2ull << jj) - 1);
        const int jr = other ? min(jend, (int)__builtin_ctzll(other)) : jend;
        if (k != kcur) {                                 // park / fetch the running sums
          if (kcur >= 0) *(f32x2v_t*)(tab + kcur * C + c2) = acc;
          acc = *(const f32x2v_t*)(tab + k * C + c2);
          kcur = k;
        }
#if WVN_KM_PARTIAL_V2
        int q = jj;
        for (; q + 4 <= jr; q += 4) {                    // four pixels: independent interpolations, then the additions in pixel order
          const f32x2v_t w0 = rec_w[q], w1 = rec_w[q + 1], w2 = rec_w[q + 2], w3 = rec_w[q + 3];
          const float r0 = rec_r[q], r1 = rec_r[q + 1], r2 = rec_r[q + 2], r3 = rec_r[q + 3];
          const f32x2v_t p0v = pixel(w0, r0), p1v = pixel(w1, r1), p2v = pixel(w2, r2), p3v = pixel(w3, r3);
          acc = acc + p0v; acc = acc + p1v; acc = acc + p2v; acc = acc + p3v;
        }
        for (; q < jr; ++q) acc = acc + pixel(rec_w[q], rec_r[q]);
#else
        for (int q = jj; q < jr; ++q) {                  // the additions, in pixel order
          const float wx0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tx.w0), q));
          const float wx1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tx.w1), q));
          const float ri = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(rj), q));
          acc = acc + pixel(f32x2v_t{wx0, wx1}, ri);
        }
#endif
        jj = jr;
      }
      j = jend;
    }
    if (kcur >= 0) *(f32x2v_t*)(tab + kcur * C + c2) = acc;
    // fold the chunk into the group partial (ascending chunk order) and clear the chunk table: a lane owns its two columns
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K && (!WVN_KM_PARTIAL_V2 || KMAX > 32 || ((touched >> k) & 1u))) {   // (an untouched cluster's chunk sum is +0: x + 0 = x)
        const f32x2v_t t = *(const f32x2v_t*)(tab + k * C + c2);
        grp[k][0] = __fadd_rn(grp[k][0], t[0]);
        grp[k][1] = __fadd_rn(grp[k][1], t[1]);
        *(f32x2v_t*)(tab + k * C + c2) = f32x2v_t{0.f, 0.f};
      }
  }
  float* dst = part + ((size_t)b * ngroup + g) * K * C;
  if (act) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) *(f32x2v_t*)(dst + k * C + c2) = grp[k];
  }
  if (lane < K) pcnt[((size_t)b * ngroup + g) * K + lane] = mycnt;
}

struct PixScratch { float* cent; float* part; int* pcnt; float* rinv; float* cpk; size_t floats; };
PixScratch pix_carve(float* base, int B, int G, int H, int C, int K) {
  const size_t P = (size_t)H * H, ngroup = (P + (size_t)km_super(P) * KM_CHUNK - 1) / ((size_t)km_super(P) * KM_CHUNK);
  PixScratch s;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return base ? base + o : (float*)nullptr; };
  s.cent = take((size_t)B * K * C);
  s.part = take((size_t)B * ngroup * K * C);
  s.pcnt = (int*)take((size_t)B * ngroup * K);
  s.rinv = take((size_t)B * P);
  s.cpk = take(K <= 2 * KM_PK_PAIRS ? (size_t)B * km_pk_floats(C) : 0);   // (the packed assign kernel's copy of the centroids)
  s.floats = off;
  return s;
}

// the screened assign kernel is eligible when K <= 20 and the two staged code rows fit the LDS
static bool pixm_ok(int G, int H, int C, int K) { return K <= 20 && H >= 2 && scr_lds_bytes(G, C) <= 150 * 1024; }
int g_km_assign_form = -1;   // -1 / 5: the PACKED VALU form where eligible (K <= 20; default), 4: packed dot products only, 0: the plain
                             // VALU form, 1: the screened MFMA form where eligible (K <= 20), 2: the screened kernel with every row sent
                             // down its exact path (tests), 3: the screened kernel counting its exact rows

template <int C>
int run_kmeans_pixels(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int K, int iters, int relabel,
                      hipStream_t st) {
  const size_t P = (size_t)H * H;
  const int nsup = km_super((long long)P), ngroup = (int)((P + (size_t)nsup * KM_CHUNK - 1) / ((size_t)nsup * KM_CHUNK));
  const PixScratch s = pix_carve(scratch, B, G, H, C, K);
  const size_t shm_rows = (size_t)2 * G * C * sizeof(float), shm_rows_pad = shm_rows;
  const int assign_threads = H >= 512 ? 512 : (H + 63) / 64 * 64;   // one wave per 64 pixels of an image row
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(96 * 1024, (const void*)km_pix_rinv_kernel<C>, (const void*)km_pix_assign_kernel<C, 20, true>,
                                (const void*)km_pix_assign_kernel<C, 32, false>, (const void*)km_pix_assign_wide_kernel<C>,
                                (const void*)km_pix_assign_pk_kernel<C, false>, (const void*)km_pix_assign_pk_kernel<C, true>)) return rc;
  const bool mfma = g_km_assign_form >= 1 && g_km_assign_form <= 3 && pixm_ok(G, H, C, K);
  if (mfma) {   // (the opt-in screened kernel's 150 KB attribute must not be able to fail the default path)
    static LdsOptIn lds_opt_in_m;
    if (const int rc = lds_opt_in_m(150 * 1024, (const void*)km_pix_assign_screen_kernel<C>)) return rc;
  }
  const int nrb = ceil_div(H, PIX_RPB);
  hipLaunchKernelGGL((km_pix_rinv_kernel<C>), dim3(nrb * B), dim3(H >= 512 ? 512 : (H + 63) / 64 * 64), shm_rows, st, code,
                     s.rinv, G, H, B, 1);
  const int pkform = K > 2 * KM_PK_PAIRS ? 0 : (g_km_assign_form < 0 || g_km_assign_form == 5) ? 5 : (g_km_assign_form == 4 ? 4 : 0);
  hipLaunchKernelGGL(km_pix_init_kernel, dim3(B), dim3(256), 0, st, code, s.rinv, s.cent, G, H, C, K, pkform ? s.cpk : (float*)nullptr);
  WVN_LAUNCH_CHECK();
  for (int it = 0; it <= iters; ++it) {
    const dim3 ga(nrb * B);
    if (mfma) hipLaunchKernelGGL((km_pix_assign_screen_kernel<C>), dim3(G * B), dim3(256), scr_lds_bytes(G, C), st, code, s.rinv, s.cent, labels, G, H, K, B, g_km_assign_form == 2 ? 1 : (g_km_assign_form == 3 ? -1 : 0));
    else if (pkform == 4) hipLaunchKernelGGL((km_pix_assign_pk_kernel<C, false>), ga, dim3(assign_threads), shm_rows_pad, st, code, s.rinv, s.cpk, labels, G, H, K, B);
    else if (pkform == 5) hipLaunchKernelGGL((km_pix_assign_pk_kernel<C, true>), ga, dim3(assign_threads), shm_rows_pad, st, code, s.rinv, s.cpk, labels, G, H, K, B);
    else if (K == 20) hipLaunchKernelGGL((km_pix_assign_kernel<C, 20, true>), ga, dim3(assign_threads), shm_rows_pad, st, code, s.rinv, s.cent, labels, G, H, K, B);
    else if (K <= 32) hipLaunchKernelGGL((km_pix_assign_kernel<C, 32, false>), ga, dim3(assign_threads), shm_rows_pad, st, code, s.rinv, s.cent, labels, G, H, K, B);
    else hipLaunchKernelGGL((km_pix_assign_wide_kernel<C>), ga, dim3(256), shm_rows, st, code, s.rinv, s.cent, labels, G, H, K, B);
    WVN_LAUNCH_CHECK();
    if (it == iters) break;
    if (K <= 20)   // (fewer group-partial registers: 5 waves per SIMD instead of 4)
      hipLaunchKernelGGL(km_pix_partial_kernel<20>, dim3(ngroup * B), dim3(64), km_pix_partial_lds(K, C), st, code, s.rinv, labels,
                         s.part, s.pcnt, G, H, C, K, ngroup, nsup, B);
    else if (K <= 32)
      hipLaunchKernelGGL(km_pix_partial_kernel<32>, dim3(ngroup * B), dim3(64), km_pix_partial_lds(K, C), st, code, s.rinv, labels,
                         s.part, s.pcnt, G, H, C, K, ngroup, nsup, B);
    else
      hipLaunchKernelGGL(km_pix_partial_kernel<KM_MAXK>, dim3(ngroup * B), dim3(64), km_pix_partial_lds(K, C), st, code, s.rinv,
                         labels, s.part, s.pcnt, G, H, C, K, ngroup, nsup, B);
    WVN_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_update_kernel, dim3(K, B), dim3(128), 0, st, s.part, s.pcnt, s.cent, C, K, ngroup, 1, pkform ? s.cpk : (float*)nullptr);
    WVN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(km_relabel_kernel, dim3(B), dim3(1024), 0, st, labels, nseg, (int)P, K, relabel);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// out[b][gy][gx][:] = 0.5 * (a[b][gy][gx][:] + m[b][gy][G-1-gx][:]): the code averaged with the flipped-back code of the mirrored
// frame (the flip pass of the upstream get_code)
__global__ void flip_average_kernel(const float* __restrict__ a, const float* __restrict__ m, float* __restrict__ out, long long n,
                                    int G, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const long long r = i / C;
  const int gx = (int)(r % G);
  const long long row = r / G;
  out[i] = __fmul_rn(__fadd_rn(a[i], m[(row * G + (G - 1 - gx)) * C + c]), 0.5f);
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

size_t wvn_kmeans_pixels_scratch_floats(int B, int G, int H, int C, int K) { return pix_carve(nullptr, B, G, H, C, K).floats; }

int wvn_kmeans_pixels_launch(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int C, int K, int iters,
                             int relabel, hipStream_t st) {
  if (!code || !labels || !nseg || !scratch || K <= 0 || K > KM_MAXK || G <= 0 || H <= 0 || B <= 0) return WVN_ERR_ARG;
  if ((size_t)2 * G * C * sizeof(float) > 96 * 1024 || (size_t)K * C * sizeof(float) > 60 * 1024 || (C & 1) || C > 128) return WVN_ERR_ARG;
  if (C == 90) return run_kmeans_pixels<90>(code, labels, nseg, scratch, B, G, H, K, iters, relabel, st);
  if (C == 16) return run_kmeans_pixels<16>(code, labels, nseg, scratch, B, G, H, K, iters, relabel, st);
  return WVN_ERR_ARG;
}

// the two steps the linear form of the pixel k-means (csrc/stego_linear.hip) shares with the direct one: rinv[b][p] of every code pixel
// and the initial centroids c_k = x at pixel floor((2k+1) P / 2K); and the ascending compaction of the used ids
int wvn_km_pix_prepare_launch(const float* code, float* rinv, float* cent, int B, int G, int H, int C, int K, hipStream_t st, int ac) {
  if ((size_t)2 * G * C * sizeof(float) > 96 * 1024) return WVN_ERR_ARG;
  const int nrb = ceil_div(H, PIX_RPB), threads = H >= 512 ? 512 : (H + 63) / 64 * 64;
  const size_t shm_rows = (size_t)2 * G * C * sizeof(float);
  static LdsOptIn opt;
  if (const int rc = opt(96 * 1024, (const void*)km_pix_rinv_kernel<90>, (const void*)km_pix_rinv_kernel<16>)) return rc;
  if (C == 90) hipLaunchKernelGGL((km_pix_rinv_kernel<90>), dim3(nrb * B), dim3(threads), shm_rows, st, code, rinv, G, H, B, ac);
  else if (C == 16) hipLaunchKernelGGL((km_pix_rinv_kernel<16>), dim3(nrb * B), dim3(threads), shm_rows, st, code, rinv, G, H, B, ac);
  else return WVN_ERR_ARG;
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(km_pix_init_kernel, dim3(B), dim3(256), 0, st, code, rinv, cent, G, H, C, K, (float*)nullptr, ac);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
int wvn_km_relabel_launch(int* labels, int* nseg, int B, long long P, int K, int relabel, hipStream_t st) {
  hipLaunchKernelGGL(km_relabel_kernel, dim3(B), dim3(1024), 0, st, labels, nseg, (int)P, K, relabel);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

void wvn_kmeans_pixels_set_assign_form(int form) { g_km_assign_form = form; }
// statistics of the screened kernel since the last call with reset != 0: out[0] = 64-pixel row groups re-done exactly, out[1] = all
int wvn_kmeans_pixels_screen_stats(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scr_exact_rows), 16);
  if (e != hipSuccess) return (int)e;
  if (reset) { const unsigned long long z[2] = {0, 0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_scr_exact_rows), z, 16); }
  return e == hipSuccess ? WVN_OK : (int)e;
}

int wvn_flip_average_launch(const float* a, const float* mirrored, float* out, int B, int G, int C, hipStream_t st) {
  if (!a || !mirrored || !out || B <= 0 || G <= 0 || C <= 0) return WVN_ERR_ARG;
  const long long n = (long long)B * G * G * C;
  hipLaunchKernelGGL(flip_average_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, mirrored, out, n, G, C);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
