// The pixel-resolution cosine k-means of the STEGO stage (stego_interface.py:94-109 as this build reads it), computed through its
// LINEARITY (round 5; oracle/kmeans_linear.py states the definition, operation for operation):
//
//   points        x_p = rinv_p * sum_t w_{p,t} code_t       the H x H bilinearly up-sampled (align_corners=True), normalised patch codes
//   assignment    argmax_k <x_p, c_k> = argmax_k sum_t w_{p,t} S[t, k],  S = code . c^T   (rinv_p > 0 moves no argmax)
//   centroid sums sum_{p in k} x_p = sum_t A[k, t] code_t,   A[k, t] = sum_{p in k} rinv_p w_{p,t}
//
// so a pass is: one [G*G, C] x [C, K] product (the similarity TABLE), per pixel the fixed-order bilinear interpolation of K table values
// and a first-maximum argmax, per band of image rows the summed tap weights of every (cluster, patch), and one [K, G*G] x [G*G, C]
// product -- about 1/20 of the arithmetic of interpolating C channels and forming K dot products of length C per pixel and pass
// (csrc/stego.hip: km_pix_*), and none of its H*H*C-sized streams.  The sums have FIXED orders (below), so labels and centroids are
// reproducible bit for bit and equal oracle/kmeans_linear.py's; against the direct statement they differ only where two similarities are
// closer than fp32 rounding (tests/test_oracle_stego.py, tests/test_gpu_stego_linear.py).
//
// Kernels of one pass (all frames at once, a frame pinned to one XCD's L2 as in stego.hip):
//   km_lin_assign   one workgroup per (frame, band b = the image rows whose upper tap is patch row b): the two table rows in LDS;
//                   lane = pixel: K bilinear interpolations + argmax; then U[y, k, j] (ascending x), P0 / P1[b, k, j] (ascending y)
//   km_lin_rowsum   one workgroup per (frame, patch row i): A[k, i, :] = P1[i-1] + P0[i] (+ P1[i] on the last row), R[i, k, d] = chain_j A code
//   km_lin_table    sums = sum_i R (ascending), normalise, keep the centroids of empty clusters, then S[t, :] for 128 patches per workgroup
// This translation unit is compiled with -ffp-contract=off: every fused operation is an explicit __fmaf_rn.
#include "common.h"
#include "wvn_internal.h"

namespace {

// 1 / max(sqrt(n2), 1e-12), both operations correctly rounded through fp64 (the reason is in stego.hip: rinv_norm)
__device__ inline float lin_rinv_norm(float n2) {
  const float n = fmaxf((float)sqrt((double)n2), 1e-12f);
  return (float)(1.0 / (double)n);
}

// workgroup id -> (part ix, frame b): with whole multiples of 8 frames every frame is worked on by ONE XCD (block i runs on XCD i mod 8)
__device__ inline void lin_frame_map(int id, int nx, int B, int& ix, int& b) {
  if ((B & 7) == 0) {
    const int xcd = id & 7, s = id >> 3, f = s / nx;
    b = f * 8 + xcd;
    ix = s - f * nx;
  } else {
    b = id / nx;
    ix = id - b * nx;
  }
}

constexpr int LIN_THREADS = 320;   // km_lin_assign: five waves (at 448^2 / 56 x 56: 5 rows x 7 wave tasks, 5 x 56 = 280 (row, patch column) tasks)
constexpr int LIN_RC = 5;          // image rows of a band worked on at a time

struct LinLds {   // byte offsets into km_lin_assign's dynamic LDS
  int S, U, P, rinv, xw0, xw1, xj, first, cnt, lab, bytes;
};
__host__ __device__ inline LinLds lin_lds(int G, int H, int KP, int RC) {
  const int HP = (H + 63) / 64 * 64;
  LinLds l;
  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 15) / 16 * 16; return r; };
  l.S = take(2 * G * KP * 4);
  l.U = take(RC * KP * G * 4);
  l.P = take(2 * KP * G * 4);
  l.rinv = take(RC * HP * 4);
  l.xw0 = take(HP * 4);
  l.xw1 = take(HP * 4);
  l.xj = take(HP * 4);
  l.first = take((G + 2) * 4);
  l.cnt = take(KP * 4);
  l.lab = take(RC * HP);
  l.bytes = o;
  return l;
}

// FINAL: the last assignment only -- labels to global memory as int32, no weight tables
template <int KP, bool FINAL>
__global__ __launch_bounds__(LIN_THREADS) void km_lin_assign_kernel(const float* __restrict__ S, const float* __restrict__ rinv,
                                                                    float* __restrict__ Pg, int* __restrict__ cntp,
                                                                    int* __restrict__ labels, int G, int H, int K, int B, int RC) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const LinLds L = lin_lds(G, H, KP, RC);
  float* Sl = (float*)(lds + L.S);        // [2][G][KP]
  float* Ul = (float*)(lds + L.U);        // [RC][KP][G]
  float* Pl = (float*)(lds + L.P);        // [2][KP][G]
  float* rl = (float*)(lds + L.rinv);     // [RC][HP]
  float* xw0 = (float*)(lds + L.xw0);     // [HP]   (the taps of index o are the same along x and along y: square frames)
  float* xw1 = (float*)(lds + L.xw1);
  int* xj = (int*)(lds + L.xj);           // [HP] i0 of the tap
  int* first = (int*)(lds + L.first);     // [G + 1]: first[j] = the first index whose i0 >= j (H if none)
  int* cntl = (int*)(lds + L.cnt);        // [KP]
  unsigned char* labl = lds + L.lab;      // [RC][HP]
  const int HP = (H + 63) / 64 * 64;
  int band, b;
  lin_frame_map(blockIdx.x, G, B, band, b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const size_t T = (size_t)G * G;
  const float scale = lerp_scale(G, H);
  for (int i = tid; i <= G; i += blockDim.x) first[i] = H;
  if (tid < KP) cntl[tid] = 0;
  for (int o = tid; o < H; o += blockDim.x) {
    const LerpTap t = lerp_tap(o, G, scale);
    xw0[o] = t.w0; xw1[o] = t.w1; xj[o] = t.i0;
  }
  for (int i = tid; i < 2 * KP * G; i += blockDim.x) Pl[i] = 0.f;
  __syncthreads();
  for (int o = tid; o < H; o += blockDim.x) {
    const int prev = o == 0 ? -1 : xj[o - 1];
    for (int j = prev + 1; j <= xj[o]; ++j) first[j] = o;   // (distinct writers: i0 is monotone)
  }
  __syncthreads();
  const int ylo = first[band], yhi = first[band + 1];
  const int band1 = band + (band < G - 1 ? 1 : 0);
  {   // the two table rows of the band
    const float* s0 = S + ((size_t)b * T + (size_t)band * G) * KP;
    const float* s1 = S + ((size_t)b * T + (size_t)band1 * G) * KP;
    const int n4 = G * KP / 4;   // KP is a multiple of 4 and the table is 16-byte aligned
    for (int i = tid; i < n4; i += blockDim.x) {
      ((f32x4_t*)Sl)[i] = ((const f32x4_t*)s0)[i];
      ((f32x4_t*)Sl)[n4 + i] = ((const f32x4_t*)s1)[i];
    }
  }
  __syncthreads();
  const int nseg = HP >> 6;
  int mycnt = 0;   // lane k: members of cluster k seen by this wave
  for (int c0 = ylo; c0 < yhi; c0 += RC) {
    const int rows = min(RC, yhi - c0);
    // ---- labels of the chunk's pixels: one wave per 64 pixels of a row ----
    for (int task = wave; task < rows * nseg; task += nwave) {
      const int r = task / nseg, x = (task - r * nseg) * 64 + lane, y = c0 + r;
      const bool valid = x < H;
      const int xc = valid ? x : H - 1;
      const int j0 = xj[xc], j1 = j0 + (j0 < G - 1 ? 1 : 0);
      const float wx0 = xw0[xc], wx1 = xw1[xc], wy0 = xw0[y], wy1 = xw1[y];
      if (!FINAL) rl[r * HP + x] = valid ? rinv[(size_t)b * H * H + (size_t)y * H + x] : 0.f;
      const float* a0 = Sl + j0 * KP;
      const float* a1 = Sl + j1 * KP;
      const float* b0 = a0 + G * KP;
      const float* b1 = a1 + G * KP;
      float best = -INFINITY;
      int bi = 0;
#pragma unroll
      for (int k4 = 0; k4 < KP / 4; ++k4) {
        const f32x4_t v00 = *(const f32x4_t*)(a0 + 4 * k4), v01 = *(const f32x4_t*)(a1 + 4 * k4);
        const f32x4_t v10 = *(const f32x4_t*)(b0 + 4 * k4), v11 = *(const f32x4_t*)(b1 + 4 * k4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = bilerp_fixed(v00[q], v01[q], v10[q], v11[q], wx0, wx1, wy0, wy1);
          if (4 * k4 + q < K && v > best) { best = v; bi = 4 * k4 + q; }
        }
      }
      if (FINAL) {
        if (valid) labels[(size_t)b * H * H + (size_t)y * H + x] = bi;
      } else {
        labl[r * HP + x] = (unsigned char)bi;
        for (unsigned long long todo = __ballot(valid); todo;) {   // (uniform loop over the labels present among the 64 pixels)
          const int k = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(todo));
          const unsigned long long m = __ballot(bi == k && valid);
          if (lane == k) mycnt += __builtin_popcountll(m);
          todo &= ~m;
        }
      }
    }
    if (FINAL) continue;
    __syncthreads();
    // ---- U[r][k][j]: the summed column weights of row r, cluster k, patch column j, pixels in ascending x (tap 0 before tap 1) ----
    for (int task = tid; task < rows * G; task += blockDim.x) {
      const int r = task / G, j = task - r * G;
      float* Ub = Ul + (size_t)r * KP * G + j;
      for (int k = 0; k < K; ++k) Ub[k * G] = 0.f;
      const int xa = first[max(j - 1, 0)], xb = first[j + 1];
      int cur = -1;          // the cluster whose running sum is in `acc` (the others are parked in Ub)
      float acc = 0.f;
      for (int x = xa; x < xb; ++x) {
        const int jx = xj[x], jx1 = jx + (jx < G - 1 ? 1 : 0);
        const int l = labl[r * HP + x];
        const float rr = rl[r * HP + x];
        if (jx == j || jx1 == j) {
          if (l != cur) {
            if (cur >= 0) Ub[cur * G] = acc;
            acc = Ub[l * G];
            cur = l;
          }
          if (jx == j) acc = __fadd_rn(acc, __fmul_rn(rr, xw0[x]));
          if (jx1 == j) acc = __fadd_rn(acc, __fmul_rn(rr, xw1[x]));
        }
      }
      if (cur >= 0) Ub[cur * G] = acc;
    }
    __syncthreads();
    // ---- P0 / P1[k][j]: chains over the band's rows in ascending y ----
    for (int t = tid; t < K * G; t += blockDim.x) {
      float p0 = Pl[t], p1 = Pl[KP * G + t];
      for (int r = 0; r < rows; ++r) {
        const float u = Ul[(size_t)r * KP * G + t];
        p0 = __fmaf_rn(xw0[c0 + r], u, p0);
        p1 = __fmaf_rn(xw1[c0 + r], u, p1);
      }
      Pl[t] = p0; Pl[KP * G + t] = p1;
    }
    __syncthreads();
  }
  if (FINAL) return;
  if (lane < K && mycnt) atomicAdd(&cntl[lane], mycnt);
  __syncthreads();
  float* dst = Pg + ((size_t)b * G + band) * 2 * KP * G;
  for (int i = tid; i < 2 * KP * G; i += blockDim.x) dst[i] = Pl[i];
  if (tid < KP) cntp[((size_t)b * G + band) * KP + tid] = cntl[tid];
}

// R[b][i][k][d] = chain_j A[k][i][j] * code[b][i][j][d]; A[k][i][:] from the band tables that touch patch row i, bands ascending, P0 before P1
template <int KP>
__global__ __launch_bounds__(128) void km_lin_rowsum_kernel(const float* __restrict__ code, const float* __restrict__ Pg,
                                                            float* __restrict__ R, int G, int C, int K, int B) {
  extern __shared__ __attribute__((aligned(16))) float Al[];   // [G][KP]
  int i, b;
  lin_frame_map(blockIdx.x, G, B, i, b);
  const int tid = threadIdx.x;
  const float* Pb = Pg + (size_t)b * G * 2 * KP * G;
  for (int t = tid; t < KP * G; t += blockDim.x) {
    const int k = t / G, j = t - k * G;
    float a = 0.f;
    if (k < K) {
      if (i >= 1) a = __fadd_rn(a, Pb[((size_t)(i - 1) * 2 + 1) * KP * G + t]);
      a = __fadd_rn(a, Pb[((size_t)i * 2 + 0) * KP * G + t]);
      if (i == G - 1) a = __fadd_rn(a, Pb[((size_t)i * 2 + 1) * KP * G + t]);
    }
    Al[j * KP + k] = a;
  }
  __syncthreads();
  const int d = tid;
  if (d >= C) return;
  float acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) acc[k] = 0.f;
  const float* cr = code + (((size_t)b * G + i) * G) * C + d;
  constexpr int UJ = 8;
  for (int j0 = 0; j0 < G; j0 += UJ) {
    float x[UJ];
#pragma unroll
    for (int u = 0; u < UJ; ++u) x[u] = cr[(size_t)min(j0 + u, G - 1) * C];
#pragma unroll
    for (int u = 0; u < UJ; ++u) {
      if (j0 + u < G) {
#pragma unroll
        for (int k4 = 0; k4 < KP / 4; ++k4) {
          const f32x4_t a4 = *(const f32x4_t*)(Al + (j0 + u) * KP + 4 * k4);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * k4 + q] = __fmaf_rn(a4[q], x[u], acc[4 * k4 + q]);
        }
      }
    }
  }
  float* dst = R + (((size_t)b * G + i) * KP) * C + d;
#pragma unroll
  for (int k = 0; k < KP; ++k)
    if (k < K) dst[(size_t)k * C] = acc[k];
}

// (unless `first`) sums[k][d] = sum_i R[b][i][k][d] in ascending i, counts, normalisation, empty clusters keep their centroid;
// then S[b][t][k] = chain_d code[b][t][d] * c_k[d] for the workgroup's 128 patches (thread = patch, its code row from an LDS tile)
constexpr int LIN_TT = 128;
template <int KP>
__global__ __launch_bounds__(LIN_TT) void km_lin_table_kernel(const float* __restrict__ code, const float* __restrict__ R,
                                                              const int* __restrict__ cntp, const float* __restrict__ cent_old,
                                                              float* __restrict__ cent_new, float* __restrict__ S, int G, int C, int K,
                                                              int B, int first) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = G * G, NT = (T + LIN_TT - 1) / LIN_TT;
  float* cl = sm;                        // [C][KP]
  float* tile = cl + C * KP;             // [LIN_TT][C]   (first used as sums[K][C])
  float* nrm = tile + LIN_TT * C;        // [KP]
  int* cn = (int*)(nrm + KP);            // [KP]
  int part, b;
  lin_frame_map(blockIdx.x, NT, B, part, b);
  const int tid = threadIdx.x;
  if (!first) {
    float* sums = tile;
    for (int e = tid; e < K * C; e += blockDim.x) {
      const int k = e / C, d = e - k * C;
      const float* src = R + ((size_t)b * G * KP + k) * C + d;
      const size_t step = (size_t)KP * C;
      float s = 0.f;
      constexpr int UB = 8;
      for (int i0 = 0; i0 < G; i0 += UB) {
        float v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = src[(size_t)min(i0 + u, G - 1) * step];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const float t = __fadd_rn(s, v[u]);
          s = i0 + u < G ? t : s;
        }
      }
      sums[e] = s;
    }
    if (tid < KP) {
      int n = 0;
      if (tid < K)
        for (int i = 0; i < G; ++i) n += cntp[((size_t)b * G + i) * KP + tid];
      cn[tid] = n;
    }
    __syncthreads();
    if (tid < K) {
      float n2 = 0.f;
      for (int d = 0; d < C; ++d) n2 = __fmaf_rn(sums[tid * C + d], sums[tid * C + d], n2);
      nrm[tid] = lin_rinv_norm(n2);
    }
    __syncthreads();
    for (int e = tid; e < KP * C; e += blockDim.x) {
      const int k = e / C, d = e - k * C;
      float v = 0.f;
      if (k < K) {
        v = cn[k] > 0 ? __fmul_rn(sums[e], nrm[k]) : cent_old[((size_t)b * K + k) * C + d];
        if (part == 0) cent_new[((size_t)b * K + k) * C + d] = v;
      }
      cl[d * KP + k] = v;
    }
  } else {
    for (int e = tid; e < KP * C; e += blockDim.x) {
      const int k = e / C, d = e - k * C;
      cl[d * KP + k] = k < K ? cent_old[((size_t)b * K + k) * C + d] : 0.f;
    }
  }
  __syncthreads();
  const int t0 = part * LIN_TT, nrow = min(LIN_TT, T - t0);
  const float* src = code + ((size_t)b * T + t0) * C;
  if (((nrow * C) & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
    for (int i = tid; i < nrow * C / 4; i += blockDim.x) ((f32x4_t*)tile)[i] = ((const f32x4_t*)src)[i];
  } else {
    for (int i = tid; i < nrow * C; i += blockDim.x) tile[i] = src[i];
  }
  __syncthreads();
  if (tid >= nrow) return;
  const float* row = tile + tid * C;
  float acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) acc[k] = 0.f;
  for (int d = 0; d < C; ++d) {
    const float x = row[d];
#pragma unroll
    for (int k4 = 0; k4 < KP / 4; ++k4) {
      const f32x4_t c4 = *(const f32x4_t*)(cl + d * KP + 4 * k4);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[4 * k4 + q] = __fmaf_rn(x, c4[q], acc[4 * k4 + q]);
    }
  }
  float* dst = S + ((size_t)b * T + t0 + tid) * KP;
#pragma unroll
  for (int k4 = 0; k4 < KP / 4; ++k4) *(f32x4_t*)(dst + 4 * k4) = f32x4_t{acc[4 * k4], acc[4 * k4 + 1], acc[4 * k4 + 2], acc[4 * k4 + 3]};
}

struct LinScratch { float *cent0, *cent1, *rinv, *S, *Pg, *R; int* cntp; size_t floats; };
LinScratch lin_carve(float* base, int B, int G, int H, int C, int K, int KP) {
  LinScratch s;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 63) / 64 * 64; return base ? base + o : (float*)nullptr; };
  s.cent0 = take((size_t)B * K * C);   // (first: include/wvn_hip.h promises the final centroids here)
  s.cent1 = take((size_t)B * K * C);
  s.rinv = take((size_t)B * H * H);
  s.S = take((size_t)B * G * G * KP);
  s.Pg = take((size_t)B * G * 2 * KP * G);
  s.R = take((size_t)B * G * KP * C);
  s.cntp = (int*)take((size_t)B * G * KP);
  s.floats = off;
  return s;
}
inline int lin_kp(int K) { return K <= 8 ? 8 : K <= 20 ? 20 : 32; }
int g_lin_rc = LIN_RC;

template <int KP>
int run_linear(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int C, int K, int iters, int relabel,
               hipStream_t st) {
  const LinScratch s = lin_carve(scratch, B, G, H, C, K, KP);
  const int RC = g_lin_rc;
  const LinLds L = lin_lds(G, H, KP, RC);
  const size_t table_lds = ((size_t)C * KP + (size_t)LIN_TT * C + 2 * KP) * sizeof(float);
  static LdsOptIn opt;
  if (const int rc = opt(160 * 1024, (const void*)km_lin_assign_kernel<KP, false>, (const void*)km_lin_assign_kernel<KP, true>,
                         (const void*)km_lin_table_kernel<KP>)) return rc;
  if (const int rc = wvn_km_pix_prepare_launch(code, s.rinv, s.cent0, B, G, H, C, K, st)) return rc;
  const int T = G * G, NT = (T + LIN_TT - 1) / LIN_TT;
  float* cur = s.cent0;
  float* nxt = s.cent1;
  hipLaunchKernelGGL((km_lin_table_kernel<KP>), dim3(NT * B), dim3(LIN_TT), table_lds, st, code, s.R, s.cntp, cur, nxt, s.S, G, C, K, B, 1);
  WVN_LAUNCH_CHECK();
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL((km_lin_assign_kernel<KP, false>), dim3(G * B), dim3(LIN_THREADS), L.bytes, st, s.S, s.rinv, s.Pg, s.cntp, labels, G, H, K, B, RC);
    WVN_LAUNCH_CHECK();
    hipLaunchKernelGGL((km_lin_rowsum_kernel<KP>), dim3(G * B), dim3(128), (size_t)G * KP * sizeof(float), st, code, s.Pg, s.R, G, C, K, B);
    WVN_LAUNCH_CHECK();
    hipLaunchKernelGGL((km_lin_table_kernel<KP>), dim3(NT * B), dim3(LIN_TT), table_lds, st, code, s.R, s.cntp, cur, nxt, s.S, G, C, K, B, 0);
    WVN_LAUNCH_CHECK();
    float* t = cur; cur = nxt; nxt = t;
  }
  hipLaunchKernelGGL((km_lin_assign_kernel<KP, true>), dim3(G * B), dim3(LIN_THREADS), L.bytes, st, s.S, s.rinv, s.Pg, s.cntp, labels, G, H, K, B, RC);
  WVN_LAUNCH_CHECK();
  if (cur != s.cent0) {
    const hipError_t e = hipMemcpyAsync(s.cent0, cur, (size_t)B * K * C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  return wvn_km_relabel_launch(labels, nseg, B, (long long)H * H, K, relabel, st);
}

}  // namespace

int wvn_kmeans_pixels_linear_supported(int G, int H, int C, int K) {
  if (G <= 0 || H <= 0 || K <= 0 || K > 32 || (C != 90 && C != 16)) return 0;
  const int KP = lin_kp(K);
  if (lin_lds(G, H, KP, g_lin_rc).bytes > 160 * 1024) return 0;
  if (((size_t)C * KP + (size_t)LIN_TT * C + 2 * KP) * sizeof(float) > 160 * 1024) return 0;
  return 1;
}

size_t wvn_kmeans_pixels_linear_scratch_floats(int B, int G, int H, int C, int K) { return lin_carve(nullptr, B, G, H, C, K, lin_kp(K)).floats; }

void wvn_kmeans_pixels_linear_set_rows(int rc) { g_lin_rc = rc >= 1 && rc <= 16 ? rc : LIN_RC; }

// labels[b][y][x] = first-maximum argmax over k < K of the fixed-order bilinear interpolation (align_corners=True) of table[b][.][k]:
// the km_lin_assign kernel on a caller-made table -- the STEGO cluster probe and linear probe at pixel resolution (a probe is linear
// in the code and the bilinear weights sum to one, so interpolating its K outputs equals applying it to the interpolated code)
int wvn_table_slots(int K) { return K > 0 && K <= 32 ? lin_kp(K) : 0; }
int wvn_table_bilerp_argmax_launch(const float* table, int* labels, int B, int G, int H, int K, hipStream_t st) {
  if (!table || !labels || B <= 0 || G <= 0 || H <= 0 || K <= 0 || K > 32 || (((uintptr_t)table) & 15)) return WVN_ERR_ARG;
  const int KP = lin_kp(K), RC = 1;
  const LinLds L = lin_lds(G, H, KP, RC);
  if (L.bytes > 160 * 1024) return WVN_ERR_ARG;
  static LdsOptIn opt;
  if (const int rc = opt(160 * 1024, (const void*)km_lin_assign_kernel<8, true>, (const void*)km_lin_assign_kernel<20, true>,
                         (const void*)km_lin_assign_kernel<32, true>)) return rc;
  const dim3 grid(G * B), block(LIN_THREADS);
  if (KP == 8) hipLaunchKernelGGL((km_lin_assign_kernel<8, true>), grid, block, L.bytes, st, table, (const float*)nullptr, (float*)nullptr, (int*)nullptr, labels, G, H, K, B, RC);
  else if (KP == 20) hipLaunchKernelGGL((km_lin_assign_kernel<20, true>), grid, block, L.bytes, st, table, (const float*)nullptr, (float*)nullptr, (int*)nullptr, labels, G, H, K, B, RC);
  else hipLaunchKernelGGL((km_lin_assign_kernel<32, true>), grid, block, L.bytes, st, table, (const float*)nullptr, (float*)nullptr, (int*)nullptr, labels, G, H, K, B, RC);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_kmeans_pixels_linear_launch(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int C, int K,
                                    int iters, int relabel, hipStream_t st) {
  if (!code || !labels || !nseg || !scratch || B <= 0 || iters < 0 || !wvn_kmeans_pixels_linear_supported(G, H, C, K)) return WVN_ERR_ARG;
  switch (lin_kp(K)) {
    case 8: return run_linear<8>(code, labels, nseg, scratch, B, G, H, C, K, iters, relabel, st);
    case 20: return run_linear<20>(code, labels, nseg, scratch, B, G, H, C, K, iters, relabel, st);
    default: return run_linear<32>(code, labels, nseg, scratch, B, G, H, C, K, iters, relabel, st);
  }
}
