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
//   km_lin_rowsum   one workgroup per (frame, pair of patch rows): A[k, i, :] = P1[i-1] + P0[i] (+ P1[i] on the last row),
//                   R[i, k, d] = chain_j A code, Q[g, k, d] = the group's R in ascending i (registers)
//   km_lin_table    sums = sum_g Q (ascending), normalise, keep the centroids of empty clusters, then S[t, :] for 256 patches per workgroup
//   (the last assignment leaves a bit mask of the ids in use; km_lin_relabel compacts them ascending)
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

#ifndef WVN_LIN_ABL
#define WVN_LIN_ABL 0   // (timing experiments only, scripts/build_variant.sh: 1 = no U phase, 2 = no P phase, 4 = no rinv loads, 8 = no member counts)
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2l_t;
constexpr int LIN_THREADS = 320;   // km_lin_assign: five waves (at 448^2 / 56 x 56: 5 rows x 7 wave tasks, 5 x 56 = 280 (row, patch column) tasks)
constexpr int LIN_RC = 4;          // image rows of a band worked on at a time (measured 3 / 4 / 5 / 9: 3.46 / 3.31 / 3.75 / 5.05 ms per 64-frame k-means)

struct LinLds {   // byte offsets into km_lin_assign's dynamic LDS
  int S, U, P, rinv, xrec, first, cnt, lab, bytes;
};
__host__ __device__ inline LinLds lin_lds(int G, int H, int KP, int RC) {
  const int HP = (H + 63) / 64 * 64;
  LinLds l;
  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 15) / 16 * 16; return r; };
  l.S = take(2 * G * KP * 4);
  l.U = take(RC * KP * G * 4);
  l.P = take(2 * KP * G * 4);
  l.rinv = take((RC * H * 4 + 1023) / 1024 * 1024);   // [RC][H], copied by whole 1 KB direct-to-LDS pieces
  l.xrec = take(HP * 16);                               // per index o: {i0 (int bits), w0, w1, -}: the taps are the same along x and along y
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
                                                                    int* __restrict__ labels, unsigned* __restrict__ used, int G, int H,
                                                                    int K, int B, int RC, int ac) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const LinLds L = lin_lds(G, H, KP, RC);
  float* Sl = (float*)(lds + L.S);        // [2][G][KP]
  float* Ul = (float*)(lds + L.U);        // [RC][KP][G]
  float* Pl = (float*)(lds + L.P);        // [2][KP][G]
  float* rl = (float*)(lds + L.rinv);     // [RC][H]
  f32x4_t* xrec = (f32x4_t*)(lds + L.xrec);   // [HP] {i0, w0, w1, -}
  int* first = (int*)(lds + L.first);     // [G + 1]: first[j] = the first index whose i0 >= j (H if none)
  int* cntl = (int*)(lds + L.cnt);        // [KP]
  unsigned char* labl = lds + L.lab;      // [RC][HP]
  const int HP = (H + 63) / 64 * 64;
  int band, b;
  lin_frame_map(blockIdx.x, G, B, band, b);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = blockDim.x >> 6;
  const size_t T = (size_t)G * G;
  const float scale = lerp_scale(G, H, ac);   // ac = 0: half-pixel (align_corners=False) taps -- the band structure below only needs i0 monotone and i1 = min(i0 + 1, G - 1)
  for (int i = tid; i <= G; i += blockDim.x) first[i] = H;
  if (tid < KP) cntl[tid] = 0;
  for (int o = tid; o < HP; o += blockDim.x) {
    const LerpTap t = lerp_tap_ac(min(o, H - 1), G, scale);
    xrec[o] = f32x4_t{__int_as_float(t.i0), t.w0, t.w1, 0.f};
  }
  for (int i = tid; i < 2 * KP * G; i += blockDim.x) Pl[i] = 0.f;
  __syncthreads();
  for (int o = tid; o < H; o += blockDim.x) {
    const int prev = o == 0 ? -1 : __float_as_int(xrec[o - 1][0]), me = __float_as_int(xrec[o][0]);
    for (int j = prev + 1; j <= me; ++j) first[j] = o;   // (distinct writers: i0 is monotone)
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
  unsigned mymask = 0;   // the labels this lane has assigned
  // the chunk's reciprocal norms travel global -> LDS directly (buffer_load ... lds, 1 KB per wave instruction), requested before the
  // labels are computed and waited for behind them: as register loads in front of LDS stores they cost the kernel 2.6 of its 4.2 ms
  // (a memory round trip per wave task, then -- hoisted -- 57 registers and a workgroup of occupancy)
  const bool dma = !FINAL && (H & 3) == 0 && !(WVN_LIN_ABL & 4);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(rinv + (FINAL ? 0 : (size_t)b * H * H)), 0,
                                                                          FINAL ? 0u : (unsigned)((size_t)H * H * sizeof(float)), 0x00020000);
  for (int c0 = ylo; c0 < yhi; c0 += RC) {
    const int rows = min(RC, yhi - c0);
    if (dma) {
      const int npiece = (rows * H * 4 + 1023) >> 10;
      for (int pc = wave; pc < npiece; pc += nwave)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (__attribute__((address_space(3))) void*)((unsigned char*)rl + pc * 1024), 16, lane * 16,
                                                 __builtin_amdgcn_readfirstlane((c0 * H) * 4 + pc * 1024), 0, 0);
    }
    // ---- labels of the chunk's pixels: one wave per 64 pixels of a row ----
    for (int task = wave; task < rows * nseg; task += nwave) {
      const int r = task / nseg, x = (task - r * nseg) * 64 + lane, y = c0 + r;
      const bool valid = x < H;
      const f32x4_t tx = xrec[x], ty = xrec[y];
      const int j0 = __float_as_int(tx[0]), j1 = j0 + (j0 < G - 1 ? 1 : 0);
      const float wx0 = tx[1], wx1 = tx[2], wy0 = ty[1], wy1 = ty[2];
      const float* a0 = Sl + j0 * KP;
      const float* a1 = Sl + j1 * KP;
      const float* b0 = a0 + G * KP;
      const float* b1 = a1 + G * KP;
      float best = -INFINITY;
      int bi = 0;
      const f32x2l_t X0 = {wx0, wx0}, X1 = {wx1, wx1}, Y0 = {wy0, wy0}, Y1 = {wy1, wy1};
#pragma unroll
      for (int k4 = 0; k4 < KP / 4; ++k4) {
        const f32x4_t v00 = *(const f32x4_t*)(a0 + 4 * k4), v01 = *(const f32x4_t*)(a1 + 4 * k4);
        const f32x4_t v10 = *(const f32x4_t*)(b0 + 4 * k4), v11 = *(const f32x4_t*)(b1 + 4 * k4);
        // (two clusters per v_pk_mul_f32 / v_pk_fma_f32: the halves are independent correctly rounded operations -- the bits of bilerp_fixed)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2l_t c00 = {v00[2 * h], v00[2 * h + 1]}, c01 = {v01[2 * h], v01[2 * h + 1]};
          const f32x2l_t c10 = {v10[2 * h], v10[2 * h + 1]}, c11 = {v11[2 * h], v11[2 * h + 1]};
          const f32x2l_t t0 = __builtin_elementwise_fma(X1, c01, X0 * c00);
          const f32x2l_t t1 = __builtin_elementwise_fma(X1, c11, X0 * c10);
          const f32x2l_t v = __builtin_elementwise_fma(Y1, t1, Y0 * t0);
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (4 * k4 + 2 * h + q < K && v[q] > best) { best = v[q]; bi = 4 * k4 + 2 * h + q; }
        }
      }
      // which clusters have members is all the update needs ("an empty cluster keeps its centroid"): every lane collects the ids it has
      // assigned in a bit mask, folded once per workgroup at the end (a ballot loop over the labels present per 64 pixels cost 0.3 ms per call)
      if (valid) mymask |= 1u << bi;
      if (FINAL) {
        if (valid) labels[(size_t)b * H * H + (size_t)y * H + x] = bi;
      } else {
        labl[r * HP + x] = (unsigned char)bi;
      }
    }
    if (FINAL) continue;
    if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      for (int e = tid; e < rows * H; e += blockDim.x) rl[e] = rinv[(size_t)b * H * H + (size_t)c0 * H + e];
    __syncthreads();
    // ---- U[r][k][j]: the summed column weights of row r, cluster k, patch column j, pixels in ascending x (tap 0 before tap 1).
    //      Four pixels' operands are fetched per LDS round trip; the additions stay one chain in pixel order ----
    for (int task = tid; task < ((WVN_LIN_ABL & 1) ? 0 : rows * G); task += blockDim.x) {
      const int r = task / G, j = task - r * G;
      float* Ub = Ul + (size_t)r * KP * G + j;
      for (int k = 0; k < K; ++k) Ub[k * G] = 0.f;
      const int xa = first[max(j - 1, 0)], xb = first[j + 1];
      int cur = -1;          // the cluster whose running sum is in `acc` (the others are parked in Ub)
      float acc = 0.f;
      const unsigned char* lrow = labl + r * HP;
      const float* rrow = rl + r * H;
      for (int x0 = xa; x0 < xb; x0 += 4) {
        f32x4_t tx[4];
        int l[4];
        float rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int x = min(x0 + u, H - 1);
          tx[u] = xrec[x]; l[u] = lrow[x]; rr[u] = rrow[x];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jx = __float_as_int(tx[u][0]), jx1 = jx + (jx < G - 1 ? 1 : 0);
          if (x0 + u < xb && (jx == j || jx1 == j)) {
            if (l[u] != cur) {
              if (cur >= 0) Ub[cur * G] = acc;
              acc = Ub[l[u] * G];
              cur = l[u];
            }
            if (jx == j) acc = __fadd_rn(acc, __fmul_rn(rr[u], tx[u][1]));
            if (jx1 == j) acc = __fadd_rn(acc, __fmul_rn(rr[u], tx[u][2]));
          }
        }
      }
      if (cur >= 0) Ub[cur * G] = acc;
    }
    __syncthreads();
    // ---- P0 / P1[k][j]: chains over the band's rows in ascending y ----
    for (int t = tid; t < ((WVN_LIN_ABL & 2) ? 0 : K * G); t += blockDim.x) {
      float p0 = Pl[t], p1 = Pl[KP * G + t];
      for (int r0 = 0; r0 < rows; r0 += 4) {
        float u[4];
        f32x4_t ty[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = min(r0 + q, rows - 1);
          u[q] = Ul[(size_t)r * KP * G + t]; ty[q] = xrec[c0 + r];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (r0 + q < rows) {
            p0 = __fmaf_rn(ty[q][1], u[q], p0);
            p1 = __fmaf_rn(ty[q][2], u[q], p1);
          }
      }
      Pl[t] = p0; Pl[KP * G + t] = p1;
    }
    __syncthreads();
  }
  // the ids in use: OR over the wave (butterfly), one LDS atomic per wave, one global word per workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mymask |= (unsigned)__shfl_xor((int)mymask, o, 64);
  if (lane == 0 && mymask) atomicOr((unsigned*)&cntl[0], mymask);
  __syncthreads();
  if (FINAL) {   // for the ascending compaction (km_lin_relabel_kernel)
    if (tid == 0 && cntl[0] && used) atomicOr(&used[b], (unsigned)cntl[0]);
    return;
  }
  float* dst = Pg + ((size_t)b * G + band) * 2 * KP * G;
  for (int i = tid; i < 2 * KP * G; i += blockDim.x) dst[i] = Pl[i];
  if (tid == 0) cntp[(size_t)b * G + band] = cntl[0];   // bit k: cluster k has members in this band
}

// Q[b][g][k][d] = sum over the patch rows i of group g (LIN_RG consecutive rows, ascending, plain adds from +0) of
// R[i][k][d] = chain_j A[k][i][j] * code[b][i][j][d];  A[k][i][:] from the band tables that touch patch row i, bands ascending, P0 before P1.
// One 128-thread team per patch row of the group (thread = channel d, the row's whole code column requested at once: the chains are
// short and a memory round trip per eight values made the kernel latency-bound), the group's rows then added in order through LDS.
constexpr int LIN_RG = 2;   // patch rows per workgroup (measured 8 / 4 / 2: 88 / 81 / 56 us per pass -- 896 workgroups of four rows were 1.17 rounds of the chip: the tail ran alone)
constexpr int LIN_MAXG = 64;   // patch columns a thread keeps in registers
__host__ __device__ inline size_t lin_rowsum_lds(int G, int C, int K, int KP) { return ((size_t)LIN_RG * G * KP + (size_t)LIN_RG * K * C) * sizeof(float); }
template <int KP>
__global__ __launch_bounds__(LIN_RG * 128) void km_lin_rowsum_kernel(const float* __restrict__ code, const float* __restrict__ Pg,
                                                                     float* __restrict__ Q, int G, int C, int K, int B) {
  extern __shared__ __attribute__((aligned(16))) float Al[];   // [LIN_RG][G][KP], then R [LIN_RG][K][C]
  float* Rl = Al + (size_t)LIN_RG * G * KP;
  const int NG = (G + LIN_RG - 1) / LIN_RG;
  int g, b;
  lin_frame_map(blockIdx.x, NG, B, g, b);
  const int tid = threadIdx.x;
  const int i0 = g * LIN_RG, nr = min(LIN_RG, G - i0);
  const float* Pb = Pg + (size_t)b * G * 2 * KP * G;
  const int r = tid >> 7, d = tid & 127;
  const bool act = r < nr && d < C;
  constexpr int XH = LIN_MAXG / 2;   // the row's code column in two halves of 32 patch columns (64 registers of it kept the kernel at three waves per SIMD)
  float x[XH];
  const float* cr = code + (((size_t)b * G + i0 + (r < nr ? r : 0)) * G) * C + (d < C ? d : 0);
  if (act) {
#pragma unroll
    for (int j = 0; j < XH; ++j) x[j] = cr[(size_t)min(j, G - 1) * C];
  }
  for (int e0 = tid; e0 < nr * KP * G; e0 += 4 * blockDim.x) {   // (four elements' band tables requested per round trip)
    float pa[4], pb[4], pc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + u * (int)blockDim.x, nr * KP * G - 1);
      const int rr = e / (KP * G), t = e - rr * KP * G, i = i0 + rr;
      pa[u] = i >= 1 ? Pb[((size_t)(i - 1) * 2 + 1) * KP * G + t] : 0.f;
      pb[u] = Pb[((size_t)i * 2 + 0) * KP * G + t];
      pc[u] = i == G - 1 ? Pb[((size_t)i * 2 + 1) * KP * G + t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      if (e < nr * KP * G) {
        const int rr = e / (KP * G), t = e - rr * KP * G, k = t / G, j = t - k * G, i = i0 + rr;
        float a = 0.f;
        if (k < K) {
          if (i >= 1) a = __fadd_rn(a, pa[u]);
          a = __fadd_rn(a, pb[u]);
          if (i == G - 1) a = __fadd_rn(a, pc[u]);
        }
        Al[((size_t)rr * G + j) * KP + k] = a;
      }
    }
  }
  __syncthreads();
  if (act) {
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.f;
    const float* Ar = Al + (size_t)r * G * KP;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1) {   // the second half is requested once the first has been consumed (the same chain order: j ascending)
#pragma unroll
        for (int j = 0; j < XH; ++j) x[j] = cr[(size_t)min(XH + j, G - 1) * C];
      }
#pragma unroll
      for (int j = 0; j < XH; ++j) {
        if (h * XH + j < G) {
#pragma unroll
          for (int k4 = 0; k4 < KP / 4; ++k4) {
            const f32x4_t a4 = *(const f32x4_t*)(Ar + (h * XH + j) * KP + 4 * k4);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * k4 + q] = __fmaf_rn(a4[q], x[j], acc[4 * k4 + q]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k)
      if (k < K) Rl[((size_t)r * K + k) * C + d] = acc[k];
  }
  __syncthreads();
  float* dst = Q + ((size_t)b * NG + g) * KP * C;
  for (int e = tid; e < K * C; e += blockDim.x) {
    float s = 0.f;
    for (int rr = 0; rr < nr; ++rr) s = __fadd_rn(s, Rl[(size_t)rr * K * C + e]);
    dst[e] = s;   // [k][d] with pitch C inside a KP * C slot
  }
}

// (unless `first`) sums[k][d] = sum_g Q[b][g][k][d] in ascending g, counts, normalisation, empty clusters keep their centroid;
// then S[b][t][k] = chain_d code[b][t][d] * c_k[d] for the workgroup's 256 patches (thread = patch: its code row straight from global
// memory, every request of the row in flight at once -- the LDS-staged tile cost a memory round trip per 16 bytes of a thread's copy loop)
constexpr int LIN_TT = 256;
constexpr int LIN_MAXNG = 32;
template <int KP, int C>
__global__ __launch_bounds__(LIN_TT) void km_lin_table_kernel(const float* __restrict__ code, const float* __restrict__ Q,
                                                              const int* __restrict__ cntp, const float* __restrict__ cent_old,
                                                              float* __restrict__ cent_new, float* __restrict__ S, int G, int K,
                                                              int B, int first) {
  __shared__ __attribute__((aligned(16))) float cl[C * KP];   // [C][KP]
  __shared__ float sums[KP * C];
  __shared__ float nrm[KP];
  __shared__ int cn[KP];
  const int T = G * G, NT = (T + LIN_TT - 1) / LIN_TT, NG = (G + LIN_RG - 1) / LIN_RG;
  int part, b;
  lin_frame_map(blockIdx.x, NT, B, part, b);
  const int tid = threadIdx.x;
  if (!first) {
    constexpr int NV = (KP * C + LIN_TT - 1) / LIN_TT;
    float v[NV][LIN_MAXNG > 8 ? 8 : LIN_MAXNG];
    // every partial of the thread's values requested before the first addition (8 groups at a time)
    for (int g0 = 0; g0 < NG; g0 += 8) {
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const int e = tid + m * LIN_TT, k = e / C, d = e - k * C;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[m][u] = (e < K * C && g0 + u < NG) ? Q[(((size_t)b * NG + g0 + u) * KP + k) * C + d] : 0.f;
      }
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const int e = tid + m * LIN_TT;
        if (e < K * C) {
          float sacc = g0 == 0 ? 0.f : sums[e];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (g0 + u < NG) sacc = __fadd_rn(sacc, v[m][u]);
          sums[e] = sacc;
        }
      }
    }
    if (tid < KP) {
      unsigned m = 0;
      for (int i = 0; i < G; ++i) m |= (unsigned)cntp[(size_t)b * G + i];
      cn[tid] = (tid < K && ((m >> tid) & 1u)) ? 1 : 0;   // cluster tid has members
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
      float val = 0.f;
      if (k < K) {
        val = cn[k] > 0 ? __fmul_rn(sums[e], nrm[k]) : cent_old[((size_t)b * K + k) * C + d];
        if (part == 0) cent_new[((size_t)b * K + k) * C + d] = val;
      }
      cl[d * KP + k] = val;
    }
  } else {
    for (int e = tid; e < KP * C; e += blockDim.x) {
      const int k = e / C, d = e - k * C;
      cl[d * KP + k] = k < K ? cent_old[((size_t)b * K + k) * C + d] : 0.f;
    }
  }
  __syncthreads();
  const int t = part * LIN_TT + tid;
  if (t >= T) return;
  static_assert((C & 1) == 0, "code rows are read as 8-byte pairs");
  const f32x2l_t* row = (const f32x2l_t*)(code + ((size_t)b * T + t) * C);   // (C even: every row is 8-byte aligned)
  f32x2l_t x2[C / 2];
#pragma unroll
  for (int i = 0; i < C / 2; ++i) x2[i] = row[i];
  float acc[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) acc[k] = 0.f;
#pragma unroll
  for (int d = 0; d < C; ++d) {
    const float x = x2[d >> 1][d & 1];
#pragma unroll
    for (int k4 = 0; k4 < KP / 4; ++k4) {
      const f32x4_t c4 = *(const f32x4_t*)(cl + d * KP + 4 * k4);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[4 * k4 + q] = __fmaf_rn(x, c4[q], acc[4 * k4 + q]);
    }
  }
  float* dst = S + ((size_t)b * T + t) * KP;
#pragma unroll
  for (int k4 = 0; k4 < KP / 4; ++k4) *(f32x4_t*)(dst + 4 * k4) = f32x4_t{acc[4 * k4], acc[4 * k4 + 1], acc[4 * k4 + 2], acc[4 * k4 + 3]};
}

// ascending compaction of the ids in use (feature_extractor.py:245-246) from the mask the final assignment left: nseg[b] = their
// number; labels -> rank of the id among the used ones (relabel != 0)
__global__ __launch_bounds__(256) void km_lin_relabel_kernel(int* __restrict__ labels, int* __restrict__ nseg, const unsigned* __restrict__ used,
                                                             long long P, int relabel, int nblk) {
  const int b = blockIdx.x / nblk, part = blockIdx.x - b * nblk;
  const unsigned m = used[b];
  if (part == 0 && threadIdx.x == 0) nseg[b] = __builtin_popcount(m);
  if (!relabel) return;
  // the frame's labels start at element b * P of the buffer: 16-byte aligned only when that is a multiple of four (odd H: not for b >= 1).  A scalar head of
  // 0 - 3 labels up to the next aligned element, the aligned body as int4, a scalar tail
  int* base = labels + (size_t)b * P;
  const long long head = min((long long)((4 - (((uintptr_t)base >> 2) & 3)) & 3), P);
  int4* lab = (int4*)(base + head);
  const long long n4 = (P - head) / 4;
  for (long long i = (long long)part * blockDim.x + threadIdx.x; i < n4; i += (long long)nblk * blockDim.x) {
    int4 v = lab[i];
    v.x = __builtin_popcount(m & ((1u << v.x) - 1u)); v.y = __builtin_popcount(m & ((1u << v.y) - 1u));
    v.z = __builtin_popcount(m & ((1u << v.z) - 1u)); v.w = __builtin_popcount(m & ((1u << v.w) - 1u));
    lab[i] = v;
  }
  if (part == 0) {
    for (long long p = threadIdx.x; p < head; p += blockDim.x) base[p] = __builtin_popcount(m & ((1u << base[p]) - 1u));
    for (long long p = head + n4 * 4 + threadIdx.x; p < P; p += blockDim.x) base[p] = __builtin_popcount(m & ((1u << base[p]) - 1u));
  }
}

struct LinScratch { float *cent0, *cent1, *rinv, *S, *Pg, *Q; int* cntp; unsigned* used; size_t floats; };
LinScratch lin_carve(float* base, int B, int G, int H, int C, int K, int KP) {
  LinScratch s;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 63) / 64 * 64; return base ? base + o : (float*)nullptr; };
  s.cent0 = take((size_t)B * K * C);   // (first: include/wvn_hip.h promises the final centroids here)
  s.cent1 = take((size_t)B * K * C);
  s.rinv = take((size_t)B * H * H);
  s.S = take((size_t)B * G * G * KP);
  s.Pg = take((size_t)B * G * 2 * KP * G);
  s.Q = take((size_t)B * ((G + LIN_RG - 1) / LIN_RG) * KP * C);
  s.cntp = (int*)take((size_t)B * G);   // per (frame, band): bit k = cluster k has members
  s.used = (unsigned*)take((size_t)B);
  s.floats = off;
  return s;
}
inline int lin_kp(int K) { return K <= 8 ? 8 : K <= 20 ? 20 : 32; }
int g_lin_rc = LIN_RC;

template <int KP, int C>
int run_linear(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int K, int iters, int relabel,
               hipStream_t st, int ac) {
  const LinScratch s = lin_carve(scratch, B, G, H, C, K, KP);
  const int RC = g_lin_rc;
  const LinLds L = lin_lds(G, H, KP, RC);
  const size_t rowsum_lds = lin_rowsum_lds(G, C, K, KP);
  static LdsOptIn opt;
  if (const int rc = opt(160 * 1024, (const void*)km_lin_assign_kernel<KP, false>, (const void*)km_lin_assign_kernel<KP, true>,
                         (const void*)km_lin_rowsum_kernel<KP>)) return rc;
  if (const int rc = wvn_km_pix_prepare_launch(code, s.rinv, s.cent0, B, G, H, C, K, st, ac)) return rc;
  const int T = G * G, NT = (T + LIN_TT - 1) / LIN_TT, NG = (G + LIN_RG - 1) / LIN_RG;
  float* cur = s.cent0;
  float* nxt = s.cent1;
  hipLaunchKernelGGL((km_lin_table_kernel<KP, C>), dim3(NT * B), dim3(LIN_TT), 0, st, code, s.Q, s.cntp, cur, nxt, s.S, G, K, B, 1);
  WVN_LAUNCH_CHECK();
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL((km_lin_assign_kernel<KP, false>), dim3(G * B), dim3(LIN_THREADS), L.bytes, st, s.S, s.rinv, s.Pg, s.cntp, labels, s.used, G, H, K, B, RC, ac);
    WVN_LAUNCH_CHECK();
    hipLaunchKernelGGL((km_lin_rowsum_kernel<KP>), dim3(NG * B), dim3(LIN_RG * 128), rowsum_lds, st, code, s.Pg, s.Q, G, C, K, B);
    WVN_LAUNCH_CHECK();
    hipLaunchKernelGGL((km_lin_table_kernel<KP, C>), dim3(NT * B), dim3(LIN_TT), 0, st, code, s.Q, s.cntp, cur, nxt, s.S, G, K, B, 0);
    WVN_LAUNCH_CHECK();
    float* t = cur; cur = nxt; nxt = t;
  }
  hipError_t e = hipMemsetAsync(s.used, 0, (size_t)B * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((km_lin_assign_kernel<KP, true>), dim3(G * B), dim3(LIN_THREADS), L.bytes, st, s.S, s.rinv, s.Pg, s.cntp, labels, s.used, G, H, K, B, RC, ac);
  WVN_LAUNCH_CHECK();
  if (cur != s.cent0) {
    e = hipMemcpyAsync(s.cent0, cur, (size_t)B * K * C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  const long long P = (long long)H * H;
  const int nblk = relabel ? (int)min((long long)64, (P / 4 + 255) / 256 > 0 ? (P / 4 + 255) / 256 : 1) : 1;
  hipLaunchKernelGGL(km_lin_relabel_kernel, dim3(nblk * B), dim3(256), 0, st, labels, nseg, s.used, P, relabel, nblk);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

int wvn_kmeans_pixels_linear_supported(int G, int H, int C, int K) {
  if (G <= 0 || H <= 0 || K <= 0 || K > 32 || (C != 90 && C != 16)) return 0;
  const int KP = lin_kp(K);
  if (lin_lds(G, H, KP, g_lin_rc).bytes > 160 * 1024) return 0;
  if (lin_rowsum_lds(G, C, K, KP) > 160 * 1024 || (G + LIN_RG - 1) / LIN_RG > LIN_MAXNG || G > LIN_MAXG) return 0;
  return 1;
}

size_t wvn_kmeans_pixels_linear_scratch_floats(int B, int G, int H, int C, int K) { return lin_carve(nullptr, B, G, H, C, K, lin_kp(K)).floats; }

void wvn_kmeans_pixels_linear_set_rows(int rc) { g_lin_rc = rc >= 1 && rc <= 16 ? rc : LIN_RC; }

// labels[b][y][x] = first-maximum argmax over k < K of the fixed-order bilinear interpolation (align_corners=True) of table[b][.][k]:
// the km_lin_assign kernel on a caller-made table -- the STEGO cluster probe and linear probe at pixel resolution (a probe is linear
// in the code and the bilinear weights sum to one, so interpolating its K outputs equals applying it to the interpolated code)
int wvn_table_slots(int K) { return K > 0 && K <= 32 ? lin_kp(K) : 0; }
int wvn_table_bilerp_argmax_launch(const float* table, int* labels, int B, int G, int H, int K, hipStream_t st, int ac) {
  if (!table || !labels || B <= 0 || G <= 0 || H <= 0 || K <= 0 || K > 32 || (((uintptr_t)table) & 15)) return WVN_ERR_ARG;
  const int KP = lin_kp(K), RC = 1;
  const LinLds L = lin_lds(G, H, KP, RC);
  if (L.bytes > 160 * 1024) return WVN_ERR_ARG;
  static LdsOptIn opt;
  if (const int rc = opt(160 * 1024, (const void*)km_lin_assign_kernel<8, true>, (const void*)km_lin_assign_kernel<20, true>,
                         (const void*)km_lin_assign_kernel<32, true>)) return rc;
  const dim3 grid(G * B), block(LIN_THREADS);
  if (KP == 8) hipLaunchKernelGGL((km_lin_assign_kernel<8, true>), grid, block, L.bytes, st, table, (const float*)nullptr, (float*)nullptr, (int*)nullptr, labels, (unsigned*)nullptr, G, H, K, B, RC, ac);
  else if (KP == 20) hipLaunchKernelGGL((km_lin_assign_kernel<20, true>), grid, block, L.bytes, st, table, (const float*)nullptr, (float*)nullptr, (int*)nullptr, labels, (unsigned*)nullptr, G, H, K, B, RC, ac);
  else hipLaunchKernelGGL((km_lin_assign_kernel<32, true>), grid, block, L.bytes, st, table, (const float*)nullptr, (float*)nullptr, (int*)nullptr, labels, (unsigned*)nullptr, G, H, K, B, RC, ac);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_kmeans_pixels_linear_launch(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int C, int K,
                                    int iters, int relabel, hipStream_t st, int ac) {
  if (!code || !labels || !nseg || !scratch || B <= 0 || iters < 0 || !wvn_kmeans_pixels_linear_supported(G, H, C, K)) return WVN_ERR_ARG;
  const int KP = lin_kp(K);
#define WVN_LIN_RUN(KP_, C_) return run_linear<KP_, C_>(code, labels, nseg, scratch, B, G, H, K, iters, relabel, st, ac)
  if (C == 90) { if (KP == 8) WVN_LIN_RUN(8, 90); if (KP == 20) WVN_LIN_RUN(20, 90); WVN_LIN_RUN(32, 90); }
  if (KP == 8) WVN_LIN_RUN(8, 16);
  if (KP == 20) WVN_LIN_RUN(20, 16);
  WVN_LIN_RUN(32, 16);
#undef WVN_LIN_RUN
}
