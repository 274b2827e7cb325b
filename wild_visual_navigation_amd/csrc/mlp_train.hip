// One optimisation step of the online traversability MLP (SimpleMLP D -> 256 -> 32 -> 1 + D) in FOUR launches, for the batch sizes
// the learning node actually trains on (8 mission nodes x <= 100 segments = a few hundred rows, traversability_estimator.py:
// 432-446; bench.py: 1280 rows):
//     fwd   : x -> h1 -> h2 -> out (sigmoid on column 0), per-row reconstruction loss, LOCAL statistic {n_lab, sum, sum^2, R}
//             -- phase A of trainer.py; the multi-GPU statistics all-reduce sits behind it
//     bwd   : loss gradient seed, dL/dh2, dL/dh1 (ReLU masks), per-row loss terms               -- phase B, data path
//     wgrad : dW3, dW2, dW1 and the three bias gradients; the loss sums join the flat gradient    -- phase B, reductions
//             (the gradient all-reduce sits behind it)
//     adam  : Adam update + the step's losses                                                     -- phase C (mlp.hip)
// The general path (mlp.hip + gemm_f32.hip) spends 17-20 launches on the same step: at these sizes every one of them is a
// launch latency.  Same arithmetic (fp32 FMA, loss.py:93-160, confidence_generator.py:78-82,182-193), other summation orders:
// every reduction here has ONE fixed order (row tiles in ascending order, rows in ascending order inside a tile, fp64 for the
// statistics), so runs are bit-reproducible and replicas of a data-parallel job stay identical.
// A workgroup owns 32 consecutive rows through all layers; weights stream from L2 (478 KB for D = 384), activations of the
// tile live in LDS.
#include "common.h"
#include "mlp_device.h"
#include "wvn_internal.h"

namespace {

constexpr int TR = 32;     // rows per workgroup
constexpr int H1 = 256, H2 = 32;
constexpr int H1P = H1 + 4;  // LDS pitch of the h1 tile (bank spread for the 8 rows a wave reads at once)

struct TrainParams {
  const float* P;                 // flat parameters: W1 [256][D], b1, W2 [32][256], b2, W3 [1+D][32], b3
  size_t oW1, ob1, oW2, ob2, oW3, ob3, ntotal;
  const float* x; int ldx;        // [R][D]
  const float* y;                 // [R]
  const unsigned char* valid;     // [R]
  const int* rows_dev;            // optional: only the first *rows_dev rows are real
  int R, D;
  float *h1, *h2, *out, *lr;      // [R][256], [R][32], [R][1+D], [R]
  float *g_out, *g_h2, *g_h1;     // [R][1+D], [R][32], [R][256]
  double* part;                   // [ntiles][4] per-tile partials (fwd: n, s1, s2; bwd: sum trav_w, sum trav_raw)
  double* stats;                  // [4] out (fwd) / in (bwd)
  unsigned* ticket;               // arrival counter of the fwd launch (zero on entry, reset by the last arriver)
  float std_factor, w_trav, w_reco;
  float* conf_out;                // optional [R]
  float* grads;                   // [ntotal + 2]
};

__device__ inline int real_rows(const TrainParams& p) { return p.rows_dev ? min(p.R, *p.rows_dev) : p.R; }

// fixed-order sum over the 32 lanes of a half-wave (lanes 0..31 hold the values): butterfly, every lane gets the total
__device__ inline double sum32_d(double v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// fwd
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_train_fwd_kernel(TrainParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, O = D + 1, DP = (D + 3) / 4 * 4 + 4;     // x tile pitch: 16-byte rows
  float* xs = sm;                       // [TR][DP]
  float* h1s = xs + TR * DP;            // [TR][H1P]
  float* h2s = h1s + TR * H1P;          // [TR][H2]
  float* outs = h2s + TR * H2;          // [TR][O]
  __shared__ int last_flag;
  const int tid = threadIdx.x, row0 = blockIdx.x * TR;
  const int Rr = real_rows(p);
  // ---- x tile (rows past R / past the device-side count: zeros) ----
  for (int i = tid; i < TR * D; i += 256) {
    const int r = i / D, k = i - r * D;
    xs[r * DP + k] = (row0 + r < Rr) ? p.x[(size_t)(row0 + r) * p.ldx + k] : 0.f;
  }
  __syncthreads();
  // ---- layer 1: h1 = relu(x W1^T + b1).  thread -> columns n0, n0 + 1, rows 16 rh .. + 15 (a wave shares rh: x reads broadcast) ----
  {
    const int n0 = (tid & 127) * 2, rh = tid >> 7;
    const float* w0 = p.P + p.oW1 + (size_t)n0 * D;
    const float* w1 = w0 + D;
    float a0[16], a1[16];
    const float b0 = p.P[p.ob1 + n0], b1 = p.P[p.ob1 + n0 + 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = b0; a1[r] = b1; }
    const float* xr = xs + (16 * rh) * DP;
    for (int k = 0; k + 1 < D; k += 2) {
      const float wa0 = w0[k], wa1 = w0[k + 1], wb0 = w1[k], wb1 = w1[k + 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x0 = xr[r * DP + k], x1 = xr[r * DP + k + 1];
        a0[r] = fmaf(x1, wa1, fmaf(x0, wa0, a0[r]));
        a1[r] = fmaf(x1, wb1, fmaf(x0, wb0, a1[r]));
      }
    }
    if (D & 1) {
      const int k = D - 1;
      const float wa0 = w0[k], wb0 = w1[k];
#pragma unroll
      for (int r = 0; r < 16; ++r) { a0[r] = fmaf(xr[r * DP + k], wa0, a0[r]); a1[r] = fmaf(xr[r * DP + k], wb0, a1[r]); }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v0 = fmaxf(a0[r], 0.f), v1 = fmaxf(a1[r], 0.f);
      h1s[(16 * rh + r) * H1P + n0] = v0;
      h1s[(16 * rh + r) * H1P + n0 + 1] = v1;
      if (row0 + 16 * rh + r < p.R) *(float2*)(p.h1 + (size_t)(row0 + 16 * rh + r) * H1 + n0) = float2{v0, v1};
    }
  }
  __syncthreads();
  // ---- layer 2: h2 = relu(h1 W2^T + b2).  thread -> column j, rows 4 rg .. + 3 ----
  {
    const int j = tid & 31, rg = tid >> 5;
    const float* w = p.P + p.oW2 + (size_t)j * H1;
    float a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = p.P[p.ob2 + j];
    for (int k = 0; k < H1; k += 4) {
      const f32x4_t w4 = *(const f32x4_t*)(w + k);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4_t h4 = *(const f32x4_t*)(h1s + (4 * rg + r) * H1P + k);
        a[r] = fmaf(h4[3], w4[3], fmaf(h4[2], w4[2], fmaf(h4[1], w4[1], fmaf(h4[0], w4[0], a[r]))));
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = fmaxf(a[r], 0.f);
      h2s[(4 * rg + r) * H2 + j] = v;
      if (row0 + 4 * rg + r < p.R) p.h2[(size_t)(row0 + 4 * rg + r) * H2 + j] = v;
    }
  }
  __syncthreads();
  // ---- layer 3: out = h2 W3^T + b3, sigmoid on column 0.  thread -> columns n = tid, tid + 256 ----
  for (int n = tid; n < O; n += 256) {
    float w[H2];
#pragma unroll
    for (int j = 0; j < H2; j += 4) {
      const f32x4_t w4 = *(const f32x4_t*)(p.P + p.oW3 + (size_t)n * H2 + j);
      w[j] = w4[0]; w[j + 1] = w4[1]; w[j + 2] = w4[2]; w[j + 3] = w4[3];
    }
    const float b = p.P[p.ob3 + n];
    for (int r = 0; r < TR; ++r) {
      float a = b;
#pragma unroll
      for (int j = 0; j < H2; ++j) a = fmaf(h2s[r * H2 + j], w[j], a);
      if (n == 0) a = sigmoid_f(a);
      outs[r * O + n] = a;
      if (row0 + r < p.R) p.out[(size_t)(row0 + r) * O + n] = a;
    }
  }
  __syncthreads();
  // ---- per-row reconstruction loss: 8 threads per row, fixed order ----
  {
    const int r = tid >> 3, q = tid & 7;
    float s = 0.f;
    for (int d = q; d < D; d += 8) {
      const float e = outs[r * O + 1 + d] - xs[r * DP + d];
      s = fmaf(e, e, s);
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float lr = (row0 + r < Rr) ? s / (float)D : 0.f;
    if (q == 0 && row0 + r < p.R) p.lr[row0 + r] = lr;
    // ---- tile partial of the confidence statistic (fp64): rows in ascending order through a butterfly over the 32 rows ----
    if (q == 0) h2s[r] = lr;     // (h2s is dead: reuse as the tile's lr vector)
  }
  __syncthreads();
  if (tid < 32) {
    const bool v = row0 + tid < Rr && p.valid[row0 + tid] != 0;
    const double l = v ? (double)h2s[tid] : 0.0;
    const double n = sum32_d(v ? 1.0 : 0.0), s1 = sum32_d(l), s2 = sum32_d(l * l);
    if (tid == 0) {
      double* d = p.part + (size_t)blockIdx.x * 4;
      d[0] = n; d[1] = s1; d[2] = s2;
      // publish: the partial must be visible device-wide before the ticket (MI355X_MICROARCH.md, producer form)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_flag = t == gridDim.x - 1;
    }
  }
  __syncthreads();
  if (last_flag) {   // the last tile to arrive folds the partials in ascending tile order
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      double a = 0, b = 0, c = 0;
      for (unsigned i = 0; i < gridDim.x; ++i) {
        const double* d = p.part + (size_t)i * 4;
        a += __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b += __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c += __hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      p.stats[0] = a; p.stats[1] = b; p.stats[2] = c; p.stats[3] = (double)Rr;
      __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next step
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// bwd (data path)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_train_bwd_kernel(TrainParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = p.D, O = D + 1;
  float* gos = sm;                     // [TR][O]   gradient seed of the tile
  float* gh2s = gos + TR * O;          // [TR][H2]
  float* tw = gh2s + TR * H2;          // [TR] trav_w, [TR] trav_raw
  const int tid = threadIdx.x, row0 = blockIdx.x * TR;
  const int Rr = real_rows(p);
  const ConfStats cs = conf_stats(p.stats);
  const float Rtot = (float)p.stats[3], nv = (float)p.stats[0];
  // ---- gradient seed (loss.py:125-147): 8 threads per row ----
  {
    const int r = tid >> 3, q = tid & 7, row = row0 + r;
    const bool real = row < Rr;
    const bool v = real && p.valid[row] != 0;
    float diff = 0.f, s = 0.f, wrow = 0.f, conf = 0.f;
    if (real) {
      conf = confidence_of(p.lr[row], cs.mean, cs.std, p.std_factor);
      s = p.out[(size_t)row * O];
      diff = s - p.y[row];
      wrow = v ? 1.f : (1.f - conf);
    }
    if (q == 0) {
      const float raw = diff * diff;
      tw[r] = real ? raw * wrow : 0.f;
      tw[TR + r] = real ? raw : 0.f;
      if (p.conf_out && row < p.R) p.conf_out[row] = conf;
      const float g0 = real ? (p.w_trav / Rtot) * wrow * 2.f * diff * s * (1.f - s) : 0.f;
      gos[r * O] = g0;
      if (row < p.R) p.g_out[(size_t)row * O] = g0;
    }
    const float cr = v ? (p.w_reco / (nv * (float)D)) * 2.f : 0.f;
    for (int d = q; d < D; d += 8) {
      const float g = real ? cr * (p.out[(size_t)row * O + 1 + d] - p.x[(size_t)row * p.ldx + d]) : 0.f;
      gos[r * O + 1 + d] = g;
      if (row < p.R) p.g_out[(size_t)row * O + 1 + d] = g;
    }
  }
  __syncthreads();
  // ---- tile partial of the loss sums (fp64, rows in ascending order through a butterfly) ----
  if (tid < 32) {
    const double a = sum32_d((double)tw[tid]), b = sum32_d((double)tw[TR + tid]);
    if (tid == 0) { p.part[(size_t)blockIdx.x * 4] = a; p.part[(size_t)blockIdx.x * 4 + 1] = b; }
  }
  // ---- g_h2 = (g_out W3) masked by h2 > 0.  thread -> row r, columns j0 .. j0 + 3; W3 [O][32] ----
  {
    const int r = tid >> 3, j0 = (tid & 7) * 4, row = row0 + r;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < O; ++n) {
      const f32x4_t w4 = *(const f32x4_t*)(p.P + p.oW3 + (size_t)n * H2 + j0);
      const float g = gos[r * O + n];
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = fmaf(g, w4[e], a[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = row < p.R ? p.h2[(size_t)row * H2 + j0 + e] : 0.f;
      const float v = m > 0.f ? a[e] : 0.f;
      gh2s[r * H2 + j0 + e] = v;
      if (row < p.R) p.g_h2[(size_t)row * H2 + j0 + e] = v;
    }
  }
  __syncthreads();
  // ---- g_h1 = (g_h2 W2) masked by h1 > 0.  thread -> column i = tid, all 32 rows; W2 [32][256] ----
  {
    float w[H2];
#pragma unroll
    for (int j = 0; j < H2; ++j) w[j] = p.P[p.oW2 + (size_t)j * H1 + tid];
    for (int r = 0; r < TR; ++r) {
      if (row0 + r >= p.R) break;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < H2; ++j) a = fmaf(gh2s[r * H2 + j], w[j], a);
      const float m = p.h1[(size_t)(row0 + r) * H1 + tid];
      p.g_h1[(size_t)(row0 + r) * H1 + tid] = m > 0.f ? a : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// wgrad: dW[m][n] = sum_r G[r][m] Hm[r][n], 32 x 32 output tiles, rows in ascending order (no split-K: one fixed order);
// the first column tile of every matrix also forms the bias gradient sum_r G[r][m]; workgroup 0 folds the loss sums
// ---------------------------------------------------------------------------------------------------------------------------
struct WTile { const float* G; int ldg; const float* Hm; int ldh; int M, N; float* dW; float* db; };
__global__ __launch_bounds__(256) void mlp_train_wgrad_kernel(TrainParams p, int t3, int t2, int ntiles_rows) {
  __shared__ float Gs[32][33], Hs[32][33];
  const int O = p.D + 1, D = p.D;
  int b = blockIdx.x;
  WTile w;
  if (b < t3) { w = WTile{p.g_out, O, p.h2, H2, O, H2, p.grads + p.oW3, p.grads + p.ob3}; }
  else if (b < t3 + t2) { b -= t3; w = WTile{p.g_h2, H2, p.h1, H1, H2, H1, p.grads + p.oW2, p.grads + p.ob2}; }
  else { b -= t3 + t2; w = WTile{p.g_h1, H1, p.x, p.ldx, H1, D, p.grads + p.oW1, p.grads + p.ob1}; }
  const int ntn = (w.N + 31) / 32, tm = b / ntn, tn = b - tm * ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const int tid = threadIdx.x, lr_ = tid >> 5, lc = tid & 31;   // loader: row lr_ + 8 i, column lc
  const int om = tid >> 4, on = (tid & 15) * 2;                 // outputs: rows om, om + 16; columns on, on + 1
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float bsum = 0.f;                                             // tn == 0, tid < 32: sum_r G[r][m0 + tid]
  const int Rr = real_rows(p);
  for (int r0 = 0; r0 < Rr; r0 += 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + lr_ + 8 * i;
      Gs[lr_ + 8 * i][lc] = (r < Rr && m0 + lc < w.M) ? w.G[(size_t)r * w.ldg + m0 + lc] : 0.f;
      Hs[lr_ + 8 * i][lc] = (r < Rr && n0 + lc < w.N) ? w.Hm[(size_t)r * w.ldh + n0 + lc] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const float g0 = Gs[r][om], g1 = Gs[r][om + 16], h0 = Hs[r][on], h1 = Hs[r][on + 1];
      acc[0][0] = fmaf(g0, h0, acc[0][0]); acc[0][1] = fmaf(g0, h1, acc[0][1]);
      acc[1][0] = fmaf(g1, h0, acc[1][0]); acc[1][1] = fmaf(g1, h1, acc[1][1]);
    }
    if (tn == 0 && tid < 32)
      for (int r = 0; r < 32; ++r) bsum += Gs[r][tid];
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int m = m0 + om + 16 * i, n = n0 + on + jn;
      if (m < w.M && n < w.N) w.dW[(size_t)m * w.N + n] = acc[i][jn];
    }
  if (tn == 0 && tid < 32 && m0 + tid < w.M) w.db[m0 + tid] = bsum;
  if (blockIdx.x == 0 && tid == 0) {   // the loss sums of the bwd launch, tiles in ascending order
    double a = 0, c = 0;
    for (int i = 0; i < ntiles_rows; ++i) { a += p.part[(size_t)i * 4]; c += p.part[(size_t)i * 4 + 1]; }
    p.grads[p.ntotal] = (float)a;
    p.grads[p.ntotal + 1] = (float)c;
  }
}

constexpr size_t FUSED_LDS_MAX = 156 * 1024;
size_t fwd_lds(int D) { return (size_t)(TR * ((D + 3) / 4 * 4 + 4) + TR * H1P + TR * H2 + TR * (D + 1)) * sizeof(float); }
size_t bwd_lds(int D) { return (size_t)(TR * (D + 1) + TR * H2 + 2 * TR) * sizeof(float); }

}  // namespace

// Eligibility of the four-launch step: the SimpleMLP geometry (256, 32), any D whose row tiles fit the LDS (<= 460), at most 2048 rows (the wgrad
// launch walks the rows without split-K).  The scratch behind the MLP workspace: part [ceil(R / 32)][4] doubles + one ticket.
bool wvn_mlp_train_fused_ok(int D, int H1_, int H2_, int R) {
  // measured (MI355X, D = 384 / 90): 75 vs 152 us per step at 160 rows, 196 vs 312 at 800, 185 vs 251 at 1280; the un-split weight
  // gradients lose from about 3000 rows on (792 vs 504 us at 6400), where the general path's split-K GEMMs take over
  return H1_ == H1 && H2_ == H2 && D > 0 && fwd_lds(D) <= FUSED_LDS_MAX && R > 0 && R <= 2048;   // D <= 460
}
size_t wvn_mlp_train_fused_scratch_bytes(int R) { return (size_t)ceil_div(R, TR) * 4 * sizeof(double); }

static TrainParams make_params(const float* P, const size_t* off, size_t ntotal, const float* x, int ldx, int R, int D, const int* rows_dev,
                               float* h1, float* h2, float* out, float* lr, void* scratch, unsigned* sync_word) {
  TrainParams p{};
  p.P = P; p.oW1 = off[0]; p.ob1 = off[1]; p.oW2 = off[2]; p.ob2 = off[3]; p.oW3 = off[4]; p.ob3 = off[5]; p.ntotal = ntotal;
  p.x = x; p.ldx = ldx; p.R = R; p.D = D; p.rows_dev = rows_dev;
  p.h1 = h1; p.h2 = h2; p.out = out; p.lr = lr;
  p.part = (double*)scratch;
  p.ticket = sync_word;
  return p;
}

int wvn_mlp_train_fwd_launch(const float* P, const size_t* off, size_t ntotal, const float* x, int ldx, const unsigned char* valid, int R,
                             int D, const int* rows_dev, float* h1, float* h2, float* out, float* lr, double* stats, void* scratch,
                             unsigned* sync_word, hipStream_t st) {
  TrainParams p = make_params(P, off, ntotal, x, ldx, R, D, rows_dev, h1, h2, out, lr, scratch, sync_word);
  p.valid = valid; p.stats = stats;
  if (fwd_lds(D) > FUSED_LDS_MAX) return WVN_ERR_ARG;
  static LdsOptIn lds_opt_in;   // (the kernels also hold a few bytes of static LDS: the dynamic limit must stay below 160 KB)
  if (const int rc = lds_opt_in((int)FUSED_LDS_MAX, (const void*)mlp_train_fwd_kernel, (const void*)mlp_train_bwd_kernel)) return rc;
  // the arrival counter is zeroed on the stream in front of every launch (a count left over by an aborted launch, or a
  // sync word the caller shares, would leave no tile "last" and the step would train on stale statistics without an error)
  if (const hipError_t e = hipMemsetAsync(sync_word, 0, sizeof(unsigned), st); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(mlp_train_fwd_kernel, dim3(ceil_div(R, TR)), dim3(256), fwd_lds(D), st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_mlp_train_bwd_launch(const float* P, const size_t* off, size_t ntotal, const float* x, int ldx, const float* y,
                             const unsigned char* valid, int R, int D, const int* rows_dev, float* h1, float* h2, float* out, float* lr,
                             float* g_out, float* g_h2, float* g_h1, const double* stats, float std_factor, float w_trav, float w_reco,
                             float* conf_out, float* grads, void* scratch, hipStream_t st) {
  TrainParams p = make_params(P, off, ntotal, x, ldx, R, D, rows_dev, h1, h2, out, lr, scratch, nullptr);
  p.y = y; p.valid = valid; p.stats = (double*)stats; p.g_out = g_out; p.g_h2 = g_h2; p.g_h1 = g_h1;
  p.std_factor = std_factor; p.w_trav = w_trav; p.w_reco = w_reco; p.conf_out = conf_out; p.grads = grads;
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in((int)FUSED_LDS_MAX, (const void*)mlp_train_fwd_kernel, (const void*)mlp_train_bwd_kernel)) return rc;
  const int ntiles = ceil_div(R, TR);
  hipLaunchKernelGGL(mlp_train_bwd_kernel, dim3(ntiles), dim3(256), bwd_lds(D), st, p);
  WVN_LAUNCH_CHECK();
  const int O = D + 1;
  const int t3 = ceil_div(O, 32) * ceil_div(H2, 32), t2 = ceil_div(H2, 32) * ceil_div(H1, 32), t1 = ceil_div(H1, 32) * ceil_div(D, 32);
  hipLaunchKernelGGL(mlp_train_wgrad_kernel, dim3(t3 + t2 + t1), dim3(256), 0, st, p, t3, t2, ntiles);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
