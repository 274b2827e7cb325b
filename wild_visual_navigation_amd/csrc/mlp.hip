// Online traversability MLP: loss, gradient seed, bias reductions and Adam (fp32 throughout).
// GEMMs of the forward / backward pass use gemm_f32.hip; this file holds the row-wise and
// reduction kernels.  Reference arithmetic:
//   wild_visual_navigation/utils/loss.py:93-160, confidence_generator.py:78-82,182-193,
//   traversability_estimator.py:100,475-477 (torch.optim.Adam defaults).
// All cross-row reductions have a fixed order (single-workgroup trees, fp64 accumulators for the
// confidence statistic) so that a 1-GPU and an N-GPU run see bit-identical local contributions.
#include "common.h"
#include "mlp_device.h"
#include "wvn_internal.h"

namespace {

// loss_reco[r] = mean_d (out[r][1+d] - x[r][d])^2        one wave per row
__global__ __launch_bounds__(256) void mlp_rowloss_kernel(const float* __restrict__ out, int ldo,
                                                          const float* __restrict__ x, int ldx,
                                                          float* __restrict__ lr, int R, int D,
                                                          const int* __restrict__ rows_dev) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  if (rows_dev && row >= *rows_dev) {   // a row past the device-side row count (compacted batch): contributes nothing
    if (lane == 0) lr[row] = 0.f;
    return;
  }
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    float e = out[(size_t)row * ldo + 1 + d] - x[(size_t)row * ldx + d];
    s = fmaf(e, e, s);
  }
  s = wave_sum(s);
  if (lane == 0) lr[row] = s / (float)D;
}

// stats = { n_valid, sum(lr[valid]), sum(lr[valid]^2), R }  in fp64, single workgroup, fixed order
__global__ __launch_bounds__(1024) void mlp_stats_kernel(const float* __restrict__ lr,
                                                         const unsigned char* __restrict__ valid, int R,
                                                         double* __restrict__ stats, const int* __restrict__ rows_dev) {
  __shared__ double sh[3][16];
  if (rows_dev) R = min(R, *rows_dev);
  double n = 0, s1 = 0, s2 = 0;
  for (int r = threadIdx.x; r < R; r += 1024)
    if (valid[r]) { double v = lr[r]; n += 1.0; s1 += v; s2 += v * v; }
  n = wave_sum_d(n); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = n; sh[1][w] = s1; sh[2][w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < 16; ++i) { a += sh[0][i]; b += sh[1][i]; c += sh[2][i]; }
    stats[0] = a; stats[1] = b; stats[2] = c; stats[3] = (double)R;
  }
}

// gradient seed wrt the pre-sigmoid / linear outputs + per-row loss terms.  One wave per row.
__global__ __launch_bounds__(256) void mlp_gradout_kernel(const float* __restrict__ out, int ldo,
                                                          const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ y,
                                                          const unsigned char* __restrict__ valid,
                                                          const float* __restrict__ lr,
                                                          const double* __restrict__ stats, float std_factor,
                                                          float w_trav, float w_reco, float* __restrict__ g, int ldg,
                                                          float* __restrict__ trav_w, float* __restrict__ trav_raw,
                                                          float* __restrict__ conf_out, int R, int D,
                                                          const int* __restrict__ rows_dev) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  if (rows_dev && row >= *rows_dev) {   // absent row: zero gradient seed, zero loss terms (the GEMMs still walk it)
    if (lane == 0) { trav_raw[row] = 0.f; trav_w[row] = 0.f; if (conf_out) conf_out[row] = 0.f; }
    for (int d = lane; d < D + 1; d += 64) g[(size_t)row * ldg + d] = 0.f;
    return;
  }
  const ConfStats cs = conf_stats(stats);
  const float Rtot = (float)stats[3], nv = (float)stats[0];
  const bool v = valid[row] != 0;
  const float conf = confidence_of(lr[row], cs.mean, cs.std, std_factor);
  const float s = out[(size_t)row * ldo];
  const float diff = s - y[row];
  const float raw = diff * diff;
  const float wrow = v ? 1.f : (1.f - conf);
  if (lane == 0) {
    trav_raw[row] = raw;
    trav_w[row] = raw * wrow;
    if (conf_out) conf_out[row] = conf;
    g[(size_t)row * ldg] = (w_trav / Rtot) * wrow * 2.f * diff * s * (1.f - s);
  }
  const float cr = v ? (w_reco / (nv * (float)D)) * 2.f : 0.f;
  for (int d = lane; d < D; d += 64)
    g[(size_t)row * ldg + 1 + d] = cr * (out[(size_t)row * ldo + 1 + d] - x[(size_t)row * ldx + d]);
}

// extra[0] = sum(trav_w), extra[1] = sum(trav_raw)   (fp64 tree, single workgroup) -> fp32
__global__ __launch_bounds__(1024) void mlp_losssum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           int R, float* __restrict__ extra) {
  __shared__ double sh[2][16];
  double sa = 0, sb = 0;
  for (int r = threadIdx.x; r < R; r += 1024) { sa += a[r]; sb += b[r]; }
  sa = wave_sum_d(sa); sb = wave_sum_d(sb);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = sa; sh[1][w] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double p = 0, q = 0;
    for (int i = 0; i < 16; ++i) { p += sh[0][i]; q += sh[1][i]; }
    extra[0] = (float)p; extra[1] = (float)q;
  }
}

// out[n] = sum_r A[r][n]  : 64 columns per workgroup, 16 row phases (1024 threads: the sum is latency-bound, R/16 loads
// per thread, 4 in flight), fixed-order combine
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ A, int lda, int R, int N,
                                                      float* __restrict__ outv) {
  __shared__ float sh[16][64];
  const int lane = threadIdx.x & 63, c = blockIdx.x * 64 + lane, g = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < N) {
    int r = g;
    for (; r + 48 < R; r += 64) {
      s0 += A[(size_t)r * lda + c];
      s1 += A[(size_t)(r + 16) * lda + c];
      s2 += A[(size_t)(r + 32) * lda + c];
      s3 += A[(size_t)(r + 48) * lda + c];
    }
    for (; r < R; r += 16) s0 += A[(size_t)r * lda + c];
  }
  sh[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < N) {
    float t = sh[0][lane];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += sh[i][lane];
    outv[c] = t;
  }
}

// torch.optim.Adam single-tensor update (no amsgrad / weight decay / maximize)
__device__ inline void write_losses(const double* __restrict__ stats, const float* __restrict__ extra, float w_trav, float w_reco,
                                    float* __restrict__ losses) {
  const ConfStats cs = conf_stats(stats);
  const float Rtot = (float)stats[3];
  const float reco = (float)(stats[1] / stats[0]);
  const float trav_conf = extra[0] / Rtot;
  losses[0] = w_trav * trav_conf + w_reco * reco;
  losses[1] = extra[1] / Rtot;
  losses[2] = reco;
  losses[3] = cs.mean;
  losses[4] = cs.std;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, const double* __restrict__ stats, const float* __restrict__ extra, float w_trav,
                            float w_reco, float* __restrict__ losses) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (losses && i == 0) write_losses(stats, extra, w_trav, w_reco, losses);   // (the step's losses ride along: one launch fewer)
  if (i >= n) return;
  const float gi = g[i];
  const float mi = m[i] * b1 + (1.f - b1) * gi;
  const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}

// losses[0] = total, [1] = loss_trav (raw mean), [2] = loss_reco (mean over labelled), [3] = mean, [4] = std
__global__ void mlp_losses_kernel(const double* __restrict__ stats, const float* __restrict__ extra, float w_trav,
                                  float w_reco, float* __restrict__ losses) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  write_losses(stats, extra, w_trav, w_reco, losses);
}

// per-row reconstruction confidence for inference (quick_start.py:207-210, loss.py:162-164)
__global__ __launch_bounds__(256) void mlp_confidence_kernel(const float* __restrict__ out, int ldo,
                                                             const float* __restrict__ x, int ldx, float mean,
                                                             float std, float std_factor, float* __restrict__ trav,
                                                             float* __restrict__ conf, int R, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    float e = out[(size_t)row * ldo + 1 + d] - x[(size_t)row * ldx + d];
    s = fmaf(e, e, s);
  }
  s = wave_sum(s) / (float)D;
  if (lane == 0) {
    if (trav) trav[row] = out[(size_t)row * ldo];
    if (conf) conf[row] = confidence_of(s, mean, std, std_factor);
  }
}

// Rows of the segments that exist -- segment s of frame b exists iff s < nseg[b] -- packed front to back in (b, s) order, the
// row count left in device memory.  Replaces the boolean-mask selection feat[keep] (a partition + a host synchronisation to
// learn the count) in front of the training step: the step then runs on B*S rows of which the first *count are real.
// Rows past the count are zeroed (finite inputs for the GEMMs that still walk them).  One workgroup per frame.
__global__ __launch_bounds__(256) void compact_segment_rows_kernel(const float* __restrict__ feat, int D, const float* __restrict__ side,
                                                                   int Ds, const int* __restrict__ nseg, int B, int S,
                                                                   float* __restrict__ x, float* __restrict__ side_out,
                                                                   int* __restrict__ count) {
  const int b = blockIdx.x;
  int off = 0, total = 0;
  for (int i = 0; i < B; ++i) {   // (uniform, B <= a few hundred)
    const int n = min(max(nseg[i], 0), S);
    if (i < b) off += n;
    total += n;
  }
  const int n = min(max(nseg[b], 0), S);
  for (int i = threadIdx.x; i < n * D; i += blockDim.x) x[(size_t)off * D + i] = feat[(size_t)b * S * D + i];
  if (side)
    for (int i = threadIdx.x; i < n * Ds; i += blockDim.x) side_out[(size_t)off * Ds + i] = side[(size_t)b * S * Ds + i];
  // the tail [total, B*S): frame b clears its share
  const int tail = B * S - total, t0 = (int)((long long)tail * b / B), t1 = (int)((long long)tail * (b + 1) / B);
  for (int i = threadIdx.x; i < (t1 - t0) * D; i += blockDim.x) x[(size_t)(total + t0) * D + i] = 0.f;
  if (side)
    for (int i = threadIdx.x; i < (t1 - t0) * Ds; i += blockDim.x) side_out[(size_t)(total + t0) * Ds + i] = 0.f;
  if (b == 0 && threadIdx.x == 0) *count = total;
}

}  // namespace

int wvn_compact_segment_rows_launch(const float* feat, int D, const float* side, int Ds, const int* nseg, int B, int S, float* x,
                                    float* side_out, int* count, hipStream_t st) {
  if (!feat || !nseg || !x || !count || B <= 0 || S <= 0 || D <= 0 || (side && (!side_out || Ds <= 0))) return WVN_ERR_ARG;
  hipLaunchKernelGGL(compact_segment_rows_kernel, dim3(B), dim3(256), 0, st, feat, D, side, Ds, nseg, B, S, x, side_out, count);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_mlp_rowloss_stats_launch(const float* out, int ldo, const float* x, int ldx, const unsigned char* valid,
                                 float* lr, double* stats, int R, int D, hipStream_t st, const int* rows_dev) {
  hipLaunchKernelGGL(mlp_rowloss_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, st, out, ldo, x, ldx, lr, R, D, rows_dev);
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp_stats_kernel, dim3(1), dim3(1024), 0, st, lr, valid, R, stats, rows_dev);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_mlp_gradout_launch(const float* out, int ldo, const float* x, int ldx, const float* y,
                           const unsigned char* valid, const float* lr, const double* stats, float std_factor,
                           float w_trav, float w_reco, float* g, int ldg, float* trav_w, float* trav_raw,
                           float* conf_out, float* extra, int R, int D, hipStream_t st, const int* rows_dev) {
  hipLaunchKernelGGL(mlp_gradout_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, st, out, ldo, x, ldx, y, valid, lr, stats,
                     std_factor, w_trav, w_reco, g, ldg, trav_w, trav_raw, conf_out, R, D, rows_dev);
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(mlp_losssum_kernel, dim3(1), dim3(1024), 0, st, trav_w, trav_raw, R, extra);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_colsum_launch(const float* A, int lda, int R, int N, float* outv, hipStream_t st) {
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64)), dim3(1024), 0, st, A, lda, R, N, outv);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_adam_launch(float* p, const float* g, float* m, float* v, int n, int step, float lr, float b1, float b2,
                    float eps, hipStream_t st, const double* stats, const float* extra, float w_trav, float w_reco, float* losses) {
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)b2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2s, stats, extra,
                     w_trav, w_reco, losses);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_mlp_losses_launch(const double* stats, const float* extra, float w_trav, float w_reco, float* losses,
                          hipStream_t st) {
  hipLaunchKernelGGL(mlp_losses_kernel, dim3(1), dim3(64), 0, st, stats, extra, w_trav, w_reco, losses);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_mlp_confidence_launch(const float* out, int ldo, const float* x, int ldx, float mean, float std,
                              float std_factor, float* trav, float* conf, int R, int D, hipStream_t st) {
  hipLaunchKernelGGL(mlp_confidence_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, st, out, ldo, x, ldx, mean, std,
                     std_factor, trav, conf, R, D);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
