/* Oracle (TEST INFRASTRUCTURE ONLY): oracle/kmeans_linear.py -- the pixel-resolution cosine k-means stated through its linearity --
 * in plain C, operation for operation (fmaf = the correctly rounded fused multiply-add; every other multiply / add is its own fp32
 * rounding: -ffp-contract=off), so that 448 x 448 frames take fractions of a second.  tests/test_oracle_stego.py pins this file
 * against the numpy statement.  Algorithm and summation orders: the header of oracle/kmeans_linear.py (stego_interface.py:94-109 as this
 * build reads it; PARITY UNPINNED, the `stego` package is absent).
 * Built into oracle/_build/libwvn_oracle.so together with kmeans_ref.c (oracle/build_oracle.py). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ROW_GROUP 2

static float lin_rinv_norm(float n2) {
  float n = sqrtf(n2);
  if (n < 1e-12f) n = 1e-12f;
  return 1.0f / n;
}

typedef struct { int* i0; int* i1; float* w0; float* w1; } taps_t;

static float bilerp_fixed(float v00, float v01, float v10, float v11, float wx0, float wx1, float wy0, float wy1) {
  const float t0 = fmaf(wx1, v01, wx0 * v00);
  const float t1 = fmaf(wx1, v11, wx0 * v10);
  return fmaf(wy1, t1, wy0 * t0);
}

__attribute__((target_clones("fma", "default")))
static void assign(const float* code, const float* cent, const taps_t* t, int G, int H, int C, int K, float* S, int* lab) {
  const long T = (long)G * G;
#pragma omp parallel for schedule(static)
  for (long p = 0; p < T; ++p)
    for (int k = 0; k < K; ++k) {
      float acc = 0.f;
      for (int d = 0; d < C; ++d) acc = fmaf(code[p * C + d], cent[(long)k * C + d], acc);
      S[p * K + k] = acc;
    }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < H; ++x) {
      const float* a0 = S + ((long)t->i0[y] * G + t->i0[x]) * K;
      const float* a1 = S + ((long)t->i0[y] * G + t->i1[x]) * K;
      const float* b0 = S + ((long)t->i1[y] * G + t->i0[x]) * K;
      const float* b1 = S + ((long)t->i1[y] * G + t->i1[x]) * K;
      int best = 0;
      float bv = -INFINITY;
      for (int k = 0; k < K; ++k) {
        const float v = bilerp_fixed(a0[k], a1[k], b0[k], b1[k], t->w0[x], t->w1[x], t->w0[y], t->w1[y]);
        if (v > bv) { bv = v; best = k; }
      }
      lab[(long)y * H + x] = best;
    }
}

/* code [G*G][C] fp32 patch codes -> labels [H*H] int32 (not compacted); when not NULL: the final centroids [K][C] and the normalised
 * up-sampled rows [H*H][C] (the points of the clustering; oracle/segmap_agreement.py).  Returns 0, or 1 when out of memory. */
__attribute__((target_clones("fma", "default")))
int wvn_oracle_kmeans_pixels_linear_ac(const float* code, int G, int H, int C, int K, int iters, int* labels, float* cent_out, float* x_out, int align_corners) {
  const long P = (long)H * H, T = (long)G * G;
  int* i0 = (int*)malloc(sizeof(int) * H); int* i1 = (int*)malloc(sizeof(int) * H);
  float* w0 = (float*)malloc(sizeof(float) * H); float* w1 = (float*)malloc(sizeof(float) * H);
  float* rinv = (float*)malloc(sizeof(float) * P);
  float* cent = (float*)malloc(sizeof(float) * K * C);
  float* S = (float*)malloc(sizeof(float) * T * K);
  float* U = (float*)malloc(sizeof(float) * (size_t)H * K * G);
  float* P0 = (float*)malloc(sizeof(float) * (size_t)G * K * G); float* P1 = (float*)malloc(sizeof(float) * (size_t)G * K * G);
  float* A = (float*)malloc(sizeof(float) * (size_t)K * G * G);
  float* R = (float*)malloc(sizeof(float) * (size_t)G * K * C);
  float* sums = (float*)malloc(sizeof(float) * K * C);
  long* cnt = (long*)malloc(sizeof(long) * K);
  float* row = (float*)malloc(sizeof(float) * C);
  if (!i0 || !i1 || !w0 || !w1 || !rinv || !cent || !S || !U || !P0 || !P1 || !A || !R || !sums || !cnt || !row) return 1;
  /* align_corners = 0: ATen's half-pixel coordinates, src = max((G / H) (o + 0.5) - 0.5, 0), every operation rounded on its own (-ffp-contract=off) */
  const float scale = align_corners ? (H > 1 ? (float)(G - 1) / (float)(H - 1) : 0.f) : (float)G / (float)H;
  for (int o = 0; o < H; ++o) {
    float s;
    if (align_corners) s = scale * (float)o;
    else {
      volatile float m = scale * ((float)o + 0.5f);
      s = m - 0.5f;
      if (s < 0.f) s = 0.f;
    }
    i0[o] = (int)s;
    i1[o] = i0[o] + (i0[o] < G - 1 ? 1 : 0);
    w1[o] = s - (float)i0[o];
    w0[o] = 1.0f - w1[o];
  }
  const taps_t t = {i0, i1, w0, w1};
  /* rinv of every pixel (and, on request, the normalised rows); the initial centroids */
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < H; ++x) {
      float n2 = 0.f;
      for (int d = 0; d < C; ++d) {
        row[d] = bilerp_fixed(code[((long)i0[y] * G + i0[x]) * C + d], code[((long)i0[y] * G + i1[x]) * C + d],
                              code[((long)i1[y] * G + i0[x]) * C + d], code[((long)i1[y] * G + i1[x]) * C + d], w0[x], w1[x], w0[y], w1[y]);
        n2 = fmaf(row[d], row[d], n2);
      }
      const float r = lin_rinv_norm(n2);
      const long p = (long)y * H + x;
      rinv[p] = r;
      if (x_out) for (int d = 0; d < C; ++d) x_out[p * C + d] = row[d] * r;
      for (int k = 0; k < K; ++k)
        if (p == ((long)(2 * k + 1) * P) / (2 * K))
          for (int d = 0; d < C; ++d) cent[(long)k * C + d] = row[d] * r;
    }
  for (int it = 0; it < iters; ++it) {
    assign(code, cent, &t, G, H, C, K, S, labels);
    memset(U, 0, sizeof(float) * (size_t)H * K * G);
    for (int k = 0; k < K; ++k) cnt[k] = 0;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < H; ++x) {   /* ascending x; tap 0 before tap 1 */
        const long p = (long)y * H + x;
        const int k = labels[p];
        float* u = U + ((long)y * K + k) * G;
        u[i0[x]] = u[i0[x]] + rinv[p] * w0[x];
        u[i1[x]] = u[i1[x]] + rinv[p] * w1[x];
        cnt[k] += 1;
      }
    memset(P0, 0, sizeof(float) * (size_t)G * K * G);
    memset(P1, 0, sizeof(float) * (size_t)G * K * G);
    for (int y = 0; y < H; ++y) {     /* ascending y inside every band */
      float* p0 = P0 + (long)i0[y] * K * G;
      float* p1 = P1 + (long)i0[y] * K * G;
      const float* u = U + (long)y * K * G;
      for (long e = 0; e < (long)K * G; ++e) { p0[e] = fmaf(w0[y], u[e], p0[e]); p1[e] = fmaf(w1[y], u[e], p1[e]); }
    }
    memset(A, 0, sizeof(float) * (size_t)K * G * G);
    for (int b = 0; b < G; ++b) {
      const int ib = b + (b < G - 1 ? 1 : 0);
      for (int k = 0; k < K; ++k)
        for (int j = 0; j < G; ++j) A[((long)k * G + b) * G + j] = A[((long)k * G + b) * G + j] + P0[((long)b * K + k) * G + j];
      for (int k = 0; k < K; ++k)
        for (int j = 0; j < G; ++j) A[((long)k * G + ib) * G + j] = A[((long)k * G + ib) * G + j] + P1[((long)b * K + k) * G + j];
    }
    for (int i = 0; i < G; ++i)
      for (int k = 0; k < K; ++k)
        for (int d = 0; d < C; ++d) {
          float acc = 0.f;
          for (int j = 0; j < G; ++j) acc = fmaf(A[((long)k * G + i) * G + j], code[((long)i * G + j) * C + d], acc);
          R[((long)i * K + k) * C + d] = acc;
        }
    for (long e = 0; e < (long)K * C; ++e) sums[e] = 0.f;
    for (int g0 = 0; g0 < G; g0 += ROW_GROUP)      /* two levels: the rows of a group of 2 patch rows ascending, then the groups ascending */
      for (long e = 0; e < (long)K * C; ++e) {
        float q = 0.f;
        for (int i = g0; i < G && i < g0 + ROW_GROUP; ++i) q = q + R[(long)i * K * C + e];
        sums[e] = sums[e] + q;
      }
    for (int k = 0; k < K; ++k) {
      if (cnt[k] == 0) continue;
      float n2 = 0.f;
      for (int d = 0; d < C; ++d) n2 = fmaf(sums[(long)k * C + d], sums[(long)k * C + d], n2);
      const float r = lin_rinv_norm(n2);
      for (int d = 0; d < C; ++d) cent[(long)k * C + d] = sums[(long)k * C + d] * r;
    }
  }
  assign(code, cent, &t, G, H, C, K, S, labels);
  if (cent_out) memcpy(cent_out, cent, sizeof(float) * K * C);
  free(i0); free(i1); free(w0); free(w1); free(rinv); free(cent); free(S); free(U); free(P0); free(P1); free(A); free(R); free(sums); free(cnt); free(row);
  return 0;
}

__attribute__((target_clones("fma", "default")))
int wvn_oracle_kmeans_pixels_linear(const float* code, int G, int H, int C, int K, int iters, int* labels, float* cent_out, float* x_out) {
  return wvn_oracle_kmeans_pixels_linear_ac(code, G, H, C, K, iters, labels, cent_out, x_out, 1);
}
