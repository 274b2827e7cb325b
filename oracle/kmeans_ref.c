/* Oracle (TEST INFRASTRUCTURE ONLY): the deterministic cosine k-means of oracle/interfaces.py::kmeans_cosine_labels in plain C.
 *
 * The numpy statement in interfaces.py is the readable definition; this file restates it operation for operation so that the
 * 448 x 448 pixel-resolution case (200 704 points x 20 centroids x 90 channels x 11 assignment passes = 4e9 fused multiply-adds)
 * finishes in a second instead of minutes: fmaf() is the correctly rounded fused multiply-add on every host (libm falls back to an
 * exact software form where the CPU has no FMA unit), and tests/test_oracle_stego.py pins this file against the numpy statement.
 * Algorithm (stego_interface.py:94-100 as this build reads it; PARITY UNPINNED, the `stego` package is absent):
 *   x_p   = code_p * (1 / max(||code_p||, 1e-12)),  ||.||^2 and every dot product an fma chain over the channel index from 0
 *   c_k^0 = x at index floor((2k+1) P / (2K))
 *   iters times: label_p = argmax_k <x_p, c_k> (lowest k wins ties); c_k = normalise(sum of its x_p) where the sum runs over
 *   chunks of 64 consecutive points (members in ascending order, from +0), chunk partials in ascending order inside groups of 8
 *   chunks (from +0), group partials in ascending order (from +0); an empty cluster keeps its centroid
 *   final labels = one more assignment.
 * Built by __graft_entry__.build() / oracle/build_oracle.py:  gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC kmeans_ref.c -lm */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK 64
/* chunk partials are folded in groups of SUPER consecutive chunks: 8 up to 8192 points, 16 above (csrc/stego.hip: km_super) */
#define SUPER (P > 8192 ? 16 : 8)

static float rinv_norm(float n2) {
  float n = sqrtf(n2); /* correctly rounded (IEEE) */
  if (n < 1e-12f) n = 1e-12f;
  return 1.0f / n;
}

/* the hot loop: FMA instructions where the CPU has them (chosen at load time), libm's exact fmaf otherwise */
__attribute__((target_clones("fma", "default")))
static void assign_labels(const float* x, const float* cent, int* lab, long P, int C, int K) {
#pragma omp parallel for schedule(static)
  for (long p = 0; p < P; ++p) {
    const float* xp = x + p * C;
    int best = 0;
    float bv = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const float* c = cent + (long)k * C;
      float acc = 0.f;
      for (int d = 0; d < C; ++d) acc = fmaf(xp[d], c[d], acc);
      if (acc > bv) { bv = acc; best = k; }
    }
    lab[p] = best;
  }
}

__attribute__((target_clones("fma", "default")))
static void normalize_rows(const float* code, float* x, long P, int C) {
#pragma omp parallel for schedule(static)
  for (long p = 0; p < P; ++p) {
    float n2 = 0.f;
    for (int d = 0; d < C; ++d) n2 = fmaf(code[p * C + d], code[p * C + d], n2);
    const float r = rinv_norm(n2);
    for (int d = 0; d < C; ++d) x[p * C + d] = code[p * C + d] * r;
  }
}

/* code [P][C] fp32 -> labels [P] int32 (not compacted), and -- when the pointers are not NULL -- the final centroids [K][C] and the
 * normalised rows [P][C] the labels were assigned from (tolerance analysis of end-to-end segment maps, oracle/segmap_agreement.py);
 * returns 0, or 1 when out of memory */
int wvn_oracle_kmeans_cosine_ex(const float* code, long P, int C, int K, int iters, int* labels, float* cent_out, float* x_out) {
  float* x = (float*)malloc((size_t)P * C * sizeof(float));
  float* cent = (float*)malloc((size_t)K * C * sizeof(float));
  float* sums = (float*)malloc((size_t)K * C * sizeof(float));
  float* grp = (float*)malloc((size_t)K * C * sizeof(float));
  float* part = (float*)malloc((size_t)K * C * sizeof(float));
  long* cnt = (long*)malloc((size_t)K * sizeof(long));
  if (!x || !cent || !sums || !grp || !part || !cnt) return 1;
  normalize_rows(code, x, P, C);
  for (int k = 0; k < K; ++k) {
    const long p0 = ((long)(2 * k + 1) * P) / (2 * K);
    memcpy(cent + (long)k * C, x + p0 * C, (size_t)C * sizeof(float));
  }
  for (int it = 0; it < iters; ++it) {
    assign_labels(x, cent, labels, P, C, K);
    for (long i = 0; i < (long)K * C; ++i) sums[i] = 0.f;
    for (int k = 0; k < K; ++k) cnt[k] = 0;
    for (long g0 = 0; g0 < P; g0 += (long)CHUNK * SUPER) {
      for (long i = 0; i < (long)K * C; ++i) grp[i] = 0.f;
      const long g1 = g0 + (long)CHUNK * SUPER < P ? g0 + (long)CHUNK * SUPER : P;
      for (long p0 = g0; p0 < g1; p0 += CHUNK) {
        for (long i = 0; i < (long)K * C; ++i) part[i] = 0.f;
        const long p1 = p0 + CHUNK < P ? p0 + CHUNK : P;
        for (long p = p0; p < p1; ++p) {
          float* t = part + (long)labels[p] * C;
          for (int d = 0; d < C; ++d) t[d] = t[d] + x[p * C + d];
          cnt[labels[p]] += 1;
        }
        for (long i = 0; i < (long)K * C; ++i) grp[i] = grp[i] + part[i];
      }
      for (long i = 0; i < (long)K * C; ++i) sums[i] = sums[i] + grp[i];
    }
    for (int k = 0; k < K; ++k) {
      if (cnt[k] == 0) continue;
      float n2 = 0.f;
      for (int d = 0; d < C; ++d) n2 = fmaf(sums[(long)k * C + d], sums[(long)k * C + d], n2);
      const float r = rinv_norm(n2);
      for (int d = 0; d < C; ++d) cent[(long)k * C + d] = sums[(long)k * C + d] * r;
    }
  }
  assign_labels(x, cent, labels, P, C, K);
  if (cent_out) memcpy(cent_out, cent, (size_t)K * C * sizeof(float));
  if (x_out) memcpy(x_out, x, (size_t)P * C * sizeof(float));
  free(x); free(cent); free(sums); free(grp); free(part); free(cnt);
  return 0;
}

int wvn_oracle_kmeans_cosine(const float* code, long P, int C, int K, int iters, int* labels) {
  return wvn_oracle_kmeans_cosine_ex(code, P, C, K, iters, labels, NULL, NULL);
}
