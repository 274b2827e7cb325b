// Exact-mode (fp32, VALU) flash attention used by the parity gate (<= 1e-3 vs the fp32 oracle).
// Same online-softmax algorithm as attention_bf16.hip, no reduced-precision step anywhere:
// 64 queries per 256-thread workgroup, 4 adjacent lanes per query; each lane scores 16 of the
// tile's 64 keys (keys 4*i+part, so the four lanes of a query read four different LDS rows whose
// float4 accesses fall on disjoint bank quads), exchanges P through LDS and accumulates 16 of the
// 64 output channels.
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int FQ = 64, FKV = 64, FD = 64, FSTR = 68;

__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out,
                                                            int heads, int ntok, int ntok_s, int npad, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[FKV * FSTR];
  __shared__ __attribute__((aligned(16))) float Vs[FKV * FSTR];
  __shared__ __attribute__((aligned(16))) float Ps[FQ * FSTR];
  const int tid = threadIdx.x;
  const int qi = tid >> 2, part = tid & 3;
  const int bh = blockIdx.y, b = bh / heads, head = bh - b * heads;
  const int qrow = blockIdx.x * FQ + qi;  // < npad always (npad % 64 == 0)

  float qr[FD];
  {
    const float* qg = q + ((size_t)bh * npad + qrow) * FD;
#pragma unroll
    for (int d = 0; d < FD; d += 4) {
      f32x4_t t = *(const f32x4_t*)(qg + d);
      qr[d] = t[0]; qr[d + 1] = t[1]; qr[d + 2] = t[2]; qr[d + 3] = t[3];
    }
  }
  float o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nt = (ntok + FKV - 1) / FKV;
  for (int it = 0; it < nt; ++it) {
    const int kv0 = it * FKV;
    __syncthreads();  // previous tile fully consumed
    for (int c = tid; c < FKV * FD / 4; c += 256) {
      int row = c >> 4, col = (c & 15) * 4;
      *(f32x4_t*)(Ks + row * FSTR + col) = *(const f32x4_t*)(k + ((size_t)bh * npad + kv0 + row) * FD + col);
      // rows >= ntok are padding (content not ours): zero them so that 0 * garbage cannot become NaN
      const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
      *(f32x4_t*)(Vs + row * FSTR + col) =
          (kv0 + row < ntok) ? *(const f32x4_t*)(v + ((size_t)bh * npad + kv0 + row) * FD + col) : zero;
    }
    __syncthreads();
    float s[16];
    float mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = 4 * i + part;
      const float* kr = Ks + key * FSTR;
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < FD; d += 4) {
        f32x4_t kk = *(const f32x4_t*)(kr + d);
        acc = fmaf(qr[d], kk[0], acc);
        acc = fmaf(qr[d + 1], kk[1], acc);
        acc = fmaf(qr[d + 2], kk[2], acc);
        acc = fmaf(qr[d + 3], kk[3], acc);
      }
      acc *= scale;
      if (kv0 + key >= ntok) acc = -INFINITY;
      s[i] = acc;
      mt = fmaxf(mt, acc);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 1, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 2, 64));
    const float m_new = fmaxf(m_run, mt);  // finite: every tile holds at least one valid key
    const float alpha = expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float p = expf(s[i] - m_new);
      ps += p;
      Ps[qi * FSTR + 4 * i + part] = p;
    }
    ps += __shfl_xor(ps, 1, 64);
    ps += __shfl_xor(ps, 2, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] *= alpha;
    __syncthreads();
    for (int key = 0; key < FKV; ++key) {
      const float p = Ps[qi * FSTR + key];
      const float* vr = Vs + key * FSTR + part * 16;
#pragma unroll
      for (int d = 0; d < 16; d += 4) {
        f32x4_t vv = *(const f32x4_t*)(vr + d);
        o[d] = fmaf(p, vv[0], o[d]);
        o[d + 1] = fmaf(p, vv[1], o[d + 1]);
        o[d + 2] = fmaf(p, vv[2], o[d + 2]);
        o[d + 3] = fmaf(p, vv[3], o[d + 3]);
      }
    }
  }
  if (qrow < ntok) {
    const float inv = 1.0f / l_run;
    float* og = out + ((size_t)b * ntok_s + qrow) * (heads * FD) + head * FD + part * 16;
#pragma unroll
    for (int d = 0; d < 16; d += 4) {
      f32x4_t t = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
      *(f32x4_t*)(og + d) = t;
    }
  }
}

}  // namespace

int wvn_attention_f32_launch(const float* q, const float* k, const float* v, float* out, int B, int heads, int ntok,
                             int ntok_s, int npad, float scale, hipStream_t st) {
  if (!q || !k || !v || !out || npad % FQ != 0 || npad < ntok) return WVN_ERR_ARG;
  dim3 grid(ceil_div(ntok, FQ), B * heads);
  hipLaunchKernelGGL(attention_f32_kernel, grid, dim3(256), 0, st, q, k, v, out, heads, ntok, ntok_s, npad, scale);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
