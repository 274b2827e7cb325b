// Fused (flash-style) multi-head self-attention for the ViT blocks on gfx950, bf16 in / fp32 accumulate.
//
//   softmax(Q K^T * scale) V   for dh = 64, sequence length ntok (3137 at 448^2/8), never
//   materialising the ntok x ntok score matrix (3.8 GB per 32 frames).
//
// Work decomposition: one workgroup = 128 queries of one (frame, head); 256 threads = 4 waves, each
// wave owns 32 queries; the workgroup streams K / V^T tiles of 64 keys through a double-buffered LDS
// ring.  K/V tiles travel global -> registers -> LDS with PD tiles in flight (register slots are
// compile-time: the tile loop is unrolled by PD); one barrier per tile.
//
// XCD-aware placement: the 25 query blocks of a (frame, head) re-read the same 819 KB of K/V.  The
// dispatcher puts workgroup b on XCD b % 8, so the 1-D grid is decoded such that ALL query blocks of a
// (frame, head) land on one XCD, consecutively: ~4 (frame, head) pairs are live per XCD at a time and
// their K/V (3.3 MB) stay in that XCD's 4 MB L2 instead of thrashing it with 30 pairs.  (Speed only.)
//
// Everything is computed TRANSPOSED so that the softmax is lane-local:
//   S^T = K Q^T   : A = K tile (rows = keys, from LDS), B = Q^T (registers, loaded once)
//                   -> accumulator lane l holds query q = l&31 and 32 of the tile's 64 keys
//                      (rows (r&3)+8(r>>2)+4(l>>5) of each 32-key sub-tile); the other 32 keys of
//                      the same query sit in lane l^32  => row max = in-lane max + one lane swap.
//   O^T = V^T P^T : A = V^T tile (rows = d, keys contiguous; produced by the QKV GEMM epilogue),
//                   B = P^T straight from the S^T accumulator registers: MFMA sums over its k-slots
//                   in a fixed but arbitrary order, so it suffices that A and B agree on which key
//                   sits in slot (l>>5, j).  We define slot (hi, j) <-> key 16*ks + (j&3) + 8*(j>>2)
//                   + 4*hi, which is exactly where the S^T accumulator already holds it: P needs NO
//                   cross-lane movement, and V^T fragments are two 8-byte LDS reads.
//                   -> O^T accumulator lane l holds query l&31 again, so the online-softmax rescale
//                      and the final 1/l are per-lane scalars.
// exp is evaluated as exp2 with scale*log2(e) folded into one FMA; the O rescale is skipped (wave-
// uniformly) when no running max moved.  Keys >= ntok (tile tail / padding, content not ours) are
// neutralised: their scores by select, their V^T columns by zeroing on the way into LDS.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int QB = 128;       // queries per workgroup (4 waves x 32)
constexpr int KVB = 64;       // keys per tile
constexpr int DH = 64;        // head dim
constexpr int LSTR = DH + 8;  // LDS row stride in bf16 (144 B)
constexpr int TILE_ELEMS = KVB * LSTR;

template <int PD, bool XCDMAP, bool PRIO, bool RESCALE_ALWAYS = false>
__global__ __launch_bounds__(256, 2) void attention_bf16_kernel(const bf16_t* __restrict__ q,
                                                                const bf16_t* __restrict__ k,
                                                                const bf16_t* __restrict__ vt,
                                                                bf16_t* __restrict__ out, int heads, int nbh, int nqb,
                                                                int ntok, int ntok_s, int npad, float c_exp) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * TILE_ELEMS];  // [stage][K | Vt][64][72]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int bh, qb;
  if constexpr (XCDMAP) {  // nbh % 8 == 0 (checked by the launcher)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bh = (idx / nqb) * 8 + xcd;
    qb = idx % nqb;
  } else {
    bh = blockIdx.x / nqb;
    qb = blockIdx.x - bh * nqb;
  }
  const int b = bh / heads, head = bh - b * heads;
  const int q0 = qb * QB + wave * 32;

  const bf16_t* qg = q + ((size_t)bh * npad + q0 + l31) * DH + hi * 8;
  bf16x8_t qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8_t*)(qg + s * 16);

  // staging: 2 K chunks + 2 V^T chunks (16 B) per thread per tile, PD register sets
  const int srow = tid >> 3, skc = tid & 7;
  const bf16_t* kg = k + ((size_t)bh * npad + srow) * DH + skc * 8;   // + kv0*DH, rows srow, srow+32
  const bf16_t* vg = vt + ((size_t)bh * DH + srow) * npad + skc * 8;  // + kv0,    rows srow, srow+32
  u32x4_t rk[PD][2], rv[PD][2];
  auto load_regs = [&](int kv0, u32x4_t (&a)[2], u32x4_t (&c)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *(const u32x4_t*)(kg + (size_t)(kv0 + 32 * i) * DH);
      c[i] = *(const u32x4_t*)(vg + (size_t)(32 * i) * npad + kv0);
    }
  };
  auto store_regs = [&](int stage, int kv0, const u32x4_t (&a)[2], const u32x4_t (&c)[2], bool may_be_tail) {
    bf16_t* Ks = lds + stage * 2 * TILE_ELEMS;
    bf16_t* Vs = Ks + TILE_ELEMS;
    const bool tail = may_be_tail && kv0 + KVB > ntok;  // workgroup-uniform
    const int kbase = kv0 + skc * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4_t v = c[i];
      if (tail) {  // keys >= ntok: zero the V^T columns (0 * NaN would poison the PV MFMA)
#pragma unroll
        for (int w = 0; w < 4; ++w)
          v[w] &= (kbase + 2 * w < ntok ? 0x0000ffffu : 0u) | (kbase + 2 * w + 1 < ntok ? 0xffff0000u : 0u);
      }
      *(u32x4_t*)(Ks + (srow + 32 * i) * LSTR + skc * 8) = a[i];
      *(u32x4_t*)(Vs + (srow + 32 * i) * LSTR + skc * 8) = v;
    }
  };

  f32x16_t ot[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (ntok + KVB - 1) / KVB;
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < nt) load_regs(u * KVB, rk[u], rv[u]);
  store_regs(0, 0, rk[0], rv[0], true);
  if (PD < nt) load_regs(PD * KVB, rk[0], rv[0]);
  __syncthreads();
  // Make the Q fragments "used" here: otherwise hipcc places their vmcnt wait at the first use INSIDE the
  // tile loop, and on every trip that s_waitcnt vmcnt(0) drains the K/V prefetches as well.
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));

  // one K/V tile: S^T = K Q^T, online softmax, O^T += V^T P^T  (operands from LDS stage `stage`)
  auto compute_tile = [&](int kv0, int stage, auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;  // steady-state tiles are never the tail tile
      const bf16_t* Ks = lds + stage * 2 * TILE_ELEMS;
      const bf16_t* Vs = Ks + TILE_ELEMS;

      // ---- S^T = K Q^T : two 32-key sub-tiles ----
      f32x16_t st[2];
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
        const bf16_t* kb = Ks + (t * 32 + l31) * LSTR + hi * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bf16x8_t kf = *(const bf16x8_t*)(kb + s * 16);
          st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[t], 0, 0, 0);
        }
      }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      // ---- mask the tail of the last tile ----
      if (!STEADY && kv0 + KVB > ntok) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= ntok) st[t][r] = -1e30f;
          }
      }
      // ---- online softmax (lane = query, both half-waves share the running max) ----
      float mt = st[0][0];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[t][r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (RESCALE_ALWAYS ? true : __any(mt > m_run)) {
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
      }
      const float mc = -m_run * c_exp;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(st[t][r], c_exp, mc));
          st[t][r] = pv;
          psum += pv;
        }
      l_run += psum;

      // ---- O^T += V^T P^T ----
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int t = ks >> 1, h8 = (ks & 1) * 8;
        union { u32x4_t u; bf16x8_t v; } pf;
        pf.u[0] = pack_bf16x2(st[t][h8 + 0], st[t][h8 + 1]);
        pf.u[1] = pack_bf16x2(st[t][h8 + 2], st[t][h8 + 3]);
        pf.u[2] = pack_bf16x2(st[t][h8 + 4], st[t][h8 + 5]);
        pf.u[3] = pack_bf16x2(st[t][h8 + 6], st[t][h8 + 7]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16_t* vb = Vs + (dt * 32 + l31) * LSTR + ks * 16 + hi * 4;
          union { u32x2_t h[2]; bf16x8_t v; } vf;
          vf.h[0] = *(const u32x2_t*)(vb);
          vf.h[1] = *(const u32x2_t*)(vb + 8);
          ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, ot[dt], 0, 0, 0);
        }
      }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };

  int it = 0;
  // steady state: straight-line per PD tiles, no conditions around the loads -> hipcc counts vmcnt exactly
  // (waits only for the tile being moved to LDS; the younger PD-1 tiles stay in flight across the barrier)
  for (; it + 2 * PD < nt; it += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int t = it + u;
      compute_tile(t * KVB, t & 1, std::true_type{});
      store_regs((t + 1) & 1, (t + 1) * KVB, rk[(u + 1) % PD], rv[(u + 1) % PD], false);
      load_regs((t + 1 + PD) * KVB, rk[(u + 1) % PD], rv[(u + 1) % PD]);
      __syncthreads();
    }
  }
  // drain: the last < 2*PD + PD tiles, same slot pattern with the end-of-sequence conditions
  for (; it < nt; it += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int t = it + u;
      if (t >= nt) break;
      compute_tile(t * KVB, t & 1, std::false_type{});
      if (t + 1 < nt) {
        store_regs((t + 1) & 1, (t + 1) * KVB, rk[(u + 1) % PD], rv[(u + 1) % PD], true);
        if (t + 1 + PD < nt) load_regs((t + 1 + PD) * KVB, rk[(u + 1) % PD], rv[(u + 1) % PD]);
      }
      __syncthreads();
    }
  }

  // ---- normalise and store: out[b*ntok_s + q][head*64 + d] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q0 + l31;
  if (qi < ntok) {
    bf16_t* og = out + ((size_t)b * ntok_s + qi) * (heads * DH) + head * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t o;
        o[0] = pack_bf16x2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
        o[1] = pack_bf16x2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
        *(u32x2_t*)(og + dt * 32 + 8 * g + 4 * hi) = o;
      }
  }
}

// WVN_ATTN_VARIANT (A/B switch): 0 = PD1, plain block order; 1 = PD1 + XCD map; 2 = PD2 + XCD map (default);
// 3 = PD2 + XCD map + s_setprio around the MFMA clusters
int attn_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("WVN_ATTN_VARIANT");
    v = e ? atoi(e) : 2;
    if (v < 0 || v > 3) v = 2;
  }
  return v;
}

}  // namespace

int wvn_attention_bf16_launch(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out, int B, int heads,
                              int ntok, int ntok_s, int npad, float scale, hipStream_t st) {
  if (!q || !k || !vt || !out || npad % QB != 0 || npad < ntok) return WVN_ERR_ARG;
  const float c_exp = scale * 1.44269504088896340736f;
  const int nqb = ceil_div(ntok, QB), nbh = B * heads;
  dim3 grid(nqb * nbh), block(256);
  int v = attn_variant();
  if (nbh % 8 != 0 && v > 0) v = 0;  // the XCD decode needs whole groups of 8 (frame, head) pairs
  switch (v) {
    case 0: hipLaunchKernelGGL((attention_bf16_kernel<1, false, false>), grid, block, 0, st, q, k, vt, out, heads, nbh, nqb, ntok, ntok_s, npad, c_exp); break;
    case 1: hipLaunchKernelGGL((attention_bf16_kernel<1, true, false>), grid, block, 0, st, q, k, vt, out, heads, nbh, nqb, ntok, ntok_s, npad, c_exp); break;
    case 3: hipLaunchKernelGGL((attention_bf16_kernel<2, true, true>), grid, block, 0, st, q, k, vt, out, heads, nbh, nqb, ntok, ntok_s, npad, c_exp); break;
    default: hipLaunchKernelGGL((attention_bf16_kernel<2, true, false>), grid, block, 0, st, q, k, vt, out, heads, nbh, nqb, ntok, ntok_s, npad, c_exp); break;
  }
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
