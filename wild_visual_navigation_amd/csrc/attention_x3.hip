// Exact-mode fused (flash-style) multi-head self-attention on the matrix pipe: the transposed scheme of attention_bf16.hip
// with every MFMA operand carried as hi + lo bf16 planes (16 significant bits) and every product formed as
// hi*hi + hi*lo + lo*hi in fp32 accumulators:
//   S^T = K Q^T   : K_hi Q_hi + K_hi Q_lo + K_lo Q_hi                      (24 MFMAs per 64-key tile and wave)
//   P             : exp2 of the fp32 scores, split into P_hi + P_lo in registers; the row sums add the fp32 P itself
//   O^T = V^T P^T : V_hi P_hi + V_hi P_lo + V_lo P_hi                      (24 MFMAs)
// The softmax VALU work is that of the bf16 kernel plus the split (one subtract + one conversion per pair of P), spread
// over three times the MFMAs, so this form is much closer to matrix-pipe bound than the bf16 one.
//
// One workgroup = 128 queries of one (frame, head); 4 waves x 32 queries; K / V^T hi and lo tiles of 64 keys arrive by DMA
// (buffer_load ... lds) in a 2-stage ring of 4 x 8 KB (64 KB of LDS, two workgroups per CU), XOR-swizzled exactly like the
// bf16 kernel's tiles, every fragment one conflict-free ds_read_b128.  V^T planes use the bf16 kernel's token permutation
// (bits 2 and 3 of the token index swapped inside aligned groups of 16); the exact-mode QKV epilogue (gemm_x3.hip) writes it.
// Keys >= ntok are masked by score; K / V^T padding must be finite.  Output: hi / lo planes [B * ntok_s, heads * 64] for the
// projection GEMM.
#include <type_traits>

#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int QB = 128, KVB = 64, DH = 64;
constexpr int TILE_BYTES = KVB * DH * 2;  // 8 KB per plane tile
constexpr int NST = 2;

__device__ inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <bool XCDMAP>
__global__ __launch_bounds__(256, 2) void attention_x3_kernel(const bf16_t* __restrict__ q_hi, const bf16_t* __restrict__ q_lo,
                                                              const bf16_t* __restrict__ k_hi, const bf16_t* __restrict__ k_lo,
                                                              const bf16_t* __restrict__ vt_hi, const bf16_t* __restrict__ vt_lo,
                                                              bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                              int heads, int nbh, int nqb, int ntok, int ntok_s, int npad,
                                                              float c_exp) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * 4 * TILE_BYTES];  // [stage][K hi | K lo | Vt hi | Vt lo][64][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int bh, qb;
  if constexpr (XCDMAP) {  // nbh % 8 == 0 (checked by the launcher): all query blocks of a (frame, head) on one XCD
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bh = (idx / nqb) * 8 + xcd;
    qb = idx % nqb;
  } else {
    bh = blockIdx.x / nqb;
    qb = blockIdx.x - bh * nqb;
  }
  const int b = bh / heads, head = bh - b * heads;
  const int q0 = qb * QB + wave * 32;

  // ---- DMA: per tile and plane 8 wave-instructions of 1 KB (8 rows x 128 B); each wave issues 2 per plane ----
  const unsigned kv_bytes = (unsigned)((size_t)nbh * npad * DH * 2);
  const __amdgpu_buffer_rsrc_t rs[4] = {
      __builtin_amdgcn_make_buffer_rsrc((void*)k_hi, 0, kv_bytes, 0x00020000),
      __builtin_amdgcn_make_buffer_rsrc((void*)k_lo, 0, kv_bytes, 0x00020000),
      __builtin_amdgcn_make_buffer_rsrc((void*)vt_hi, 0, kv_bytes, 0x00020000),
      __builtin_amdgcn_make_buffer_rsrc((void*)vt_lo, 0, kv_bytes, 0x00020000)};
  unsigned koff[2], voff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (wave * 2 + u) * 8 + (lane >> 3);          // key (K tile) or d (V^T tile)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);           // source chunk that lands in LDS chunk lane & 7
    koff[u] = (unsigned)((((size_t)bh * npad + row) * DH + chunk * 8) * 2);   // + kv0 * 128
    voff[u] = (unsigned)((((size_t)bh * DH + row) * npad + chunk * 8) * 2);   // + kv0 * 2
  }
  auto issue = [&](int t) {
    unsigned char* dst = lds + (t % NST) * 4 * TILE_BYTES + wave * 2048;
    const unsigned ks = __builtin_amdgcn_readfirstlane(t * KVB * DH * 2), vs = __builtin_amdgcn_readfirstlane(t * KVB * 2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[0], (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, koff[u], ks, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[1], (__attribute__((address_space(3))) void*)(dst + TILE_BYTES + u * 1024), 16, koff[u], ks, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[2], (__attribute__((address_space(3))) void*)(dst + 2 * TILE_BYTES + u * 1024), 16, voff[u], vs, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[3], (__attribute__((address_space(3))) void*)(dst + 3 * TILE_BYTES + u * 1024), 16, voff[u], vs, 0, 0);
    }
  };
  const int nt = (ntok + KVB - 1) / KVB;
  issue(0);

  // ---- Q^T fragments (B operand): query l31, d = 16 s + 8 hi .. + 7 ----
  const size_t qoff = ((size_t)bh * npad + q0 + l31) * DH + hi * 8;
  bf16x8_t qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qh[s] = *(const bf16x8_t*)(q_hi + qoff + s * 16);
    ql[s] = *(const bf16x8_t*)(q_lo + qoff + s * 16);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {  // consume the loads here, not inside the tile loop (their vmcnt wait would drain the DMA queue there)
    asm volatile("" : "+v"(qh[s]));
    asm volatile("" : "+v"(ql[s]));
  }

  f32x16_t ot[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f, l_run1 = 0.f;

  const unsigned rd_row = l31 * 128;
  const int xorc = (l31 >> 1) & 7;
  unsigned rdo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rdo[s] = rd_row + (((2 * s + hi) ^ xorc) << 4);

  auto compute_tile = [&](int kv0, int stage, int t_issue, auto tail_tag) {
    constexpr bool MAYBE_TAIL = decltype(tail_tag)::value;
    unsigned fa[4];
    unsigned so = (unsigned)(stage * 4 * TILE_BYTES);
    asm volatile("" : "+s"(so));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa[s] = rdo[s] + so;
      asm volatile("" : "+v"(fa[s]));
    }
    // ---- S^T = K Q^T, three products per fragment pair (small terms first) ----
    f32x16_t st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8_t kh = *(const bf16x8_t*)(lds + fa[s] + t * 4096);
        const bf16x8_t kl = *(const bf16x8_t*)(lds + fa[s] + TILE_BYTES + t * 4096);
        if (s == 0) st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], (f32x16_t)(0.f), 0, 0, 0);
        else st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], st[t], 0, 0, 0);
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], st[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t_issue >= 0) issue(t_issue);
    __builtin_amdgcn_sched_barrier(0);
    if (MAYBE_TAIL && kv0 + KVB > ntok) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= ntok) st[t][r] = -1e30f;
        }
    }
    // ---- online softmax (lane = query; the two half-waves share the running max) ----
    float ma = max3f(st[0][0], st[0][1], st[0][2]), mb = max3f(st[1][0], st[1][1], st[1][2]);
    ma = max3f(ma, st[0][3], st[0][4]); mb = max3f(mb, st[1][3], st[1][4]);
#pragma unroll
    for (int r = 5; r < 15; r += 2) { ma = max3f(ma, st[0][r], st[0][r + 1]); mb = max3f(mb, st[1][r], st[1][r + 1]); }
    float mt = max3f(ma, mb, st[0][15]);
    mt = fmaxf(mt, st[1][15]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = max3f(mt, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    // deferred, wave-uniform rescale (P bounded by 2^THR instead of 1: harmless in fp32, cancels in O / l)
    constexpr float THR = 6.0f;
    if (__any((mt - m_run) * c_exp > THR)) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
      m_run = m_new;
      l_run *= alpha;
      l_run1 *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
    }
    const float mc = -m_run * c_exp;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = __builtin_amdgcn_exp2f(fmaf(st[t][r], c_exp, mc));

    // ---- O^T += V^T P^T : 4 groups of 16 keys; P split into hi + lo planes in registers ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int t = ks >> 1, h8 = (ks & 1) * 8;
      union { u32x4_t u; bf16x8_t v; } ph, pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = st[t][h8 + 2 * e], p1 = st[t][h8 + 2 * e + 1];
        const uint32_t hw = pack_bf16x2(p0, p1);
        ph.u[e] = hw;
        pl.u[e] = pack_bf16x2(p0 - __uint_as_float(hw << 16), p1 - __uint_as_float(hw & 0xffff0000u));
        l_run += p0;
        l_run1 += p1;
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8_t vh = *(const bf16x8_t*)(lds + fa[ks] + 2 * TILE_BYTES + dt * 4096);
        const bf16x8_t vl = *(const bf16x8_t*)(lds + fa[ks] + 3 * TILE_BYTES + dt * 4096);
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl.v, ot[dt], 0, 0, 0);
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph.v, ot[dt], 0, 0, 0);
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph.v, ot[dt], 0, 0, 0);
      }
    }
  };

  // tile loop: wait for tile t, barrier (everyone has it and has finished reading tile t - 1, whose slot the next DMA
  // overwrites), request tile t + 1 after the QK^T MFMAs have been issued, compute.  The last tile (the only one that may need
  // masking) is peeled so that the accumulators keep their registers across the two instantiations.
  for (int t = 0; t + 1 < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute_tile(t * KVB, t % NST, t + 1, std::false_type{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  compute_tile((nt - 1) * KVB, (nt - 1) % NST, -1, std::true_type{});

  // ---- normalise, split, store: out[b*ntok_s + q][head*64 + d] ----
  l_run += l_run1;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q0 + l31;
  if (qi < ntok) {
    const size_t o = ((size_t)b * ntok_s + qi) * (heads * DH) + head * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t oh, ol;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float a = ot[dt][4 * g + 2 * e] * inv, c = ot[dt][4 * g + 2 * e + 1] * inv;
          const uint32_t hw = pack_bf16x2(a, c);
          oh[e] = hw;
          ol[e] = pack_bf16x2(a - __uint_as_float(hw << 16), c - __uint_as_float(hw & 0xffff0000u));
        }
        *(u32x2_t*)(out_hi + o + dt * 32 + 8 * g + 4 * hi) = oh;
        *(u32x2_t*)(out_lo + o + dt * 32 + 8 * g + 4 * hi) = ol;
      }
  }
}

}  // namespace

// q*, k*: [B, heads, npad, 64] planes; vt*: [B, heads, 64, npad] planes (token-permuted like the bf16 path); out*: [B*ntok_s, heads*64]
int wvn_attention_x3_launch(const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* k_hi, const bf16_t* k_lo, const bf16_t* vt_hi,
                            const bf16_t* vt_lo, bf16_t* out_hi, bf16_t* out_lo, int B, int heads, int ntok, int ntok_s,
                            int npad, float scale, hipStream_t st) {
  if (!q_hi || !q_lo || !k_hi || !k_lo || !vt_hi || !vt_lo || !out_hi || !out_lo || npad % QB != 0 || npad < ntok || scale <= 0.f)
    return WVN_ERR_ARG;
  const int nqb = ceil_div(ntok, QB), nbh = B * heads;
  if ((size_t)nbh * npad * DH * 2 >= (1ull << 32)) return WVN_ERR_ARG;  // 32-bit buffer offsets
  const float c_exp = scale * 1.44269504088896340736f;
  dim3 grid(nqb * nbh);
  if ((nbh % 8) == 0)
    hipLaunchKernelGGL((attention_x3_kernel<true>), grid, dim3(256), 0, st, q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_hi, out_lo,
                       heads, nbh, nqb, ntok, ntok_s, npad, c_exp);
  else
    hipLaunchKernelGGL((attention_x3_kernel<false>), grid, dim3(256), 0, st, q_hi, q_lo, k_hi, k_lo, vt_hi, vt_lo, out_hi, out_lo,
                       heads, nbh, nqb, ntok, ntok_s, npad, c_exp);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
