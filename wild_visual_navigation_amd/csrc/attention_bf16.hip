// Fused (flash-style) multi-head self-attention for the ViT blocks on gfx950, bf16 in / fp32 accumulate.
//
//   softmax(Q K^T * scale) V   for dh = 64, sequence length ntok (3137 at 448^2/8), never
//   materialising the ntok x ntok score matrix (3.8 GB per 32 frames).
//
// Work decomposition: one workgroup = 128 queries of one (frame, head); 256 threads = 4 waves, each
// wave owns 32 queries; the workgroup streams K / V^T tiles of 64 keys through an LDS ring.
//
// K / V^T tiles go global -> LDS by DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write
// traffic (the LDS pipe is the scarce resource here: every MFMA needs 1 KB of it).  The DMA writes
// lane-linear, so tiles are unpadded ([64 rows][128 B]) and bank conflicts are removed by XOR-swizzling the
// 16-byte chunk index with (row >> 1) & 7 on the per-lane SOURCE address and again on the read: every
// fragment read is one conflict-free ds_read_b128.  One s_barrier per tile; with NST ring stages NST - 1
// tiles are in flight, waited for with a counted vmcnt.
//
// XCD-aware placement: the 25 query blocks of a (frame, head) re-read the same 819 KB of K/V.  The
// dispatcher puts workgroup b on XCD b % 8, so the 1-D grid is decoded such that ALL query blocks of a
// (frame, head) land on one XCD, consecutively, and their K/V stay in that XCD's 4 MB L2.  (Speed only.)
//
// Everything is computed TRANSPOSED so that the softmax is lane-local:
//   S^T = K Q^T   : A = K tile (rows = keys, from LDS), B = Q^T (registers, loaded once)
//                   -> accumulator lane l holds query q = l&31 and 32 of the tile's 64 keys
//                      (rows (r&3)+8(r>>2)+4(l>>5) of each 32-key sub-tile); the other 32 keys of
//                      the same query sit in lane l^32  => row max = in-lane max + one lane swap.
//   O^T = V^T P^T : A = V^T tile (rows = d, keys along the row), B = P^T straight from the S^T accumulator
//                   registers.  MFMA sums over its k-slots in a fixed but arbitrary order, so A and B only have
//                   to agree on which key sits in slot (l>>5, j): the accumulator already holds, for half-wave
//                   hi, keys {4hi..4hi+3} and {8+4hi..8+4hi+3} of each group of 16.  V^T is therefore STORED
//                   with the tokens of every aligned group of 16 permuted (bits 2 and 3 of the token index
//                   swapped: 0-3, 8-11, 4-7, 12-15), which makes those 8 keys one contiguous 16-byte chunk.
//                   The QKV GEMM epilogues write V^T in this order (wvn_hip.h documents it for direct callers).
//                   -> O^T accumulator lane l holds query l&31 again, so the online-softmax rescale
//                      and the final 1/l are per-lane scalars.
// Softmax, two forms of the same online algorithm:
//   raw q (scale > 0)   : exp2 with scale*log2(e) folded into one v_pk_fma per two scores;
//   pre-scaled q (PRE)  : q already carries scale*log2(e) (the QKV epilogue multiplies before the bf16 rounding) and the
//                         running max enters the S^T chains as the MFMA C operand, so the accumulators come out as exp2
//                         arguments and the per-element fma disappears.  This is the form wvn_vit_forward uses.
// In both the O rescale is deferred and wave-uniform (it runs when some row's score outgrows the running max by 2^6,
// i.e. almost only on the first tiles).  The kernel is bound by VALU issue, not by the matrix pipe or LDS (DESIGN.md
// section 4): what is left of the softmax is 32 v_exp, 16 v_max3, 16 v_cvt_pk, 16 v_dot2c per 16 MFMAs.
// Keys >= ntok (tile tail / padding) are neutralised by a select on their scores (p = 0); K / V^T padding must be
// finite (0 * finite = 0 in the PV MFMA).  The masked last tile is peeled out of the tile loop.
#include <stdlib.h>

#include <type_traits>

#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int QB = 128;            // queries per workgroup (4 waves x 32)
constexpr int KVB = 64;            // keys per tile
constexpr int DH = 64;             // head dim
constexpr int TILE_BYTES = KVB * DH * 2;  // 8 KB (K tile; V^T tile is the same size)

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// The softmax is bounded by per-wave VALU issue, so instruction count matters: the row max uses 3-input max.
// This file is compiled with -fno-honor-nans so that fmaxf on MFMA results needs no canonicalising v_max (scores
// are finite; masking uses -1e30, not infinities).  NOT inline asm: hipcc pads no MFMA-result hazards for an asm
// statement's operands (a v_max3 in asm read accumulator registers before the MFMA had written them).
__device__ inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// PRE: q arrives pre-multiplied by scale * log2(e) (the QKV GEMM epilogue does it before rounding to bf16, so no extra
// rounding) and the running max enters the S^T MFMA chain as its C operand (a 16-register block holding -M, rewritten
// only when the max moves): the accumulators come out as exp2 arguments and the 16 v_pk_fma per tile disappear.
// SUM: how the row sums are formed.  0: v_dot2c_f32_bf16 on the packed P (16 per tile); 1: plain adds on the fp32 P.
// QSPLIT: 0 = one q plane; 1 = q as two planes of the operand format (eight more MFMAs per tile); 2 (round 6, fp16 build) = the second plane as ONE scaled
// e5m2 MFMA of K = 64 per 32-key sub-tile: q_lo -> e5m2(q_lo * 2^12) once per workgroup, the keys' e5m2 image = the top bytes of the fp16 K fragments already in
// registers (two v_perm per fragment: truncation, a 9 % shortfall of a term that is itself 2^-12 of the score) -- 64 matrix-pipe cycles per sub-tile instead of 128
template <int NST, bool XCDMAP, int OCC, bool TIMING = false, bool PRE = false, int SUM = 0, bool LAZY = false, int QSPLIT = 0>
__global__ __launch_bounds__(256, OCC) void attention_bf16_kernel(const op16_t* __restrict__ q,
                                                                  const op16_t* __restrict__ k,
                                                                  const op16_t* __restrict__ vt,
                                                                  op16_t* __restrict__ out, int heads, int nbh,
                                                                  int nqb, int ntok, int ntok_s, int npad,
                                                                  float c_exp, long long* dbg, op16_t* __restrict__ out_lo,
                                                                  const op16_t* __restrict__ q_lo, int out_frag) {
  wvn_fp16_saturate();
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * 2 * TILE_BYTES];  // [stage][K | Vt][64][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // ---- K / V^T DMA: per tile 8 + 8 wave-instructions of 1 KB (8 rows x 128 B); each wave issues 2 + 2 -------
  const unsigned kv_bytes = (unsigned)((size_t)nbh * npad * DH * 2);
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)k, 0, kv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vt, 0, kv_bytes, 0x00020000);
  unsigned koff[2], voff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (wave * 2 + u) * 8 + (lane >> 3);          // key (K tile) or d (V^T tile)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);           // source chunk that lands in LDS chunk lane & 7
    koff[u] = (unsigned)((((size_t)bh * npad + row) * DH + chunk * 8) * 2);   // + kv0 * 128
    voff[u] = (unsigned)((((size_t)bh * DH + row) * npad + chunk * 8) * 2);   // + kv0 * 2
  }
  auto issue = [&](int t) {
    unsigned char* dst = lds + (t % NST) * 2 * TILE_BYTES + wave * 2048;
    const unsigned ks = __builtin_amdgcn_readfirstlane(t * KVB * DH * 2), vs = __builtin_amdgcn_readfirstlane(t * KVB * 2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16,
                                               koff[u], ks, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (__attribute__((address_space(3))) void*)(dst + TILE_BYTES + u * 1024),
                                               16, voff[u], vs, 0, 0);
    }
  };
  const int nt = (ntok + KVB - 1) / KVB;
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nt) issue(t);

  // ---- Q^T fragments (B operand): query l31, d = 16 s + 8 hi .. + 7 ------------------------------------------
  const op16_t* qg = q + ((size_t)bh * npad + q0 + l31) * DH + hi * 8;
  opx8_t qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *(const opx8_t*)(qg + s * 16);
  // Make the Q fragments "used" here: otherwise hipcc places their vmcnt wait at the first use INSIDE the
  // tile loop, where it would also drain the K/V DMA queue on every trip.
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));
  // QSPLIT (WVN_PREC_MIX): q as TWO planes, q = qf + ql (ql = the rounding residue of qf in the same format): S^T = K qf^T + K ql^T, eight
  // more MFMAs per tile on the K fragments already read.  Why q and not k: a key's rounding error is independent from key to key and
  // averages out over the thousands of keys of a row; the query's error is the SAME direction against every key of its row -- on the
  // reference's real 448^2 frame q alone accounts for the 1.0e-3 token error of single-plane attention (split: 7.6e-5; k split: no
  // change; profiles/r04d_error_budget_real_frame.md)
  opx8_t ql[4];
  typedef __attribute__((ext_vector_type(8))) int i32x8_t;
  i32x8_t ql8 = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (QSPLIT != 0) {
    const op16_t* qlg = q_lo + ((size_t)bh * npad + q0 + l31) * DH + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) ql[s] = *(const opx8_t*)(qlg + s * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(ql[s]));
  }
#if WVN_OPERAND_F16
  if constexpr (QSPLIT == 2) {   // byte (s, j) of the lane's 32 = e5m2(q_lo[16 s + 8 hi + j] * 2^12): the order the K fragments' top bytes are gathered in below
    typedef __attribute__((ext_vector_type(2))) short s16x2_t;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4_t raw = __builtin_bit_cast(u32x4_t, ql[s]);
      uint32_t d[2] = {0, 0};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t pr = raw[e];   // (scalar copy: clang's bit_cast of a vector element reads element 0)
        const s16x2_t o = __builtin_bit_cast(s16x2_t, d[e >> 1]);
        d[e >> 1] = __builtin_bit_cast(uint32_t, (e & 1) ? __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f / 4096.0f, true)
                                                         : __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f / 4096.0f, false));
      }
      ql8[2 * s] = (int)d[0];
      ql8[2 * s + 1] = (int)d[1];
    }
    asm volatile("" : "+v"(ql8));
  }
#endif

  f32x16_t ot[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
  float m_run = PRE ? 0.f : -1e30f;  // PRE: the running max in exp2 units (0 until the first tile sets it)
  float l_run = 0.f, l_run1 = 0.f;    // two independent row-sum chains
  f32x16_t cneg;                                  // PRE: -m_run in every register (C operand of the S^T chains)
#pragma unroll
  for (int r = 0; r < 16; ++r) cneg[r] = 0.f;
  bool fresh = true;                              // PRE: no tile processed yet (wave-uniform)

  // per-lane read offset inside a tile: row l31 (+32 per sub-tile), chunk XOR key (row >> 1) & 7 (same for row+32)
  const unsigned rd_row = l31 * 128;
  const int xorc = (l31 >> 1) & 7;
  unsigned rdo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) rdo[s] = rd_row + (((2 * s + hi) ^ xorc) << 4);

  long long tm[5] = {0, 0, 0, 0, 0};  // TIMING: wait+barrier+issue, QK^T, softmax, PV, total
  auto now = [&]() -> long long {
    if constexpr (TIMING) { __builtin_amdgcn_sched_barrier(0); return (long long)__builtin_amdgcn_s_memtime(); }
    return 0;
  };
  const long long t_begin = now();
  auto compute_tile = [&](int kv0, int stage, int t_issue, auto tail_tag) {
    constexpr bool MAYBE_TAIL = decltype(tail_tag)::value;
    const long long c0 = now();
    // four per-lane addresses per tile (stage base + swizzled chunk of row l31); every fragment read below is one of them
    // plus an immediate (K sub-tile +4096, V^T +8192, +12288).  Opaque to the optimiser, which otherwise re-derives an
    // address per ds_read (16 v_add3_u32 per tile).
    unsigned fa[4];
    unsigned so = (unsigned)(stage * 2 * TILE_BYTES);
    asm volatile("" : "+s"(so));  // keep the stage offset a scalar: one v_add per address, no loop-carried vector adds
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa[s] = rdo[s] + so;
      asm volatile("" : "+v"(fa[s]));
    }

    // ---- S^T = K Q^T : two 32-key sub-tiles (first MFMA of each chain takes a literal-zero C) ----
    f32x16_t st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const opx8_t kf = *(const opx8_t*)(lds + fa[s] + t * 4096);
        if (s == 0) st[t] = wvn_mfma_32x32x16(kf, qf[s], PRE ? cneg : (f32x16_t)(0.f), 0, 0, 0);
        else st[t] = wvn_mfma_32x32x16(kf, qf[s], st[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t_issue >= 0) issue(t_issue);
    // From here to the end of the tile the wave is VALU-heavy (softmax, packing, PV with its fillers); it wins issue
    // arbitration over the waves that are in their MFMA-only S^T phase, which need few issue slots (same-box A/B: -0.6 %).
    // (A second s_setprio between softmax and PV fences hipcc's interleaving of the exps with the PV MFMAs: +4 %.)
    if constexpr (PRE) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TIMING) asm volatile("s_nop 7\ns_nop 7" ::: "memory");
    const long long c1 = now();
    // ---- mask the tail of the last tile ----
    if (MAYBE_TAIL && kv0 + KVB > ntok) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= ntok) st[t][r] = -1e30f;
        }
    }
    // ---- online softmax (lane = query, both half-waves share the running max) ----
    // row max: 16 three-input max (two independent chains), then the other half-wave's
    float ma = max3f(st[0][0], st[0][1], st[0][2]), mb = max3f(st[1][0], st[1][1], st[1][2]);
    ma = max3f(ma, st[0][3], st[0][4]); mb = max3f(mb, st[1][3], st[1][4]);
#pragma unroll
    for (int r = 5; r < 15; r += 2) { ma = max3f(ma, st[0][r], st[0][r + 1]); mb = max3f(mb, st[1][r], st[1][r + 1]); }
    float mt = max3f(ma, mb, st[0][15]);
    mt = max3f(mt, st[1][15], st[1][15]);
    {  // the other half-wave's max of the same query: v_permlane32_swap (VALU) instead of a ds_bpermute LDS round trip
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = max3f(mt, __uint_as_float(sw[0]), __uint_as_float(sw[1]));  // one of the two is this lane's own value
    }
    // Deferred rescale: the running max only follows when some row's scores outgrow it by more than THR (in
    // exp2 units).  P is then bounded by 2^THR instead of 1 -- harmless in fp32 accumulators, and P / l cancel
    // exactly the same factor -- and the wave-uniform O rescale almost never runs after the first tiles.
    constexpr float THR = 6.0f;
    if constexpr (PRE) {
      // st = c s - m_run already.  The max follows when a row outgrows it by THR, or unconditionally on the first tile
      // (m_run = 0 there: rows whose scores are all far below 0 must not underflow).
      if (fresh || __any(mt > THR)) {
        const float delta = fresh ? mt : fmaxf(mt, 0.f);
        const float alpha = fresh ? 1.f : __builtin_amdgcn_exp2f(-delta);
        fresh = false;
        m_run += delta;
        l_run *= alpha;
        l_run1 *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[t][r] -= delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) cneg[r] = -m_run;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = __builtin_amdgcn_exp2f(st[t][r]);
    } else {
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
      const f32x2_t c2v = {c_exp, c_exp}, mc2 = {mc, mc};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t y = f32x2_t{st[t][r], st[t][r + 1]} * c2v + mc2;  // v_pk_fma_f32
          st[t][r] = __builtin_amdgcn_exp2f(y[0]);
          st[t][r + 1] = __builtin_amdgcn_exp2f(y[1]);
        }
    }

    const long long c2 = now();
    // ---- O^T += V^T P^T : 4 groups of 16 keys; the half-wave's 8 keys of a group are one 16-byte chunk ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int t = ks >> 1, h8 = (ks & 1) * 8;
      union { u32x4_t u; opx8_t v; } pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t pk = pack_op2(st[t][h8 + 2 * e], st[t][h8 + 2 * e + 1]);
        pf.u[e] = pk;  // (bit_cast of the scalar, not of the vector element: clang reads element 0 for the latter)
        if constexpr (SUM == 0) {  // row sum of the rounded P (exactly what the PV MFMA multiplies)
          if (e & 1) l_run1 = dot2_ones_op(pk, l_run1);
          else l_run = dot2_ones_op(pk, l_run);
        } else {  // plain adds on the fp32 P.  NOT inline asm: its operands are v_exp results, and hipcc pads the
                  // transcendental-result hazard only for instructions it emits itself (an asm v_add read stale values)
          l_run += st[t][h8 + 2 * e];
          l_run1 += st[t][h8 + 2 * e + 1];
        }
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const opx8_t vf = *(const opx8_t*)(lds + fa[ks] + TILE_BYTES + dt * 4096);
        ot[dt] = wvn_mfma_32x32x16(vf, pf.v, ot[dt], 0, 0, 0);
      }
    }
    if constexpr (PRE) __builtin_amdgcn_s_setprio(0);
    if constexpr (TIMING) {
      const long long c3 = now();
      tm[1] += c1 - c0; tm[2] += c2 - c1; tm[3] += c3 - c2;
    }
  };

  // LAZY (pre-scaled q only): the same online softmax WITHOUT the per-tile row max.  The kernel is bound by VALU issue
  // (scripts/ubench/mfma_rate.hip: the 2 exp2 + max3 + cvt_pk + dot2c that go with every MFMA cost 60 cycles per MFMA in one wave
  // against 36 without), and the 16 v_max3 + lane swap per tile only serve a decision that is almost always "no".  So: exponentiate
  // against the running max as it stands, pack, sum -- the row sums are needed anyway -- and let the SUMS raise the alarm: a
  // probability above 2^16 (a score 16 exp2-units above the running max) makes its lane's partial sum exceed 2^16, overflow to inf
  // included.  Only then (and on the first tile) the tile is redone the exact way: S^T again from the K tile still in LDS, exact row
  // max, rescale of O / l / the C operand, exponentials again.  Undetected probabilities are <= 2^16 instead of <= 2^6: harmless in
  // fp32 / bf16 floating point, P and l carry the same factor.
  auto compute_tile_lazy = [&](int kv0, int stage, int t_issue, auto tail_tag) {
    constexpr bool MAYBE_TAIL = decltype(tail_tag)::value;
    unsigned fa[4];
    unsigned so = (unsigned)(stage * 2 * TILE_BYTES);
    asm volatile("" : "+s"(so));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa[s] = rdo[s] + so;
      asm volatile("" : "+v"(fa[s]));
    }
    f32x16_t st[2];
    auto qk = [&]() {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        {
          i32x8_t k8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const opx8_t kf = *(const opx8_t*)(lds + fa[s] + t * 4096);
            st[t] = wvn_mfma_32x32x16(kf, qf[s], s == 0 ? cneg : st[t], 0, 0, 0);
            if constexpr (QSPLIT == 1) st[t] = wvn_mfma_32x32x16(kf, ql[s], st[t], 0, 0, 0);
            if constexpr (QSPLIT == 2) {   // the top bytes of the eight fp16 values = their e5m2 image, truncated
              const u32x4_t kr = __builtin_bit_cast(u32x4_t, kf);
              k8[2 * s] = (int)__builtin_amdgcn_perm(kr[1], kr[0], 0x07050301u);
              k8[2 * s + 1] = (int)__builtin_amdgcn_perm(kr[3], kr[2], 0x07050301u);
            }
          }
#if WVN_OPERAND_F16
          if constexpr (QSPLIT == 2) st[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8, ql8, st[t], 1, 1, 0, 0x7f7f7f7f, 0, 0x73737373);
#endif
        }
      if (MAYBE_TAIL && kv0 + KVB > ntok) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= ntok) st[t][r] = -1e30f;
          }
      }
    };
    u32x4_t pfu[4];
    float s0, s1;
    auto expsum = [&]() {   // P = exp2(st) packed to bf16 (the PV operand fragments) and the lane's partial row sums of the rounded P
      s0 = 0.f; s1 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int t = ks >> 1, h8 = (ks & 1) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float e0 = __builtin_amdgcn_exp2f(st[t][h8 + 2 * e]), e1 = __builtin_amdgcn_exp2f(st[t][h8 + 2 * e + 1]);
          const uint32_t pk = pack_op2(e0, e1);
          pfu[ks][e] = pk;
          if constexpr (SUM == 0) {   // row sums of the ROUNDED P: one v_dot2c per pair
            if (e & 1) s1 = dot2_ones_op(pk, s1);
            else s0 = dot2_ones_op(pk, s0);
          } else {                    // plain fp32 adds on the unrounded P (this file is compiled with -fno-slp-vectorize: packed
            s0 += e0;                 // v_pk_add_f32 beside MFMAs costs more than two scalar adds, MI355X_MICROARCH.md)
            s1 += e1;
          }
        }
      }
    };
    qk();
    __builtin_amdgcn_sched_barrier(0);
    if (t_issue >= 0) issue(t_issue);
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_sched_barrier(0);
    // fp16 operands: a probability must stay below 65504, so the alarm rings at 2^12 (a lane's partial sum of 16 values
    // <= 2^12 each cannot reach fp16's infinity unnoticed: anything above 2^12 already fires)
    constexpr float ALARM = WVN_OPERAND_F16 ? 4096.f : 65536.f;
    if (!fresh) expsum();
    if (fresh || __any(s0 + s1 > ALARM)) {
      if (!fresh) qk();   // the exponentials overwrote the scores
      float ma = max3f(st[0][0], st[0][1], st[0][2]), mb = max3f(st[1][0], st[1][1], st[1][2]);
      ma = max3f(ma, st[0][3], st[0][4]); mb = max3f(mb, st[1][3], st[1][4]);
#pragma unroll
      for (int r = 5; r < 15; r += 2) { ma = max3f(ma, st[0][r], st[0][r + 1]); mb = max3f(mb, st[1][r], st[1][r + 1]); }
      float mt = max3f(ma, mb, st[0][15]);
      mt = fmaxf(mt, st[1][15]);
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = max3f(mt, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
      const float delta = fresh ? mt : fmaxf(mt, 0.f);
      const float alpha = fresh ? 1.f : __builtin_amdgcn_exp2f(-delta);
      fresh = false;
      m_run += delta;
      l_run *= alpha;
      l_run1 *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] -= delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) cneg[r] = -m_run;
      expsum();
    }
    l_run += s0;
    l_run1 += s1;
    __builtin_amdgcn_sched_barrier(0);
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      union { u32x4_t u; opx8_t v; } pf;
      pf.u = pfu[ks];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const opx8_t vf = *(const opx8_t*)(lds + fa[ks] + TILE_BYTES + dt * 4096);
        ot[dt] = wvn_mfma_32x32x16(vf, pf.v, ot[dt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // tile loop: wait for tile t (NST - 2 younger tiles may stay in flight), barrier (everyone has the tile and
  // has finished reading tile t - 1, whose slot the next DMA overwrites), issue tile t + NST - 1, compute
  // The last tile (the only one that may need masking) is peeled: with both instantiations of compute_tile inside one
  // loop body hipcc gave the O^T accumulators different registers on the two paths and copied all 32 of them twice per
  // tile (32 v_mov_b64 that also wait for the PV MFMAs to drain).
  for (int t = 0; t + 1 < nt; ++t) {
    const long long w0 = now();
    if (t + NST - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (TIMING) tm[0] += now() - w0;
    // (the next tile's DMA is requested inside compute_tile, after the QK^T MFMAs have been issued: the four DMA
    // instructions cost the wave a few hundred cycles of issue time, which then overlap the matrix pipe's work)
    if constexpr (LAZY) compute_tile_lazy(t * KVB, t % NST, t + NST - 1 < nt ? t + NST - 1 : -1, std::false_type{});
    else compute_tile(t * KVB, t % NST, t + NST - 1 < nt ? t + NST - 1 : -1, std::false_type{});
  }
  {
    const long long w0 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (TIMING) tm[0] += now() - w0;
    if constexpr (LAZY) compute_tile_lazy((nt - 1) * KVB, (nt - 1) % NST, -1, std::true_type{});
    else compute_tile((nt - 1) * KVB, (nt - 1) % NST, -1, std::true_type{});
  }

  if constexpr (TIMING) {
    if (lane == 0 && dbg) {
      long long* d = dbg + ((size_t)blockIdx.x * 4 + wave) * 5;
      d[0] = tm[0]; d[1] = tm[1]; d[2] = tm[2]; d[3] = tm[3]; d[4] = now() - t_begin;
    }
  }
  // ---- normalise and store: out[b*ntok_s + q][head*64 + d] ----
  l_run += l_run1;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q0 + l31;
  if (out_lo && out_frag == 2) {
    // (uniform) the MX operand planes of the projection (csrc/gemm_n384_x3.hip, gemm_n384_mx_pair_kernel): per element h = fp16(o) as FRAGMENT-MAJOR
    // fp16 (out: the layout of the branch below) and l8 = e5m2((o - h) * 2^12) as [row group][head = 64-k step][half dt][64 lanes][16 bytes] (out_lo; the
    // consumer derives h8 = e5m2(h) itself) -- byte (u, j) of half dt = column 16 (2 dt + u) + swap23(8 hi + j) of the head,
    // exactly what this lane's accumulators hold.  The padding queries of a frame (ntok <= q < ntok_s) are written as ZEROS: stale bytes
    // reinterpreted as e5m2 would be NaN / inf one time in 32, and the padding rows feed the next block's (masked, but multiplied) K / V^T.
    if (qi < ntok_s) {
      typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
      typedef __attribute__((ext_vector_type(2))) short s16x2_t;
      const float keep = qi < ntok ? inv : 0.f;
      const size_t m = (size_t)b * ntok_s + qi;
      const int nks = heads * 4;
      op16_t* fb = out + (((m >> 5) * nks + head * 4) * 64 + hi * 32 + (m & 31)) * 8;
      unsigned char* l8 = (unsigned char*)out_lo + (((m >> 5) * heads + head) * 2 * 64 + hi * 32 + (m & 31)) * 16;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x4_t o8l = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u32x4_t oh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * (2 * u + (e >> 1)) + 2 * (e & 1);
            const float a = ot[dt][r] * keep, c = ot[dt][r + 1] * keep;
            const uint32_t hb = pack_f16x2(a, c);
            oh[e] = hb;
            const h2_t hh = __builtin_bit_cast(h2_t, hb);
            const int d = 2 * u + (e >> 1);
            // (the old value travels as a SCALAR: clang's bit_cast of a vector element reads element 0)
            const uint32_t ol8 = o8l[d];
            if (e & 1) o8l[d] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(__builtin_bit_cast(s16x2_t, ol8), a - (float)hh[0], c - (float)hh[1], 1.0f / 4096.0f, true));
            else o8l[d] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(__builtin_bit_cast(s16x2_t, ol8), a - (float)hh[0], c - (float)hh[1], 1.0f / 4096.0f, false));
          }
          *(u32x4_t*)(fb + (2 * dt + u) * 512) = oh;
        }
        *(u32x4_t*)(l8 + dt * 1024) = o8l;
      }
    }
  } else if (qi < ntok) {
    const size_t oo = ((size_t)b * ntok_s + qi) * (heads * DH) + head * DH;
    op16_t* og = out + oo;
    if (out_lo && out_frag) {
      // (uniform) hi / lo bf16 planes, FRAGMENT-MAJOR: for the 32-row group R = m >> 5 of the flat token row m and k-step s = 4 head + 2 dt + u
      // of the projection, one 1 KB block per plane at ((R * 24 + s) * 1024), lane-major (16 bytes at ((hi * 32 + (m & 31)) * 16): the
      // eight values whose column indices are 16 s + swap23(8 hi + j) -- exactly what this lane's accumulators hold (the layout of
      // gemm_a384_x3.hip's EPI_GELU_FRAG; the projection's weight carries the same column permutation)
      const size_t m = (size_t)b * ntok_s + qi;
      const int nks = heads * 4;
      op16_t* fb = out + (((m >> 5) * nks + head * 4) * 64 + hi * 32 + (m & 31)) * 8;
      op16_t* fl = out_lo + (((m >> 5) * nks + head * 4) * 64 + hi * 32 + (m & 31)) * 8;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u32x4_t oh, olo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * (2 * u + (e >> 1)) + 2 * (e & 1);      // g = 2 u + (e >> 1), elements 2 (e & 1), + 1
            const float a = ot[dt][r] * inv, c = ot[dt][r + 1] * inv;
            oh[e] = pack_bf16x2(a, c);
            olo[e] = pack_bf16x2(a - __uint_as_float(oh[e] << 16), c - __uint_as_float(oh[e] & 0xffff0000u));
          }
          *(u32x4_t*)(fb + (2 * dt + u) * 512) = oh;
          *(u32x4_t*)(fl + (2 * dt + u) * 512) = olo;
        }
    } else if (out_lo) {   // (uniform) hi / lo bf16 planes, row-major
      op16_t* ol = out_lo + oo;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2_t oh, olo;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float a = ot[dt][4 * g + 2 * e] * inv, c = ot[dt][4 * g + 2 * e + 1] * inv;
            oh[e] = pack_bf16x2(a, c);
            olo[e] = pack_bf16x2(a - __uint_as_float(oh[e] << 16), c - __uint_as_float(oh[e] & 0xffff0000u));
          }
          *(u32x2_t*)(og + dt * 32 + 8 * g + 4 * hi) = oh;
          *(u32x2_t*)(ol + dt * 32 + 8 * g + 4 * hi) = olo;
        }
    } else {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t o;
        o[0] = pack_op2(ot[dt][4 * g + 0] * inv, ot[dt][4 * g + 1] * inv);
        o[1] = pack_op2(ot[dt][4 * g + 2] * inv, ot[dt][4 * g + 3] * inv);
        *(u32x2_t*)(og + dt * 32 + 8 * g + 4 * hi) = o;
      }
    }
  }
}

long long* g_attn_dbg = nullptr;  // set by wvn_debug_attention_timing (scripts/attn_timing.py)

constexpr int ATTN_DEFAULT = 1;
int g_attn_variant = ATTN_DEFAULT;
int g_attn_qsplit_form = 2;   // two-plane q: 2 = the second plane on a scaled e5m2 MFMA (round 6, fp16 build), 1 = both planes on fp16 MFMAs (round 4)   // 0: exact per-tile row max, 1: lazy (alarm on the row sums; what ships: -4 % attention time)

void launch_pre(bool xcd, dim3 grid, hipStream_t st, const op16_t* q, const op16_t* k, const op16_t* vt, op16_t* out,
                int heads, int nbh, int nqb, int ntok, int ntok_s, int npad, op16_t* out_lo, const op16_t* q_lo, int out_frag) {
  if (q_lo && WVN_OPERAND_F16 && g_attn_qsplit_form == 2) {   // two-plane q, the second plane on ONE scaled e5m2 MFMA per sub-tile (round 6); 148 registers: three
    // workgroups per CU (forced to four -- 128 registers, 68 bytes of scratch -- it runs 14 % slower: 31.0 against 27.2 ms of attention per step)
    if (xcd)
      hipLaunchKernelGGL((attention_bf16_kernel<2, true, 3, false, true, 0, true, 2>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, q_lo, out_frag);
    else
      hipLaunchKernelGGL((attention_bf16_kernel<2, false, 3, false, true, 0, true, 2>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, q_lo, out_frag);
    return;
  }
  if (q_lo) {   // two-plane q: the lazy form with 16 more registers -- three workgroups per CU
    if (xcd)
      hipLaunchKernelGGL((attention_bf16_kernel<2, true, 3, false, true, 0, true, 1>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, q_lo, out_frag);
    else
      hipLaunchKernelGGL((attention_bf16_kernel<2, false, 3, false, true, 0, true, 1>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, q_lo, out_frag);
    return;
  }
  // 2-stage ring, 4 workgroups per CU (the pre-scaled kernel needs 128 VGPRs: 11.96 ms per step against 12.28 for 3 stages /
  // 3 workgroups and 13.2 for 4 stages / 2); row sums by v_dot2c_f32_bf16 on the packed P (plain fp32 adds, which hipcc packs
  // into v_pk_add_f32, measured 12.0 ms per step against 11.7).  The same arithmetic with and without the XCD block order:
  // results must not depend on the batch size (tests/test_gpu_attention_xcd.py).
  if (g_attn_variant == 2) {   // lazy max, row sums by scalar fp32 adds
    if (xcd)
      hipLaunchKernelGGL((attention_bf16_kernel<2, true, 4, false, true, 1, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, nullptr, out_frag);
    else
      hipLaunchKernelGGL((attention_bf16_kernel<2, false, 4, false, true, 1, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, nullptr, out_frag);
    return;
  }
  if (g_attn_variant == 1) {
    if (xcd)
      hipLaunchKernelGGL((attention_bf16_kernel<2, true, 4, false, true, 0, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, nullptr, out_frag);
    else
      hipLaunchKernelGGL((attention_bf16_kernel<2, false, 4, false, true, 0, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                         nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, nullptr, out_frag);
    return;
  }
  if (xcd)
    hipLaunchKernelGGL((attention_bf16_kernel<2, true, 4, false, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                       nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, nullptr, out_frag);
  else
    hipLaunchKernelGGL((attention_bf16_kernel<2, false, 4, false, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh,
                       nqb, ntok, ntok_s, npad, 1.f, nullptr, out_lo, nullptr, out_frag);
}

template <int NST, int OCC>
void launch_v(bool xcd, dim3 grid, hipStream_t st, const op16_t* q, const op16_t* k, const op16_t* vt, op16_t* out,
              int heads, int nbh, int nqb, int ntok, int ntok_s, int npad, float c_exp, op16_t* out_lo) {
  if (g_attn_dbg) {
    hipLaunchKernelGGL((attention_bf16_kernel<NST, true, OCC, true>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh, nqb,
                       ntok, ntok_s, npad, c_exp, g_attn_dbg, out_lo, nullptr, 0);
    return;
  }
  if (xcd)
    hipLaunchKernelGGL((attention_bf16_kernel<NST, true, OCC>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh, nqb, ntok,
                       ntok_s, npad, c_exp, nullptr, out_lo, nullptr, 0);
  else
    hipLaunchKernelGGL((attention_bf16_kernel<NST, false, OCC>), grid, dim3(256), 0, st, q, k, vt, out, heads, nbh, nqb, ntok,
                       ntok_s, npad, c_exp, nullptr, out_lo, nullptr, 0);
}

}  // namespace

void WVN_OPSYM(wvn_attention_bf16_set_debug)(long long* dbg) { g_attn_dbg = dbg; }
void WVN_OPSYM(wvn_attention_bf16_set_variant)(int v) {
  if (v >= 16) { g_attn_qsplit_form = v - 16; return; }   // 17 / 18: the form of the two-plane q (wvn_debug_attention_variant)
  g_attn_variant = v < 0 ? ATTN_DEFAULT : v;
}  // < 0: back to the default

// scale > 0: q holds the raw projections.  scale == 0: q is pre-multiplied by softmax_scale * log2(e) (EPI_QKV with
// q_scale set), the kernel with the running max folded into the S^T MFMA chain runs.
int WVN_OPSYM(wvn_attention_bf16_launch)(const op16_t* q, const op16_t* k, const op16_t* vt, op16_t* out, int B, int heads,
                              int ntok, int ntok_s, int npad, float scale, hipStream_t st, op16_t* out_lo, const op16_t* q_lo, int out_frag) {
  // out_lo (WVN_PREC_MIX): the normalised output leaves as two bf16 planes, hi = bf16(o) -> out, lo = bf16(o - hi) -> out_lo (the
  // operand representation of the exact-mode projection GEMM), straight from the fp32 accumulators -- whatever this build's own
  // operand format is
  if (!q || !k || !vt || !out || npad % QB != 0 || npad < ntok) return WVN_ERR_ARG;
  const int nqb = ceil_div(ntok, QB), nbh = B * heads;
  if ((size_t)nbh * npad * DH * 2 >= (1ull << 32)) return WVN_ERR_ARG;  // 32-bit buffer offsets
  const float c_exp = scale * 1.44269504088896340736f;
  dim3 grid(nqb * nbh);
  const bool xcd = (nbh % 8) == 0;  // the XCD decode needs whole groups of 8 (frame, head) pairs
  if (scale == 0.f && !g_attn_dbg) {
    launch_pre(xcd, grid, st, q, k, vt, out, heads, nbh, nqb, ntok, ntok_s, npad, out_lo, q_lo, out_frag);
    WVN_LAUNCH_CHECK();
    return WVN_OK;
  }
  if (scale == 0.f || q_lo) return WVN_ERR_ARG;
  launch_v<3, 3>(xcd, grid, st, q, k, vt, out, heads, nbh, nqb, ntok, ntok_s, npad, c_exp, out_lo);  // raw-q form: 3-stage ring, 3 workgroups / CU
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
