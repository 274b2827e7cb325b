// LayerNorm + QKV projection of a ViT block in one kernel (D = 384, gfx950):
//
//   xn = LayerNorm(x) ;  [q | k | v] = xn W^T + b ;  q *= softmax_scale * log2(e)
//   q, k -> [B*h][npad][64] bf16,  v -> V^T [B*h][64][npad] bf16 with the token order of attention_bf16.hip
//
// (blocks.i.norm1 + blocks.i.attn.qkv; the un-fused pair is the LayerNorm kernel -- 310 MB read, 155 MB written -- and
// gemm_a384.hip, which reads those 155 MB back).  Built like mlp_fused.hip, on the same two measurements (scripts/ubench):
//   * a workgroup is 4 waves = one wave per SIMD with the whole register file.  A wave owns 64 token rows: it normalises them once
//     per row block (a lane holds half a row, its partner lane ^ 32 the other half) and keeps them as 48 MFMA operand fragments
//     (192 registers) for all 18 column tiles;
//   * W ([1152][384]) streams through an LDS ring of 16 KB slices [64 n][128 k] by direct-to-LDS DMA; every fragment read from it
//     feeds TWO MFMAs (the wave's two 32-row sub-tiles) -- half the LDS traffic and half the DMA per MFMA of the 32-rows-per-wave
//     kernels, which is what bounds those;
//   * one s_barrier per slice, placed in the MIDDLE of the previous slice so that the fragment stream never stops at a slice
//     boundary;
//   * q / k column tiles run transposed (lane = token, registers = 4 consecutive features: rows of 128 B leave through a
//     wave-private LDS image), v tiles straight (lane = feature d, registers = 4 consecutive tokens: V^T rows).  Same epilogue
//     arithmetic as gemm_a384.hip: accumulators start at the bias, q is scaled before its bf16 rounding.
// Load balance: 788 row blocks of 256 rows on 256 CUs are 3.08 rounds; the 20 row blocks of the last, thin round are split by
// column tiles over all workgroups (each re-normalises its rows: cheap next to idling 236 CUs for a whole round).
#include <type_traits>

#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int KD = 384;            // model dim (K)
constexpr int BNT = 64;            // columns per tile = one head of q, k or v
constexpr int SLICE = 16384;       // ring slice bytes: [64 n][128 k]
constexpr int NS = 5;              // ring depth
constexpr int RING = NS * SLICE;   // 81,920
constexpr int RW = 64;             // rows per wave
constexpr int BM = 4 * RW;         // rows per workgroup
constexpr int STG_ROW = 144;       // staged bf16 row: 64 values + 16 B pad
constexpr int STG_BYTES = 64 * STG_ROW;               // 9,216 per wave
constexpr int STG_OFF = RING;
constexpr int TAB_OFF = STG_OFF + 4 * STG_BYTES;      // 118,784: bias [N], then LayerNorm gamma [384], beta [384]

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

struct QkvFusedParams {
  const float* X; int ldx;       // residual stream [M][384] fp32
  const op16_t* XN;              // PRE: the rows already normalised, as operand fragments (layout below); X / ln_* unused
  const float *ln_g, *ln_b; float ln_eps;
  const op16_t* W;               // [3 * heads * 64][384]
  const float* bias;             // [3 * heads * 64]
  op16_t* base; unsigned q_off, k_off, v_off, bytes;   // one buffer descriptor over q / k / v^T
  int heads, npad, ntok_s;
  float q_scale;
  int M;
  long long* dbg;   // TIMING: per wave {LayerNorm prologue, slices, epilogues, total} in shader cycles
};

// (see mlp_fused.hip: a buffer_store_dwordx4 with an SGPR soffset must not have its data registers overwritten in the next two
// issue slots on gfx950; LLVM does not know)
__device__ inline void store_b128_guarded(u32x4_t v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

// PRE: the LayerNorm has been applied by the kernel that produced the rows (mlp_fused.hip, resident form with NXT), which left them
// as ready-made operand fragments: fragment (32-row group R, k-step s) = 1 KB at XN + (R * 24 + s) * 1024, lane l's 16 bytes at
// l * 16 -- one fully coalesced load per fragment, no LDS pass, no statistics, half the bytes of the fp32 rows.  The k-slots of a
// lane are the ones the accumulator layout hands over (columns 16 s + {4 hi .. + 3, 8 + 4 hi .. + 3}), so W is the copy of
// qkv.weight with bits 2 and 3 of its column index swapped.
template <bool TIMING, bool PRE = false>
__global__ __launch_bounds__(256, 1) void qkv_fused_kernel(QkvFusedParams p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int N = 3 * p.heads * 64, NT = N / BNT, NQK = 2 * p.heads;   // column tiles; the first NQK are q / k tiles
  const int nrb = (p.M + BM - 1) / BM, G = gridDim.x;
  // ---- this workgroup's segments: nfull whole row blocks, then (perhaps) a few column tiles of a row block of the thin round ----
  // (Tried and dropped, scripts/bench_qkv_fused.py: the LayerNorm prologues of all workgroups run at the same time and are
  //  HBM-bound, 39 % of the kernel.  Shifting the workgroups a quarter period apart made each prologue faster and the MFMA slices
  //  60 % slower -- their epilogue stores then queue behind the other groups' row reads -- for a net loss; software prefetch of
  //  the next rows does not help either, a round of row blocks is three times the L2.)
  const int nfull = nrb / G, left = nrb - nfull * G;         // left < G row blocks remain after the whole rounds
  const int parts = left > 0 ? G / left : 1;                 // each of them is split into `parts` column ranges
  const bool has_tail = left > 0 && (int)blockIdx.x < left * parts;
  const int tail_rb = nfull * G + (int)blockIdx.x / parts, tail_part = (int)blockIdx.x % parts;
  const int tail_t0 = tail_part * NT / parts, tail_t1 = (tail_part + 1) * NT / parts;
  const int nseg = nfull + (has_tail && tail_t1 > tail_t0 ? 1 : 0);
  auto seg_rb = [&](int s) { return s < nfull ? (int)blockIdx.x + s * G : tail_rb; };
  auto seg_t0 = [&](int s) { return s < nfull ? 0 : tail_t0; };
  auto seg_t1 = [&](int s) { return s < nfull ? NT : tail_t1; };
  const int total = 3 * (nfull * NT + (nseg > nfull ? tail_t1 - tail_t0 : 0));   // ring slices this workgroup consumes

  float* bias_l = (float*)(smem + TAB_OFF);
  float* lng_l = bias_l + N;
  for (int i = tid; i < N; i += 256) bias_l[i] = p.bias ? p.bias[i] : 0.f;
  if constexpr (!PRE)
    for (int i = tid; i < KD; i += 256) { lng_l[i] = p.ln_g[i]; lng_l[KD + i] = p.ln_b[i]; }
  if (total == 0) return;
  const __amdgpu_buffer_rsrc_t rs_xn = __builtin_amdgcn_make_buffer_rsrc((void*)(PRE ? p.XN : p.W), 0, (unsigned)((size_t)((p.M + 31) / 32) * 24 * 1024), 0x00020000);

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)N * KD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.base, 0, p.bytes, 0x00020000);
  // ---- W slice DMA: 16 wave-instructions of 1 KB (4 rows of 256 B), 4 per wave; chunk XOR (row & 15) (gemm_a384.hip) ----
  unsigned woff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = (wave * 4 + u) * 4 + (lane >> 4);
    woff[u] = (unsigned)((r * KD + (((lane & 15) ^ (r & 15)) * 8)) * 2);
  }
  // issue stream: slices in consumption order (segment, tile, k-slice), running ahead of the MFMAs
  int is_seg = 0, is_tile = seg_t0(0), is_ks = 0, is_slot = 0, issued = 0;
  auto issue_next = [&]() {
    unsigned char* dst = smem + is_slot * SLICE + wave * 4096;
    const unsigned soff = __builtin_amdgcn_readfirstlane((is_tile * BNT * KD + is_ks * 128) * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, woff[u], soff, 0, 0);
    ++issued;
    is_slot = is_slot == NS - 1 ? 0 : is_slot + 1;
    if (++is_ks == 3) {
      is_ks = 0;
      if (++is_tile == seg_t1(is_seg)) { ++is_seg; is_tile = is_seg < nseg ? seg_t0(is_seg) : 0; }
    }
  };
  static_assert(NS == 5, "the waits below are written for a ring of 5");
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (issued < total) issue_next();

  // fragment addressing: fragment i (0..15) of the slice in ring slot `slot`: k-step i >> 1, 32-column sub-tile i & 1
  unsigned off[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) off[s] = l31 * 256 + (((2 * s + hi) ^ (l31 & 15)) << 4);
  auto frag = [&](int slot, int i) -> opx8_t { return *(const opx8_t*)(smem + slot * SLICE + off[i >> 1] + (i & 1) * 8192); };
  unsigned char* stg = smem + STG_OFF + wave * STG_BYTES;

  int si = 0, rslot = 0;   // slice being multiplied (workgroup-local index) and its ring slot
  bool stored = false;     // a tile epilogue has run (wave-uniform)
  int pf_slack = 0;        // PRE: open_next calls whose slice was issued BEFORE the latest prefetch (see prefetch below)
  // "slice si + 1 is readable" (runs in the middle of slice si; see mlp_fused.hip open_next).  The wave's VM queue behind the DMA
  // of slice si + 1 (issued three slices ago): the DMAs of si + 2 and si + 3 (4 each) and -- tile epilogues come every three
  // slices -- the 8 stores of exactly one epilogue, once there has been one.  Vector memory operations retire in order (the
  // counted waits of gemm_a384.hip rest on the same fact), so "at most that many outstanding" means the slice has landed;
  // a LayerNorm prologue in between drains the queue altogether.
  auto open_next = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (si + 1 < total) {
      if (si + 3 < total) {
        if (pf_slack > 0) {   // PRE: the NPF = 36 prefetch loads of the next segment sit behind slice si + 1 in the queue, too
          --pf_slack;
          if (stored) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
        } else if (stored) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (issued < total) issue_next();
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto shape8 = [&]() {   // scheduling shape of half a slice: 8 x (one fragment read, two MFMAs)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  __syncthreads();   // tables visible
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (total < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (issued < total) issue_next();
  opx8_t wf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wf[i] = frag(0, i);

  long long tm[4] = {0, 0, 0, 0};
  auto now = [&]() -> long long {
    if constexpr (TIMING) { __builtin_amdgcn_sched_barrier(0); return (long long)__builtin_amdgcn_s_memtime(); }
    return 0;
  };
  const long long t_begin = now();
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NPF = PRE ? 36 : 1;   // PRE: fragments of the next segment fetched ahead (k-steps 0 .. 17 of both row sub-tiles: 144 registers)
  u32x4_t pf[NPF];
  for (int s = 0; s < nseg; ++s) {
    const int m0w = seg_rb(s) * BM + wave * RW;
    const long long c_ln0 = now();
    // ---- LayerNorm of the wave's 64 rows -> MFMA operand fragments xf[row sub-tile][k-step] (row l31, k = 16 s + 8 hi .. + 7) ----
    opx8_t xf[2][KD / 16];
    int dep = 0;   // serialises the two 32-row passes (384 fp32 values in flight at once would not fit the register file)
    if constexpr (PRE) {
      // fragment (rs, k) of the segment's rows; in the order the slices consume them (k-major): loads retire in order.  From the
      // second segment on, the first NPF of the 48 are already here: they were fetched while the previous segment was being multiplied
      // (all 256 workgroups pulling 192 KB each at the same moment is a 50 MB burst at HBM speed: 19 % of the kernel without this)
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        const int k = i >> 1, rs = i & 1;
        union { u32x4_t u; opx8_t v; } a;
        if (i < NPF && s > 0) a.u = pf[i < NPF ? i : 0];
        else {
          const unsigned so = __builtin_amdgcn_readfirstlane(((m0w >> 5) + rs) * 24 + k) * 1024u;   // groups past M: zeros (bounds)
          a.u = __builtin_amdgcn_raw_buffer_load_b128(rs_xn, lane * 16, so, 0);
        }
        xf[rs][k] = a.v;
      }
    } else
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      // The rows come in COALESCED (an instruction = 4 rows x 256 contiguous bytes = 8 whole lines) and are turned into the fragment
      // layout (lane = row) through the wave's LDS image, 32 rows x 64 columns at a time.  Loading them in the fragment layout
      // directly (32 rows x 32 bytes per instruction: 32 lines touched for 1 KB) made this prologue 38 % of the kernel: the L1
      // takes lines, not bytes (gemm_proj.hip measured the same thing on the residual).
      f32x4_t xq[KD / 8];
      {
        const int lr = lane >> 4, lc = (lane & 15) * 4;            // loader: row 4 i + lr, columns lc .. lc + 3 of the chunk
        const float* xg = p.X + (size_t)dep + lc;
        float* stgf = (float*)stg;                                  // [32][68] fp32 (8,704 B of the 9,216)
        f32x4_t buf[2][8];
        auto load_chunk = [&](int c, f32x4_t (&b)[8]) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            b[i] = *(const f32x4_t*)(xg + (size_t)min(m0w + rs * 32 + 4 * i + lr, p.M - 1) * p.ldx + 64 * c);   // rows past M: clamped, never stored
        };
        load_chunk(0, buf[0]);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (c + 1 < 6) load_chunk(c + 1, buf[(c + 1) & 1]);
#pragma unroll
          for (int i = 0; i < 8; ++i) *(f32x4_t*)(stgf + (4 * i + lr) * 68 + lc) = buf[c & 1][i];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            xq[2 * (4 * c + kk)] = *(const f32x4_t*)(stgf + l31 * 68 + 16 * kk + 8 * hi);
            xq[2 * (4 * c + kk) + 1] = *(const f32x4_t*)(stgf + l31 * 68 + 16 * kk + 8 * hi + 4);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < KD / 8; ++i) sm += (xq[i][0] + xq[i][1]) + (xq[i][2] + xq[i][3]);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm / 384.f;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < KD / 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xq[i][e] - mean; q += d * d; }
      q += __shfl_xor(q, 32, 64);
      const float rstd = 1.0f / sqrtf(q / 384.f + p.ln_eps);
#pragma unroll
      for (int k = 0; k < KD / 16; ++k) {
        union { u32x4_t u; opx8_t v; } o;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4_t g4 = *(const f32x4_t*)(lng_l + 16 * k + 8 * hi + 4 * h2);
          const f32x4_t b4 = *(const f32x4_t*)(lng_l + KD + 16 * k + 8 * hi + 4 * h2);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (xq[2 * k + h2][e] - mean) * rstd * g4[e] + b4[e];
          o.u[2 * h2] = pack_op2(y[0], y[1]);
          o.u[2 * h2 + 1] = pack_op2(y[2], y[3]);
        }
        // park the fragment in accumulation registers right away (MFMA operands may live there): left to itself the register
        // allocator keeps it next to the 192 fp32 row values in the architectural half of the file and spills both
        union { u32x4_t u; opx8_t v; } a;
#pragma unroll
        for (int e = 0; e < 4; ++e) { uint32_t t; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(t) : "v"(o.u[e])); a.u[e] = t; }
        xf[rs][k] = a.v;
      }
      {
        union { opx8_t v; u32x4_t u; } last;
        last.v = xf[rs][KD / 16 - 1];
        asm volatile("" : "+v"(dep) : "a"(last.u[3]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- destination offsets of the wave's rows.  Frames start at multiples of 16 rows (ntok_s % 16 == 0), so every group of 16
    // rows lies in one frame.  q / k: row it*8 + (lane >> 3), 16 B at column (lane & 7) * 8 of the head; group it >> 1.
    // v^T: feature d = it*8 + (lane >> 3), 8 tokens (16 B) at row m0w + (lane & 7) * 8. ----
    if constexpr (TIMING) tm[0] += now() - c_ln0;
    unsigned voff[4], vt_off;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = m0w + g * 16 + (lane >> 3);
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      voff[g] = m < p.M ? (unsigned)((((size_t)b * p.heads * p.npad + tk) * 64 + (lane & 7) * 8) * 2) : OOB;
    }
    {
      const int m = m0w + (lane & 7) * 8;
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      vt_off = m < p.M ? (unsigned)((((size_t)b * p.heads * 64 + (lane >> 3)) * p.npad + tk) * 2) : OOB;
    }

    // one column tile: 3 slices of 16 fragments, each feeding the two row sub-tiles; then its epilogue
    auto tile = [&](int j, auto tr_tag) {
      constexpr bool TR = decltype(tr_tag)::value;
      const int n0 = j * BNT;
      f32x16_t acc[2][2];   // [row sub-tile][column sub-tile], starting at the bias
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if constexpr (TR) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4_t b4 = *(const f32x4_t*)(bias_l + n0 + 32 * t + 8 * g + 4 * hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[0][t][4 * g + e] = b4[e]; acc[1][t][4 * g + e] = b4[e]; }
          }
        } else {
          const float b = bias_l[n0 + 32 * t + l31];
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[0][t][r] = b; acc[1][t][r] = b; }
        }
      }
      const long long c_s0 = now();
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int nslot = rslot == NS - 1 ? 0 : rslot + 1;
        auto step = [&](int i) {
#pragma unroll
          for (int rs = 0; rs < 2; ++rs) {
            if constexpr (TR) acc[rs][i & 1] = wvn_mfma_32x32x16(wf[i & 3], xf[rs][ks * 8 + (i >> 1)], acc[rs][i & 1], 0, 0, 0);
            else acc[rs][i & 1] = wvn_mfma_32x32x16(xf[rs][ks * 8 + (i >> 1)], wf[i & 3], acc[rs][i & 1], 0, 0, 0);
          }
          wf[i & 3] = i + 4 < 16 ? frag(rslot, i + 4) : frag(nslot, i + 4 - 16);
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) step(i);
        shape8();
        open_next();
#pragma unroll
        for (int i = 8; i < 16; ++i) step(i);
        shape8();
        ++si;
        rslot = nslot;
      }
      const long long c_e0 = now();
      if constexpr (TIMING) tm[1] += c_e0 - c_s0;
      // ---- epilogue: bf16 through the wave's LDS image, whole 128-byte rows out ----
      if constexpr (TR) {   // q / k tile: image [64 token rows][64 features]
        const float qs = n0 < p.heads * 64 ? p.q_scale : 1.f;
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x2_t a = f32x2_t{acc[rs][t][4 * g + 0], acc[rs][t][4 * g + 1]} * f32x2_t{qs, qs};
              const f32x2_t b = f32x2_t{acc[rs][t][4 * g + 2], acc[rs][t][4 * g + 3]} * f32x2_t{qs, qs};
              const u32x2_t o = {pack_op2(a[0], a[1]), pack_op2(b[0], b[1])};
              *(u32x2_t*)(stg + (rs * 32 + l31) * STG_ROW + (32 * t + 8 * g + 4 * hi) * 2) = o;
            }
        const int D = p.heads * 64;
        const int which = n0 / D, head = (n0 - which * D) >> 6;
        const unsigned so = __builtin_amdgcn_readfirstlane((which == 0 ? p.q_off : p.k_off) + head * p.npad * 64 * 2);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const u32x4_t val = *(const u32x4_t*)(stg + ((lane >> 3) + it * 8) * STG_ROW + (lane & 7) * 16);
          store_b128_guarded(val, rs_c, voff[it >> 1], so + (it & 1) * 1024);
        }
      } else {              // v tile: image [64 features d][64 tokens], tokens of every aligned 16 with bits 2 and 3 swapped
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              // tokens 32 rs + 8 g + 4 hi + e of the wave's 64 -> stored position 32 rs + 16 (g >> 1) + 8 hi + 4 (g & 1) + e
              const int mloc = 32 * rs + 16 * (g >> 1) + 8 * hi + 4 * (g & 1);
              const u32x2_t o = {pack_op2(acc[rs][t][4 * g + 0], acc[rs][t][4 * g + 1]), pack_op2(acc[rs][t][4 * g + 2], acc[rs][t][4 * g + 3])};
              *(u32x2_t*)(stg + (32 * t + l31) * STG_ROW + mloc * 2) = o;
            }
        const int head = (n0 - 2 * p.heads * 64) >> 6;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const u32x4_t val = *(const u32x4_t*)(stg + ((lane >> 3) + it * 8) * STG_ROW + (lane & 7) * 16);
          const unsigned so = __builtin_amdgcn_readfirstlane(p.v_off + (head * 64 + it * 8) * p.npad * 2);
          store_b128_guarded(val, rs_c, vt_off, so);
        }
      }
      stored = true;
      if constexpr (TIMING) tm[2] += now() - c_e0;
      __builtin_amdgcn_sched_barrier(0);
    };
    const int t0 = seg_t0(s), t1 = seg_t1(s);
    bool pf_done = false;
    auto prefetch = [&](int j) {
      if constexpr (PRE) {
        // Once per segment, after a tile that differs from workgroup to workgroup: 256 workgroups fetching together are a 37 MB
        // burst that takes ~10 us, and vector memory operations retire in order -- the weight slices issued after the prefetch
        // cannot be used before it has landed; spread over the tiles, a prefetch is back within the three slices that are in flight
        // ahead of it.  (No branch around the loads themselves: past the last segment they re-read this segment's fragments.)
        if (j != t0 + (int)((blockIdx.x * 5u) % 12u) && !(j == t1 - 1 && !pf_done)) return;
        if (pf_done) return;
        pf_done = true;
        pf_slack = 3;
        const int m0n = seg_rb(s + 1 < nseg ? s + 1 : s) * BM + wave * RW;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const unsigned so = __builtin_amdgcn_readfirstlane(((m0n >> 5) + (i & 1)) * 24 + (i >> 1)) * 1024u;
          pf[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_xn, lane * 16, so, 0);
        }
      }
    };
    for (int j = t0; j < t1 && j < NQK; ++j) { tile(j, std::true_type{}); prefetch(j); }
    for (int j = t0 > NQK ? t0 : NQK; j < t1; ++j) { tile(j, std::false_type{}); prefetch(j); }
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
      d[0] = tm[0]; d[1] = tm[1]; d[2] = tm[2]; d[3] = now() - t_begin;
    }
  }
}

int qkv_fused_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace

// Eligibility: D == 384 (heads == 6), ntok_s % 16 == 0, M % 16 == 0, npad % 16 == 0, q / k / v^T within 2 GB of each other.
long long* WVN_OPSYM(g_qkv_fused_dbg) = nullptr;   // wvn_debug_qkv_fused_timing (scripts/bench_qkv_fused.py)

// xn_frag != nullptr: the PRE form (x, ln_* unused; W = the column-swapped copy of qkv.weight; (M + 31) / 32 * 24 KB of fragments).
int WVN_OPSYM(wvn_qkv_fused_launch)(const float* x, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const op16_t* W, const float* bias,
                         op16_t* q, op16_t* k, op16_t* vt, int heads, int npad, int ntok_s, float q_scale, int M, hipStream_t st,
                         const op16_t* xn_frag) {
  const bool pre = xn_frag != nullptr;
  if (pre) { if (((uintptr_t)xn_frag & 15) != 0) return WVN_ERR_ARG; x = (const float*)xn_frag; ldx = 4; }
  if (!x || (!pre && (!ln_g || !ln_b)) || !W || !q || !k || !vt || M <= 0 || heads * 64 != KD || (ldx % 4) != 0) return WVN_ERR_ARG;
  if ((ntok_s % 16) || (M % 16) || (npad % 16)) return WVN_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)W | (uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) != 0) return WVN_ERR_ARG;
  const uintptr_t lo = std::min({(uintptr_t)q, (uintptr_t)k, (uintptr_t)vt}), hi = std::max({(uintptr_t)q, (uintptr_t)k, (uintptr_t)vt});
  const size_t frames = (size_t)ceil_div(M, ntok_s), one = frames * heads * npad * 64 * 2;
  if (hi - lo + one >= (1ull << 31)) return WVN_ERR_ARG;
  const int N = 3 * heads * 64, lds = TAB_OFF + (N + 2 * KD) * 4;
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(160 * 1024, (const void*)qkv_fused_kernel<false>, (const void*)qkv_fused_kernel<true>, (const void*)qkv_fused_kernel<false, true>, (const void*)qkv_fused_kernel<true, true>)) return rc;
  QkvFusedParams p{};
  p.XN = xn_frag;
  p.X = x; p.ldx = ldx; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps; p.W = W; p.bias = bias;
  p.base = (op16_t*)lo; p.q_off = (unsigned)((uintptr_t)q - lo); p.k_off = (unsigned)((uintptr_t)k - lo); p.v_off = (unsigned)((uintptr_t)vt - lo);
  p.bytes = (unsigned)(hi - lo + one);
  p.heads = heads; p.npad = npad; p.ntok_s = ntok_s; p.q_scale = q_scale != 0.f ? q_scale : 1.f; p.M = M;
  const int nrb = ceil_div(M, BM), ncu = qkv_fused_num_cus();
  p.dbg = WVN_OPSYM(g_qkv_fused_dbg);
  const dim3 grid(nrb < ncu ? nrb : ncu);
  if (pre && p.dbg) hipLaunchKernelGGL((qkv_fused_kernel<true, true>), grid, dim3(256), lds, st, p);
  else if (pre) hipLaunchKernelGGL((qkv_fused_kernel<false, true>), grid, dim3(256), lds, st, p);
  else if (p.dbg) hipLaunchKernelGGL(qkv_fused_kernel<true>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(qkv_fused_kernel<false>, grid, dim3(256), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
