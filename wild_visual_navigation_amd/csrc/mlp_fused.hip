// Fused ViT MLP for D = 384 on gfx950:   x[M,384] (fp32, in place) += gelu(xn[M,384] W1^T + b1) W2^T + b2
//
// The un-fused pair (gemm_a384 fc1 -> 620 MB of bf16 hidden activations written and read back per 64 frames -> gemm_n384
// fc2) moves 1.24 GB per block that this kernel never creates: the hidden activation lives in registers for the few hundred
// cycles between being produced and being consumed.
//
//   * a workgroup is 4 waves = ONE wave per SIMD with the whole 512-entry register file; a wave owns 32 token rows for the
//     whole MLP: its xn rows as MFMA operands (96 VGPRs, loaded once), the 32 x 384 fp32 output accumulator (192), and one
//     32 x 64 tile of hidden pre-activations (32);
//   * W1 ([F][384]) and W2 ([384][F], hidden index permuted, see below) stream through ONE LDS ring of 16 KB slices written by
//     direct-to-LDS DMA: per 64 hidden units three W1 slices [64 h][128 k] and three W2 slices [128 n][64 h]; one s_barrier
//     and a counted vmcnt per slice, 16 MFMAs per wave per slice, the ring never drains (it keeps streaming across row blocks);
//   * fc1 runs in the "transposed" orientation mfma(W1 frag, xn frag): lane = token row, registers = 4 consecutive hidden
//     units -- after bias (accumulator init) and GELU those registers, packed to bf16, ARE the B operand of the fc2 MFMA
//     mfma(W2 frag, h frag): an MFMA sums over its k-slots in a fixed but arbitrary order, so operands only have to agree on
//     which hidden unit sits in which slot.  The accumulator hands lane half hi the units {4 hi .. 4 hi + 3} and
//     {8 + 4 hi .. 8 + 4 hi + 3} of every group of 16; W2 is therefore stored with bits 2 and 3 of the hidden index swapped
//     inside every aligned group of 16 (the V^T trick of attention_bf16.hip), done once when the model is packed;
//   * epilogue once per row block: accumulators -> wave-private LDS image -> (+ b2) + residual read-modify-write of whole
//     512-byte rows, exactly the association of the un-fused kernels ((acc + bias) + x).
//
// GELU: the 0.25-bf16-ulp polynomial form of gemm_a384.hip (same code, same bits as the un-fused fc1 epilogue).
#include <type_traits>
#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int KD = 384;            // model dim (K of fc1, N of fc2)
constexpr int HT = 64;             // hidden units per tile
constexpr int SLICE = 16384;       // ring slice bytes
constexpr int NS = 5;              // ring depth
#ifndef WVN_MLP_SPREAD
#define WVN_MLP_SPREAD 0   // measured: 811 us against 679 us per launch (resident form) -- a piece between MFMAs costs more than four behind a barrier
#endif
constexpr bool SPREAD = WVN_MLP_SPREAD != 0;
#ifndef WVN_MLP_STAGGER
#define WVN_MLP_STAGGER 0   // measured: 705 us against 677 us per launch -- see SPREAD
#endif
// STAGGER: the four waves issue their DMA pieces of a slice at four different points of its second half (wave h after MFMA 8 + 2 h)
// instead of all of them right behind the barrier: the CU has ONE address path (64 B/clk: 16 cycles per 1 KB piece), sixteen pieces
// arriving together keep it busy for 256 cycles, and every wave waited for most of that (in-kernel timing: 230 cycles per slice
// = 15 % of the kernel).
constexpr bool STAGGER = WVN_MLP_STAGGER != 0 && !SPREAD;
constexpr int RING = NS * SLICE;   // 81,920
constexpr int BM = 128;            // rows per workgroup (4 waves x 32)
constexpr int STG_PITCH = 132;                        // floats per staged row (128 columns + 4)
constexpr int STG_BYTES = 32 * STG_PITCH * 4;         // 16,896 per wave
constexpr int STG_OFF = RING;
constexpr int B2_OFF = STG_OFF + 4 * STG_BYTES;       // 149,504
constexpr int B1_OFF = B2_OFF + KD * 4;               // 151,040
// + F * 4 bytes of b1 (6 KB at F = 1536): 157,184 <= 163,840

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// Timing experiments (scripts/build_variant.sh ... -DWVN_MLP_EXP=<bits>; results are WRONG with any bit set): 1 no DMA after the
// prologue, 2 no barrier, 4 no fragment reads, 8 no vmcnt waits, 16 fragment reads at compile-time slot offsets -- what each
// ingredient of the slice loop costs.
#ifndef WVN_MLP_EXP
#define WVN_MLP_EXP 0
#endif

// the fc1 epilogue GELU of gemm_a384.hip (x * sigmoid(g(x)), g fitted to logit Phi(x): within 0.25 bf16 ulp of erf GELU)
#ifndef WVN_GELU_SCALAR
#define WVN_GELU_SCALAR 0   // 1: the same arithmetic on scalar VALU instructions (A/B of packed-f32 VALU beside MFMAs, scripts/ab_lib.sh)
#endif
__device__ inline float gelu_fast1(float x) {
  const float x2 = x * x;
  float t = x2 * -1.285982656e-05f + 1.435476415e-03f;
  t = t * x2 + -1.096917929e-01f;
  t = t * x2 + -2.296416554e+00f;
  const float e = __builtin_amdgcn_exp2f(t * x) + 1.f;
  return x * __builtin_amdgcn_rcpf(e);
}
__device__ inline f32x2_t gelu_fast2(f32x2_t x) {
  if constexpr (WVN_GELU_SCALAR != 0) return f32x2_t{gelu_fast1(x[0]), gelu_fast1(x[1])};
  const f32x2_t k3 = {-1.285982656e-05f, -1.285982656e-05f}, k2 = {1.435476415e-03f, 1.435476415e-03f},
                k1 = {-1.096917929e-01f, -1.096917929e-01f}, k0 = {-2.296416554e+00f, -2.296416554e+00f}, one = {1.f, 1.f};
  const f32x2_t x2 = x * x;
  f32x2_t t = x2 * k3 + k2;
  t = t * x2 + k1;
  t = t * x2 + k0;
  const f32x2_t y = t * x;
  f32x2_t e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
  e = e + one;
  const f32x2_t r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
  return x * r;
}

struct MlpFusedParams {
  const op16_t* A; int lda;      // xn [M][384] bf16 (LNF = false)
  const float *ln_g, *ln_b;      // LayerNorm affine (LNF = true: the kernel normalises X's rows itself)
  float ln_eps;
  const op16_t* W1;              // [F][384]
  const op16_t* W2;              // [384][F], hidden index permuted (bits 2 <-> 3 inside groups of 16)
  const float* b1;               // [F]
  const float* b2;               // [384]
  const float* ls;               // optional LayerScale [384] (nullptr = none)
  float* X; int ldx;             // residual stream [M][384] fp32, updated in place
  int M, F;
  // PROJ: the attention output projection of the same block runs in this kernel's prologue: x += (attn Wp^T + bp) (* ls1) first
  const op16_t* attn; int lda_attn;   // attention output rows [M][384] bf16
  const float* bp;               // [384]
  const float* ls1;              // optional LayerScale of the attention branch
  const op16_t* Wp;              // [384][384]
  // NXT (with ACC): the LayerNorm that FOLLOWS this block (norm1 of the next block) is applied to the finished rows before they
  // leave, and its result written as ready-made operand fragments for qkv_fused.hip's PRE form (layout there)
  const float *nx_g, *nx_b; float nx_eps;
  op16_t* xn_next;
  long long* dbg;   // TIMING: per wave {prologue (rows / LayerNorm), fc1 slices, GELU + pack, fc2 slices, epilogue, projection slices (in [0] too), -, total, and inside the slices: wait for slice si + 1, barrier, DMA issue} in shader cycles
};

// One 16-byte store of a row piece.  gfx950: a buffer_store_dwordx4 with an SGPR soffset is still reading its data registers for a
// few cycles after issue, and a VALU write to them in the next two issue slots corrupts some lanes of the stored data; LLVM's
// hazard recognizer covers only the immediate-soffset form (scripts/check_store_hazard.py screens the whole library for the
// pattern).  The nop keeps the register allocator's favourite move -- recycling the result registers at once -- out of the window.
__device__ inline void store_b128_guarded(u32x4_t v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

// ACC (with PROJ, no LayerScale): the residual rows live in the ACCUMULATOR registers for the whole row block.  `out` starts as
// x + bp (loaded in the accumulator layout: lane = row, 4 consecutive columns per register group), the projection MFMAs add to
// it, the LayerNorm reads it where it lies -- the lane pair (l, l ^ 32) holds one row; and because a lane's 8 values of every 16
// columns are the k-slots {4 hi .. + 3, 8 + 4 hi .. + 3}, they ARE the fc1 operand fragment once fc1.weight is stored with the same
// bit-2 <-> bit-3 swap of its column index as W2's hidden index (p.W1 = that copy) -- and fc2 accumulates on top.  One read and one
// write of the residual stream per block instead of two and two, no LDS transposes except the final store, and the next row
// block's rows are fetched into the accumulator registers as the store phase frees them.
template <bool LNF, bool TIMING = false, bool PROJ = false, bool ACC = false, bool NXT = false>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(MlpFusedParams p) {
  wvn_fp16_saturate();
  static_assert(!ACC || (PROJ && LNF), "ACC is a form of the projection + LayerNorm + MLP kernel");
  static_assert(!NXT || ACC, "NXT is an extension of the resident form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int NTL = p.F / HT;                         // hidden tiles (24)
  const int nrb = (p.M + BM - 1) / BM;
  const int my_rb = (nrb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // row blocks of this workgroup
  constexpr int PS = PROJ ? 18 : 0;                 // projection slices per row block: 6 column pairs x 3 k-slices of Wp
  const int per_rb = PS + NTL * 6;
  const int total = my_rb * per_rb;                // ring slices this workgroup consumes
  float* b1_l = (float*)(smem + B1_OFF);
  float* b2_l = (float*)(smem + B2_OFF);
  float* lng_l = (float*)(smem + B1_OFF + p.F * 4);   // LNF: gamma [384], beta [384]
  float* bp_l = lng_l + 2 * KD;                        // PROJ: projection bias [384]
  for (int i = tid; i < p.F; i += 256) b1_l[i] = p.b1 ? p.b1[i] : 0.f;
  for (int i = tid; i < KD; i += 256) b2_l[i] = p.b2 ? p.b2[i] : 0.f;
  if constexpr (LNF)
    for (int i = tid; i < KD; i += 256) { lng_l[i] = p.ln_g[i]; lng_l[KD + i] = p.ln_b[i]; }
  if constexpr (PROJ && !ACC)
    for (int i = tid; i < KD; i += 256) bp_l[i] = p.bp ? p.bp[i] : 0.f;
  float* nxg_l = bp_l;   // NXT: gamma [384], beta [384] of the following LayerNorm (ACC takes its biases from memory: 12 values per lane and row block)
  if constexpr (NXT)
    for (int i = tid; i < KD; i += 256) { nxg_l[i] = p.nx_g[i]; nxg_l[KD + i] = p.nx_b[i]; }

  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, (unsigned)((size_t)p.F * KD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2, 0, (unsigned)((size_t)KD * p.F * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc((void*)(PROJ ? p.Wp : p.W1), 0, (unsigned)(KD * KD * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((size_t)p.M * p.ldx * 4), 0x00020000);
  // ---- DMA lane offsets: 16 wave-instructions of 1 KB per slice, 4 per wave ------------------------------------------------
  // W1 / Wp slice [64 rows][128 k]: LDS rows of 256 B, instruction = 4 rows; chunk XOR (row & 15)    (gemm_a384.hip)
  // W2 slice [128 n][64 h]: LDS rows of 128 B, instruction = 8 rows; chunk XOR ((row >> 1) & 7)      (attention_bf16.hip K tile)
  unsigned w1off[4], w2off[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int inst = wave * 4 + u;
    const int r1 = inst * 4 + (lane >> 4);
    w1off[u] = (unsigned)((r1 * KD + (((lane & 15) ^ (r1 & 15)) * 8)) * 2);
    const int r2 = inst * 8 + (lane >> 3);
    w2off[u] = (unsigned)(((size_t)r2 * p.F + (((lane & 7) ^ ((r2 >> 1) & 7)) * 8)) * 2);
  }
  // Slice stream in consumption order, the DMA running four slices ahead of the MFMAs.  Per row block: (PROJ) 18 slices of Wp,
  // column pair cp x k-slice ks; then per hidden tile j positions 0, 1, 2 = W1 k-slices and 3, 4, 5 = W2 n-thirds.  The KIND of
  // every slice is static at its issue site (0 Wp, 1 W1, 2 W2): its own descriptor, a scalar offset, one of two lane-offset sets.
  auto issue = [&](auto Kc, int slot, int a, int b, int u0 = 0, int u1 = 4) {   // pieces u0 .. u1 - 1 of the slice (4 per wave)
    constexpr int K = decltype(Kc)::value;
    unsigned char* dst = smem + slot * SLICE + wave * 4096;
    const unsigned soff = __builtin_amdgcn_readfirstlane(K == 2 ? (unsigned)(((b * 128) * p.F + a * HT) * 2) : (unsigned)((a * HT * KD + b * 128) * 2));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < u0 || u >= u1) continue;
      if constexpr (K == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wp, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, w1off[u], soff, 0, 0);
      else if constexpr (K == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, w1off[u], soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, w2off[u], soff, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
  // slice at position T (0..5) of hidden tile jj
  auto issue_mlp = [&](auto Tc, int slot, int jj, int u0 = 0, int u1 = 4) {
    constexpr int T = decltype(Tc)::value;
    if constexpr (T < 3) issue(K1{}, slot, jj, T, u0, u1);
    else issue(K2{}, slot, jj, T - 3, u0, u1);
  };
  long long tm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto now_ = [&]() -> long long {
    if constexpr (TIMING) { __builtin_amdgcn_sched_barrier(0); return (long long)__builtin_amdgcn_s_memtime(); }
    return 0;
  };
  static_assert(NS == 5, "the waits below are written for a ring of 5");
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
  // prologue: slices 0, 1, 2 in flight (every workgroup owns at least one row block)
  if constexpr (PROJ) { issue(K0{}, 0, 0, 0); issue(K0{}, 1, 0, 1); issue(K0{}, 2, 0, 2); }
  else { issue_mlp(I0{}, 0, 0); issue_mlp(I1{}, 1, 0); issue_mlp(I2{}, 2, 0); }

  // fragment addressing: fragment i of a slice at position Q (the i-th of its 16 MFMAs) in ring slot `slot`
  // (12 precomputed swizzled lane offsets: a read costs one v_add of the slot base)
  unsigned off1[8], off2[4];
#pragma unroll
  for (int s = 0; s < 8; ++s) off1[s] = l31 * 256 + (((2 * s + hi) ^ (l31 & 15)) << 4);          // W1 slice: row l31, k-step s
#pragma unroll
  for (int sg = 0; sg < 4; ++sg) off2[sg] = l31 * 128 + (((2 * sg + hi) ^ ((l31 >> 1) & 7)) << 4);  // W2 slice: row l31, k-step sg
  auto frag = [&](auto Qc, int slot, int i) -> opx8_t {
    constexpr int Q = decltype(Qc)::value;
    if constexpr (WVN_MLP_EXP & 16) slot = Q % NS;   // (timing experiment: what compile-time slot offsets would buy -- a ring of six)
    if constexpr (Q < 3) return *(const opx8_t*)(smem + slot * SLICE + off1[i >> 1] + (i & 1) * 8192);   // sub-tile t = i & 1
    else return *(const opx8_t*)(smem + slot * SLICE + off2[i >> 2] + (i & 3) * 4096);                    // column tile T = i & 3
  };
  const unsigned cvoff = (unsigned)(((lane >> 5) * p.ldx + (lane & 31) * 4) * 4);
  float* stg = (float*)(smem + STG_OFF + wave * STG_BYTES);

  int si = 0;      // workgroup-local index of the slice being multiplied
  int rslot = 0;   // its ring slot
  // "slice n = si + 1 is readable": it has landed for this wave when at most the two younger slices (4 DMAs each) are outstanding
  // -- loads complete in order among themselves, and whatever else is in flight (epilogue loads / stores, A rows) only adds to the
  // counter -- and for everybody after the barrier, which also says that every wave is done with slice si - 1: its slot takes
  // slice si + 4 (the next of the issue stream).  Runs in the MIDDLE of slice si, so that the tail of si can
  // already fetch the first fragments of si + 1.
  // SPREAD: only the first of the wave's four DMA pieces is issued here, the others follow two MFMAs apart in the second half of
  // the slice (late_piece): a piece issued right behind another waits for it in the address path (60 - 180 cycles each, MI355X
  // guide), and a wave alone on its SIMD has nothing to issue meanwhile.
  auto late_piece = [&](auto issue_fn, int u) {
    if constexpr (SPREAD && !(WVN_MLP_EXP & 1)) {
      __builtin_amdgcn_sched_barrier(0);
      if (si + 4 < total) issue_fn(rslot == 0 ? NS - 1 : rslot - 1, u, u + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto open_next = [&](auto issue_fn) {   // issue_fn(slot, u0, u1): the DMA of slice si + 4, whose kind the caller knows statically
    __builtin_amdgcn_sched_barrier(0);
    if (si + 1 < total) {
      const long long o0 = now_();
      if constexpr (!(WVN_MLP_EXP & 8)) {
        if (si + 3 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const long long o1 = now_();
      if constexpr (!(WVN_MLP_EXP & 2)) __builtin_amdgcn_s_barrier();
      const long long o2 = now_();
      if constexpr (!(WVN_MLP_EXP & 1))
        if (si + 4 < total && (!STAGGER || wave == 0)) issue_fn(rslot == 0 ? NS - 1 : rslot - 1, 0, SPREAD ? 1 : 4);   // into the slot of slice si - 1
      if constexpr (TIMING) { const long long o3 = now_(); tm[8] += o1 - o0; tm[9] += o2 - o1; tm[10] += o3 - o2; }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // scheduling shape of half a slice: 8 x (one MFMA, one fragment read)
  auto shape8 = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // MFMAs 8 .. 15 of a slice (+ their fragment reads), with the remaining DMA pieces of slice si + 4 between them
  auto second_half = [&](auto step, auto next_dma) {
    if constexpr (SPREAD) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        step(8 + 2 * h);
        step(9 + 2 * h);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (h < 3) late_piece(next_dma, h + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (STAGGER && !(WVN_MLP_EXP & 1)) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        step(8 + 2 * h);
        step(9 + 2 * h);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (h < 3) {
          if (wave == h + 1 && si + 4 < total) next_dma(rslot == 0 ? NS - 1 : rslot - 1, 0, 4);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int i = 8; i < 16; ++i) step(i);
      shape8();
    }
  };

  __syncthreads();   // bias / LayerNorm tables visible
  // open slice 0 and fetch its first four fragments; from here on wf[] always holds the fragments of the next four MFMAs
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (PROJ) issue(K0{}, 3, 1, 0);
  else issue_mlp(I3{}, 3, 0);
  opx8_t wf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wf[i] = frag(I0{}, 0, i);

  auto now = now_;
  const long long t_begin = now();
  f32x16_t out[12];
  // ACC: the residual rows of row block rbn in the accumulator layout, tiles t0 .. t0 + 3 (rows past M read as zero: descriptor bounds)
  const unsigned avoff = (unsigned)((l31 * p.ldx + 4 * hi) * 4);
  auto load_xacc = [&](int rbn, int t0) {
#pragma unroll
    for (int t = t0; t < t0 + 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned so = __builtin_amdgcn_readfirstlane(((rbn * BM + wave * 32) * p.ldx + 32 * t + 8 * g) * 4);
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, avoff, so, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[t][4 * g + e] = __uint_as_float(v[e]);
      }
  };
  // ACC: out += bias[column], through the matrix pipe (see the projection phase)
  auto add_bias = [&](const float* bias) {
    if (!bias) return;
    union { u32x4_t u; opx8_t v; } ones;
    ones.u = u32x4_t{hi == 0 ? WVN_OP_ONE2 : 0u, 0u, 0u, 0u};
    float bv[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) bv[t] = bias[32 * t + l31];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const float bh = op_to_f32(f32_to_op(bv[t]));
      union { u32x4_t u; opx8_t v; } bf;
      bf.u = u32x4_t{hi == 0 ? pack_op2(bh, bv[t] - bh) : 0u, 0u, 0u, 0u};
      out[t] = wvn_mfma_32x32x16(bf.v, ones.v, out[t], 0, 0, 0);
    }
  };
  // ACC: LayerNorm of the rows where they lie (the accumulators), handed out as the 24 operand fragments of the wave's 32 rows:
  // emit(s, fragment bits).  Lane (l31, hi) holds the columns 32 t + 8 g + 4 hi + e of row l31; the accumulators are only ever READ
  // here, by explicit accumulator reads: an ordinary VALU use of an accumulator value would tie all 192 of them to the 256
  // architectural VGPRs for the whole kernel, and there is no room for that beside the operand fragments.
  auto ln_acc = [&](const float* gb_tab, float eps, auto emit) {
    auto rd4 = [&](int t, int g) -> f32x4_t {
      f32x4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(out[t][4 * g + e]));
        v[e] = r;
      }
      return v;
    };
    // statistics in ONE sweep over the accumulators (each read is an instruction): sums of d = x - c and of d^2 around a per-lane
    // pivot c (the lane's first value: no cancellation however far the row sits from zero), two partial sums each; the lane pair
    // (l, l ^ 32) that shares a row combines its halves by the parallel-variance rule
    f32x2_t s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
    float piv = 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t v = rd4(t, g);
        if (t == 0 && g == 0) piv = v[0];
        const f32x2_t d0 = f32x2_t{v[0], v[1]} - f32x2_t{piv, piv}, d1 = f32x2_t{v[2], v[3]} - f32x2_t{piv, piv};
        s1a += d0; s1b += d1;
        s2a += d0 * d0; s2b += d1 * d1;
      }
    const float s1 = (s1a[0] + s1a[1]) + (s1b[0] + s1b[1]), s2 = (s2a[0] + s2a[1]) + (s2b[0] + s2b[1]);
    const float mean_l = piv + s1 * (1.f / 192.f), m2_l = s2 - s1 * s1 * (1.f / 192.f);
    const float mean_p = __shfl_xor(mean_l, 32, 64), m2_p = __shfl_xor(m2_l, 32, 64);
    const float mean = 0.5f * (mean_l + mean_p), dm = mean_l - mean_p;
    const float var = fmaxf((m2_l + m2_p + dm * dm * 96.f) * (1.f / 384.f), 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KD / 16; ++s) {   // k-step s = columns 16 s .. + 15: this lane's slots are 16 s + 4 hi + e and 16 s + 8 + 4 hi + e
      u32x4_t o;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const f32x4_t g4 = *(const f32x4_t*)(gb_tab + 16 * s + 8 * h2 + 4 * hi);
        const f32x4_t b4 = *(const f32x4_t*)(gb_tab + KD + 16 * s + 8 * h2 + 4 * hi);
        const f32x4_t v = rd4(s >> 1, 2 * (s & 1) + h2);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[e] - mean) * (rstd * g4[e]) + b4[e];
        o[2 * h2] = pack_op2(y[0], y[1]);
        o[2 * h2 + 1] = pack_op2(y[2], y[3]);
      }
      asm volatile("" : "+v"(o));   // pins the arithmetic between the accumulator reads of this step and of the next: code sinking
      emit(s, o);                   // would otherwise park all 192 read values (and the table rows) until the first use of the fragments
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (and the scheduler would hoist all 48 table reads to the top: 192 registers)
    }
  };
  const __amdgpu_buffer_rsrc_t rs_nx = __builtin_amdgcn_make_buffer_rsrc((void*)(NXT ? p.xn_next : (op16_t*)p.X), 0, (unsigned)((size_t)((p.M + 31) / 32) * 24 * 1024), 0x00020000);
  if constexpr (ACC) { load_xacc(blockIdx.x, 0); load_xacc(blockIdx.x, 4); load_xacc(blockIdx.x, 8); }
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m0w = rb * BM + wave * 32;
    const long long c_p0 = now();
    f32x4_t xq[KD / 8];   // the wave's rows in the fragment layout (lane = row l31, columns 16 k + 8 hi .. + 7), fp32: LayerNorm input
    // keep_tag: also hand the updated rows over in xq (the LayerNorm that follows then needs no memory at all)
    auto residual_update = [&](const float* bias_tab, const float* ls_vec, auto keep_tag) {
      constexpr bool KEEP = decltype(keep_tag)::value;
      // ---- residual update: x[rows of this wave][384] += (out + bias) (* ls), 128 columns at a time through the wave's LDS image ----
      // (the staging area does not overlap the ring: the W stream of the next row block keeps flowing meanwhile).  The residual
      // rows are fetched one 128-column chunk ahead: a wave alone on its SIMD has nothing else to hide a load behind.
      u32x4_t xr0[16], xr1[16];
      auto load_x = [&](int c, u32x4_t (&dst)[16]) {
  #pragma unroll
        for (int it = 0; it < 16; ++it) {
          const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 2 * it) * p.ldx + 128 * c) * 4);
          dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, cvoff, so, 0);
        }
      };
      load_x(0, xr0);
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c == 0) load_x(1, xr1);
        if (c == 1) load_x(2, xr0);
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
  #pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16_t& a = out[4 * c + tt];
            const f32x4_t o = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
            *(f32x4_t*)(stg + l31 * STG_PITCH + 32 * tt + 8 * g + 4 * hi) = o;
          }
        const f32x4_t b4 = *(const f32x4_t*)(bias_tab + 128 * c + (lane & 31) * 4);
        f32x4_t l4 = {1.f, 1.f, 1.f, 1.f};
        if (ls_vec) l4 = *(const f32x4_t*)(ls_vec + 128 * c + (lane & 31) * 4);
  #pragma unroll
        for (int it = 0; it < 16; ++it) {
          const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 2 * it) * p.ldx + 128 * c) * 4);
          const f32x4_t v = *(const f32x4_t*)(stg + (2 * it + (lane >> 5)) * STG_PITCH + (lane & 31) * 4);
          const u32x4_t r = (c & 1) ? xr1[it] : xr0[it];
          u32x4_t o;
  #pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __float_as_uint((v[e] + b4[e]) * l4[e] + __uint_as_float(r[e]));
          store_b128_guarded(o, rs_x, cvoff, so);  // rows >= M fall outside num_records: dropped
          if constexpr (KEEP) {   // back into the image, in place (row layout): the new rows
            f32x4_t of;
  #pragma unroll
            for (int e = 0; e < 4; ++e) of[e] = __uint_as_float(o[e]);
            *(f32x4_t*)(stg + (2 * it + (lane >> 5)) * STG_PITCH + (lane & 31) * 4) = of;
          }
        }
        if constexpr (KEEP) {     // ... and out again in the fragment layout: 8 k-steps of this 128-column chunk
  #pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            xq[2 * (8 * c + kk)] = *(const f32x4_t*)(stg + l31 * STG_PITCH + 16 * kk + 8 * hi);
            xq[2 * (8 * c + kk) + 1] = *(const f32x4_t*)(stg + l31 * STG_PITCH + 16 * kk + 8 * hi + 4);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (PROJ) {
      // ---- attention projection of this block first: x += (attn Wp^T + bp) (* ls1), the wave's 32 rows, all 384 columns ----
      opx8_t af[KD / 16];
      {
        const op16_t* ap = p.attn + (size_t)min(m0w + l31, p.M - 1) * p.lda_attn + hi * 8;
#pragma unroll
        for (int s = 0; s < KD / 16; ++s) af[s] = *(const opx8_t*)(ap + s * 16);
      }
      if constexpr (ACC) {
        // out = x (in flight since the previous store phase) + bp.  The bias joins through the matrix pipe -- one MFMA per column
        // tile whose only non-zero k-slots are 0 and 1: bp split into two operand-format terms (hi + lo, exact to 2^-17 / 2^-22
        // relative) against ones -- so that no VALU instruction ever writes an accumulator (see the LayerNorm below).
        add_bias(p.bp);
      } else {
#pragma unroll
        for (int t = 0; t < 12; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[t][r] = 0.f;
      }
      // one slice of Wp: rows (output columns) 64 cp .. + 63, k 128 ks .. + 127: 16 MFMAs on out[2 cp], out[2 cp + 1]
      auto pslice = [&](auto CPc, auto KSc) {
        constexpr int cp = decltype(CPc)::value, ks = decltype(KSc)::value;
        const int nslot = rslot == NS - 1 ? 0 : rslot + 1;
        auto step = [&](int i) {
          out[2 * cp + (i & 1)] = wvn_mfma_32x32x16(wf[i & 3], af[ks * 8 + (i >> 1)], out[2 * cp + (i & 1)], 0, 0, 0);
          wf[i & 3] = i + 4 < 16 ? frag(I0{}, rslot, i + 4) : frag(I0{}, nslot, i + 4 - 16);   // (the next slice is W1-shaped too)
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) step(i);
        shape8();
        auto next_dma = [&](int slot, int u0, int u1) {   // slice 3 cp + ks + 4 of this row block: still Wp, or one of the first four of hidden tile 0
          constexpr int t = 3 * cp + ks + 4;
          if constexpr (t < 18) issue(K0{}, slot, t / 3, t % 3, u0, u1);
          else issue_mlp(std::integral_constant<int, t - 18>{}, slot, 0, u0, u1);
        };
        open_next(next_dma);
        second_half(step, next_dma);
        ++si;
        rslot = nslot;
      };
      const long long c_ps = now();
      pslice(I0{}, I0{}); pslice(I0{}, I1{}); pslice(I0{}, I2{});
      pslice(I1{}, I0{}); pslice(I1{}, I1{}); pslice(I1{}, I2{});
      pslice(I2{}, I0{}); pslice(I2{}, I1{}); pslice(I2{}, I2{});
      pslice(I3{}, I0{}); pslice(I3{}, I1{}); pslice(I3{}, I2{});
      pslice(I4{}, I0{}); pslice(I4{}, I1{}); pslice(I4{}, I2{});
      pslice(I5{}, I0{}); pslice(I5{}, I1{}); pslice(I5{}, I2{});
      if constexpr (TIMING) tm[5] += now() - c_ps;
      if constexpr (!ACC) residual_update(bp_l, p.ls1, std::true_type{});   // ... and the updated rows stay in registers for the LayerNorm below
    }
    // ---- A rows -> registers (MFMA operand layout: row l31, k = 16 s + 8 hi .. + 7); rows past M are clamped ----
    opx8_t xf[KD / 16];
    if constexpr (ACC) {
      ln_acc(lng_l, p.ln_eps, [&](int s, u32x4_t o) { union { u32x4_t u; opx8_t v; } f; f.u = o; xf[s] = f.v; });
    } else if constexpr (LNF) {
      // LayerNorm of the wave's 32 residual rows, once per row block: a lane holds half a row (the k-slots it feeds the MFMAs),
      // its partner lane ^ 32 the other half
      // The rows come in coalesced (an instruction = 4 rows x 256 contiguous bytes = 8 whole lines; the fragment layout would
      // touch 32 lines for the same 1 KB, and the L1 takes lines, not bytes) and reach the fragment layout through the wave's LDS
      // image, 64 columns at a time.
      if constexpr (!PROJ) {
        const int lr = lane >> 4, lc = (lane & 15) * 4;            // loader: row 4 i + lr, columns lc .. lc + 3 of the chunk
        f32x4_t buf[2][8];
        auto load_chunk = [&](int c, f32x4_t (&b)[8]) {
#pragma unroll
          for (int i = 0; i < 8; ++i) b[i] = *(const f32x4_t*)(p.X + (size_t)min(m0w + 4 * i + lr, p.M - 1) * p.ldx + 64 * c + lc);
        };
        load_chunk(0, buf[0]);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          if (c + 1 < 6) load_chunk(c + 1, buf[(c + 1) & 1]);
#pragma unroll
          for (int i = 0; i < 8; ++i) *(f32x4_t*)(stg + (4 * i + lr) * 68 + lc) = buf[c & 1][i];   // [32][68] fp32
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            xq[2 * (4 * c + kk)] = *(const f32x4_t*)(stg + l31 * 68 + 16 * kk + 8 * hi);
            xq[2 * (4 * c + kk) + 1] = *(const f32x4_t*)(stg + l31 * 68 + 16 * kk + 8 * hi + 4);
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
      for (int s = 0; s < KD / 16; ++s) {
        union { u32x4_t u; opx8_t v; } o;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4_t g4 = *(const f32x4_t*)(lng_l + 16 * s + 8 * hi + 4 * h2);
          const f32x4_t b4 = *(const f32x4_t*)(lng_l + KD + 16 * s + 8 * hi + 4 * h2);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (xq[2 * s + h2][e] - mean) * rstd * g4[e] + b4[e];
          o.u[2 * h2] = pack_op2(y[0], y[1]);
          o.u[2 * h2 + 1] = pack_op2(y[2], y[3]);
        }
        xf[s] = o.v;
      }
    } else {
      const op16_t* ap = p.A + (size_t)min(m0w + l31, p.M - 1) * p.lda + hi * 8;
#pragma unroll
      for (int s = 0; s < KD / 16; ++s) xf[s] = *(const opx8_t*)(ap + s * 16);
    }
    if constexpr (TIMING) tm[0] += now() - c_p0;
    if constexpr (!ACC) {
#pragma unroll
      for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[t][r] = 0.f;
    }

    for (int j = 0; j < NTL; ++j) {
      const int jn = j + 1 == NTL ? 0 : j + 1;
      // ---- fc1: h[32 rows][64 hidden] = xn W1_j^T + b1_j  (accumulators start at the bias) ----
      f32x16_t hacc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t b4 = *(const f32x4_t*)(b1_l + j * HT + 32 * t + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) hacc[t][4 * g + e] = b4[e];
        }
      opx8_t hf[4];
      // one slice = 16 MFMAs; MFMA i takes wf[i & 3], which is then refilled with fragment i + 4 -- of this slice, or (i >= 12) of
      // the next one, opened half-way through
      auto slice = [&](auto Qc) {
        constexpr int Q = decltype(Qc)::value;
        constexpr int QN = (Q + 1) % 6;
        const int nslot = rslot == NS - 1 ? 0 : rslot + 1;
        auto step = [&](int i) {
          if constexpr (Q < 3) hacc[i & 1] = wvn_mfma_32x32x16(wf[i & 3], xf[Q * 8 + (i >> 1)], hacc[i & 1], 0, 0, 0);
          else out[4 * (Q - 3) + (i & 3)] = wvn_mfma_32x32x16(wf[i & 3], hf[i >> 2], out[4 * (Q - 3) + (i & 3)], 0, 0, 0);
          if constexpr (WVN_MLP_EXP & 4) asm volatile("" : "+v"(wf[i & 3]));
          else if (i + 4 < 16) wf[i & 3] = frag(Qc, rslot, i + 4);
          else wf[i & 3] = frag(std::integral_constant<int, QN>{}, nslot, i + 4 - 16);
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) step(i);
        shape8();
        auto next_dma = [&](int slot, int u0, int u1) {   // position Q + 4: of this tile, of the next tile, or (PROJ, last tile) a Wp slice of the next row block
          if constexpr (Q < 2) issue_mlp(std::integral_constant<int, Q + 4>{}, slot, j, u0, u1);
          else if (PROJ && j + 1 == NTL) issue(K0{}, slot, (Q - 2) / 3, (Q - 2) % 3, u0, u1);
          else issue_mlp(std::integral_constant<int, Q - 2>{}, slot, jn, u0, u1);
        };
        open_next(next_dma);
        second_half(step, next_dma);
        ++si;
        rslot = nslot;
      };
      const long long c0 = now();
      slice(I0{}); slice(I1{}); slice(I2{});
      const long long c1 = now();
      // ---- GELU, pack: the accumulator registers become the fc2 B-operand fragments (k-step sigma = 2 t + (g >> 1)) ----
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          union { u32x4_t u; opx8_t v; } o;
#pragma unroll
          for (int q = 0; q < 2; ++q) {   // g = 2 gp + q: slots j = 4 q + e
            const int g = 2 * gp + q;
            const f32x2_t a = gelu_fast2(f32x2_t{hacc[t][4 * g + 0], hacc[t][4 * g + 1]});
            const f32x2_t b = gelu_fast2(f32x2_t{hacc[t][4 * g + 2], hacc[t][4 * g + 3]});
            o.u[2 * q] = pack_op2(a[0], a[1]);
            o.u[2 * q + 1] = pack_op2(b[0], b[1]);
          }
          hf[2 * t + gp] = o.v;
        }
      __builtin_amdgcn_sched_barrier(0);
      // ---- fc2: out[32 rows][384] += h W2_j^T, 128 output columns per slice ----
      const long long c2 = now();
      slice(I3{}); slice(I4{}); slice(I5{});
      if constexpr (TIMING) { tm[1] += c1 - c0; tm[2] += c2 - c1; tm[3] += now() - c2; }
    }
    const long long c_e0 = now();
    if constexpr (ACC) {
      // the next row block's rows (coalesced row pieces), two of the three 128-column chunks at once, the third as soon as the first
      // has moved into the accumulators
      const int rbn = rb + (int)gridDim.x;
      constexpr bool more = true;
      u32x4_t xa[16], xb[16];
      auto load_rows = [&](int c, u32x4_t (&dst)[16]) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const unsigned so = __builtin_amdgcn_readfirstlane(((rbn * BM + wave * 32 + 2 * it) * p.ldx + 128 * c) * 4);
          dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, cvoff, so, 0);   // rows past M: zeros (descriptor bounds)
        }
      };
      // (NXT: after the LayerNorm below -- loaded values that live across its several hundred pinned statements make the register
      //  allocator spill every one of them the moment it arrives)
      if constexpr (!NXT) { load_rows(0, xa); load_rows(1, xb); }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NXT) {
        // the rows are finished once b2 has joined (through the matrix pipe, as bp did); the following LayerNorm reads them where
        // they lie and its fragments leave as 24 coalesced 1 KB stores
        add_bias(p.b2);
        ln_acc(nxg_l, p.nx_eps, [&](int s, u32x4_t o) {
          const unsigned so = __builtin_amdgcn_readfirstlane((((rb * BM) >> 5) + wave) * 24 + s) * 1024u;   // groups past M: dropped (bounds)
          store_b128_guarded(o, rs_nx, lane * 16, so);
        });
        load_rows(0, xa);
        load_rows(1, xb);
      }
      // ---- store phase: out (+ b2) -> the wave's LDS image -> whole 512-byte row pieces.  The next row block's rows come the other
      //      way through the same image (coalesced row pieces -> LDS -> accumulator layout), fetched one 128-column chunk ahead, into
      //      the accumulator registers the stores have just freed: their latency hides behind this phase ----
      // (no branches in here: past the workgroup's last row block the loads fall outside the descriptor and return zeros)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16_t& a = out[4 * c + tt];
            const f32x4_t o = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
            *(f32x4_t*)(stg + l31 * STG_PITCH + 32 * tt + 8 * g + 4 * hi) = o;
          }
        f32x4_t b4 = *(const f32x4_t*)(b2_l + 128 * c + (lane & 31) * 4);
        if constexpr (NXT) b4 = f32x4_t{0.f, 0.f, 0.f, 0.f};   // (already in the accumulators)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 2 * it) * p.ldx + 128 * c) * 4);
          const f32x4_t v = *(const f32x4_t*)(stg + (2 * it + (lane >> 5)) * STG_PITCH + (lane & 31) * 4);
          u32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(v[e] + b4[e]);
          store_b128_guarded(o, rs_x, cvoff, so);  // rows >= M fall outside num_records: dropped
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
          for (int it = 0; it < 16; ++it)
            *(u32x4_t*)(stg + (2 * it + (lane >> 5)) * STG_PITCH + (lane & 31) * 4) = (c & 1) ? xb[it] : xa[it];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4_t v = *(const f32x4_t*)(stg + l31 * STG_PITCH + 32 * tt + 8 * g + 4 * hi);
#pragma unroll
              for (int e = 0; e < 4; ++e) out[4 * c + tt][4 * g + e] = v[e];
            }
          if (c == 0) load_rows(2, xa);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
    residual_update(b2_l, p.ls, std::false_type{});
    if constexpr (TIMING) tm[4] += now() - c_e0;
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
      for (int i = 0; i < 12; ++i) d[i] = tm[i];
      d[7] = now() - t_begin;
    }
  }
}

int mlp_fused_num_cus() {
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

// Eligibility: D == 384, F % 64 == 0 (F * 4 + 154,112 bytes of LDS), 16-byte aligned operands, 32-bit byte offsets.
// W2 must be stored with the hidden index permuted (wvn_hip.h: WVN_VIT_MLP_FUSED).  xn == nullptr: the kernel applies
// LayerNorm(ln_g, ln_b, ln_eps) to the rows of x itself (once per row block).  WVN_ERR_ARG otherwise.
long long* WVN_OPSYM(g_mlp_fused_dbg) = nullptr;   // wvn_debug_mlp_fused_timing (scripts/bench_mlp_fused.py)

static int mlp_fused_launch_impl(const op16_t* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const op16_t* W1, const float* b1,
                                 const op16_t* W2p, const float* b2, const float* ls, float* x, int ldx, int M, int F, const op16_t* attn,
                                 int lda_attn, const op16_t* Wp, const float* bp, const float* ls1, hipStream_t st,
                                 const op16_t* W1p = nullptr, const float* nx_g = nullptr, const float* nx_b = nullptr, float nx_eps = 0.f,
                                 op16_t* xn_next = nullptr) {
  const bool lnf = xn == nullptr, proj = attn != nullptr;
  // the residual rows stay in the accumulators (W1p: fc1.weight with the swapped column bits); its row prefetch runs one grid stride
  // past the last row block (out of the descriptor's bounds, but the byte offset must not wrap)
  const bool acc = proj && W1p && !ls && !ls1 && ((size_t)M + 257 * BM) * ldx * 4 < (1ull << 32);
  if (acc) W1 = W1p;
  const bool nxt = xn_next != nullptr;
  if (nxt && (!acc || !nx_g || !nx_b || ((uintptr_t)xn_next & 15) != 0)) return WVN_ERR_ARG;
  if (!W1 || !W2p || !x || M <= 0 || F <= 0 || (F % HT) != 0 || (ldx % 4) != 0) return WVN_ERR_ARG;
  if (lnf ? (!ln_g || !ln_b) : ((lda % 8) != 0 || ((uintptr_t)xn & 15) != 0)) return WVN_ERR_ARG;
  if ((((uintptr_t)W1 | (uintptr_t)W2p | (uintptr_t)x) & 15) != 0) return WVN_ERR_ARG;
  if (proj && (!lnf || !Wp || (lda_attn % 8) != 0 || (((uintptr_t)attn | (uintptr_t)Wp) & 15) != 0)) return WVN_ERR_ARG;
  const int lds = B1_OFF + F * 4 + (nxt ? 4 : 3) * KD * 4;
  if (lds > 160 * 1024) return WVN_ERR_ARG;
  if ((size_t)M * ldx * 4 >= (1ull << 32)) return WVN_ERR_ARG;
  if ((size_t)F * KD * 2 >= (1ull << 32)) return WVN_ERR_ARG;
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(160 * 1024, (const void*)mlp_fused_kernel<false>, (const void*)mlp_fused_kernel<true>, (const void*)mlp_fused_kernel<true, true>, (const void*)mlp_fused_kernel<true, false, true>, (const void*)mlp_fused_kernel<true, false, true, true>, (const void*)mlp_fused_kernel<true, true, true, true>, (const void*)mlp_fused_kernel<true, false, true, true, true>)) return rc;
  MlpFusedParams p{};
  p.A = xn; p.lda = lda; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps; p.W1 = W1; p.W2 = W2p; p.b1 = b1; p.b2 = b2; p.ls = ls;
  p.X = x; p.ldx = ldx; p.M = M; p.F = F;
  p.attn = attn; p.lda_attn = lda_attn; p.bp = bp; p.ls1 = ls1;
  p.Wp = Wp;
  p.nx_g = nx_g; p.nx_b = nx_b; p.nx_eps = nx_eps; p.xn_next = xn_next;
  const int nrb = ceil_div(M, BM), ncu = mlp_fused_num_cus();
  const dim3 grid(nrb < ncu ? nrb : ncu);
  p.dbg = WVN_OPSYM(g_mlp_fused_dbg);
  if (nxt) hipLaunchKernelGGL((mlp_fused_kernel<true, false, true, true, true>), grid, dim3(256), lds, st, p);
  else if (acc && WVN_OPSYM(g_mlp_fused_dbg)) hipLaunchKernelGGL((mlp_fused_kernel<true, true, true, true>), grid, dim3(256), lds, st, p);
  else if (acc) hipLaunchKernelGGL((mlp_fused_kernel<true, false, true, true>), grid, dim3(256), lds, st, p);
  else if (proj) hipLaunchKernelGGL((mlp_fused_kernel<true, false, true>), grid, dim3(256), lds, st, p);
  else if (lnf && WVN_OPSYM(g_mlp_fused_dbg)) hipLaunchKernelGGL((mlp_fused_kernel<true, true>), grid, dim3(256), lds, st, p);
  else if (lnf) hipLaunchKernelGGL(mlp_fused_kernel<true>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(mlp_fused_kernel<false>, grid, dim3(256), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int WVN_OPSYM(wvn_mlp_fused_launch)(const op16_t* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const op16_t* W1,
                         const float* b1, const op16_t* W2p, const float* b2, const float* ls, float* x, int ldx, int M, int F,
                         hipStream_t st) {
  return mlp_fused_launch_impl(xn, lda, ln_g, ln_b, ln_eps, W1, b1, W2p, b2, ls, x, ldx, M, F, nullptr, 0, nullptr, nullptr, nullptr, st);
}

// The same with the attention output projection of the block in front: x += (attn Wp^T + bp) (* ls1); x += MLP(LayerNorm(x)).
// W1p (optional): fc1.weight with bits 2 and 3 of its COLUMN index swapped inside every aligned group of 16 -- with it, and
// without LayerScale, the kernel keeps the residual rows in its accumulators (ACC above).  W1 may then be nullptr.
// xn_next (optional, with W1p): also apply LayerNorm(nx_g, nx_b, nx_eps) -- the norm1 of the NEXT block -- to the finished rows and
// write it as operand fragments for the PRE form of qkv_fused.hip ((M + 31) / 32 * 24 KB).  WVN_ERR_ARG when the resident form
// does not apply (the caller then runs the plain form and the LayerNorm inside the QKV kernel).
int WVN_OPSYM(wvn_proj_mlp_fused_launch)(const op16_t* attn, int lda_attn, const op16_t* Wp, const float* bp, const float* ls1, const float* ln_g,
                              const float* ln_b, float ln_eps, const op16_t* W1, const float* b1, const op16_t* W2p, const float* b2,
                              const float* ls2, float* x, int ldx, int M, int F, hipStream_t st, const op16_t* W1p,
                              const float* nx_g, const float* nx_b, float nx_eps, op16_t* xn_next) {
  if (!attn || (!W1 && !W1p)) return WVN_ERR_ARG;
  if (!W1 && (ls1 || ls2)) return WVN_ERR_ARG;   // the LayerScale form needs fc1.weight in its own column order
  return mlp_fused_launch_impl(nullptr, 0, ln_g, ln_b, ln_eps, W1 ? W1 : W1p, b1, W2p, b2, ls2, x, ldx, M, F, attn, lda_attn, Wp, bp, ls1, st,
                               W1p, nx_g, nx_b, nx_eps, xn_next);
}
