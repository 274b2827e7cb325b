// A-stationary bf16 MFMA GEMM for the K = 384 linears of the ViT (QKV, attention projection, fc1, STEGO
// hidden layer) on gfx950:  C = epilogue(A[M,384] * W[N,384]^T + bias).
//
// Why a second GEMM kernel: with K = 384 a classic output-tiled GEMM has six K-tiles per tile, so its
// prologue (pipeline fill) and epilogue (bias / GELU / layout change / stores) are as long as its MFMA
// loop.  Here the roles are turned around:
//
//   * a workgroup (8 waves) owns 256 rows of A for the WHOLE N range.  Each wave keeps its 32 rows x 384
//     of A in registers (96 VGPRs, loaded once, already in MFMA operand layout) -- A is never staged
//     through LDS and never re-read;
//   * W streams through a 4-deep LDS ring in slices of 64 (n) x 128 (k) bf16 = 16 KB, written by
//     direct-to-LDS loads (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write), read by all 8 waves.
//     The ring never drains: one s_barrier per slice, counted s_waitcnt vmcnt so two slices stay in flight
//     across every barrier.  LDS rows are 256 B, 16-byte chunks XOR-swizzled by (row & 15) on the SOURCE
//     address (the DMA writes lane-linear) and on the read -> ds_read_b128 is bank-conflict free;
//   * the output is produced 64 columns at a time (3 slices = 48 MFMAs per wave per column tile).  The
//     epilogue of column tile j-1 (bias, GELU, bf16 pack, wave-private LDS transpose, 16-byte coalesced
//     row stores) is software-pipelined into the MFMA loop of tile j, one third per slice, and the two
//     waves that share a SIMD run the two halves in opposite order ("ping-pong": waves 0-3 MFMA then
//     epilogue, waves 4-7 epilogue then MFMA), so within every slice period one wave feeds the matrix pipe
//     while its partner does the VALU / LDS / store work of the epilogue.
//
// MFMA: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  "TR" orientation mfma(Wfrag, Afrag) leaves lane =
// output row m, registers = 4 consecutive columns n (row-major outputs); the V third of the QKV projection
// uses mfma(Afrag, Wfrag) (lane = n, registers = 4 consecutive tokens) so V^T rows are token-contiguous.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int KD = 384;                   // the K this kernel is specialised for
constexpr int BNT = 64;                   // output columns per column tile
constexpr int SLK = 128;                  // k per ring slice
constexpr int NSL = KD / SLK;             // slices per column tile (3)
constexpr int NS = 4;                     // ring depth
constexpr int SLICE_BYTES = BNT * SLK * 2;  // 16 KB
constexpr int RING_BYTES = NS * SLICE_BYTES;
constexpr int STG_BYTES = 8704;           // wave-private epilogue staging (32 x 64 fp32 rows of 272 B)
constexpr int STG_OFF = RING_BYTES;
constexpr int BIAS_OFF = STG_OFF + 8 * STG_BYTES;
constexpr int BM = 256;

enum { A_BF16 = 0, A_GELU = 1, A_RELU = 2, A_RESID = 3, A_QK = 4, A_V = 5 };  // QKV runs as two launches: q|k (TR tiles) and v (V^T tiles)

struct A384Params {
  const op16_t* A; int lda;
  const op16_t* W;     // [N][384]
  const float* bias;   // [N] or nullptr
  void* C; int ldc;    // bf16 or fp32 (A_RESID: in/out)
  int M, N;
  op16_t* q; op16_t* k; op16_t* vt; int heads; int npad; int ntok_s;
  float q_scale;       // A_QK: multiplier of the q column tiles (1 = none)
  op16_t* qkv_base; unsigned q_off, k_off, v_off, qkv_bytes;  // one buffer descriptor for q / k / v^T (byte offsets from qkv_base)
  long long* dbg;  // TIMING builds: per wave {wait+barrier, mfma, epilogue, total} shader cycles
};

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// GELU for a bf16 output: x * sigmoid(g(x)) with g an odd degree-7 polynomial fitted (minimax, weighted by the bf16 ulp of
// the result) to logit(Phi(x)), Phi = the normal CDF of the exact erf GELU (torch.nn.GELU default).  Deviation from the erf
// form: at most 0.25 ulp of the bf16 value it is stored as, over the whole real line (max 2.4e-4 absolute near x = -0.78;
// tests/test_host_logic.py::test_gelu_polynomial pins the bound, tests/test_gpu_gemm_a384.py checks the kernel against erf).
// The common tanh form (degree 3) is up to 170 bf16 ulps off on the negative tail, hence the two extra terms.
// Two elements per call so the plain arithmetic maps to packed fp32 instructions (v_pk_mul/fma/add_f32): 3.5 VALU +
// 2 transcendental issues per element (the Abramowitz-Stegun erf form: 14 + 2).  The coefficients carry the -log2(e) of
// exp(-g) = exp2(-g log2 e).
__device__ inline f32x2_t gelu_fast2(f32x2_t x) {
  const f32x2_t k3 = {-1.285982656e-05f, -1.285982656e-05f}, k2 = {1.435476415e-03f, 1.435476415e-03f},
                k1 = {-1.096917929e-01f, -1.096917929e-01f}, k0 = {-2.296416554e+00f, -2.296416554e+00f}, one = {1.f, 1.f};
  const f32x2_t x2 = x * x;
  f32x2_t t = x2 * k3 + k2;
  t = t * x2 + k1;
  t = t * x2 + k0;
  f32x2_t y = t * x;
  // (fitted on |x| <= 9; beyond that the polynomial factor stays negative -- -1.286e-5 x^2 + 1.435e-3 < 0 from x^2 = 112 on and
  // the outer terms only add negative amounts -- so y -> -inf / +inf, exp2 -> 0 / inf, and the result saturates to x / -0
  // for every finite x without a clamp)
  f32x2_t e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};  // exp(-g)
  e = e + one;
  const f32x2_t r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
  return x * r;
}

template <int EPI>
__device__ inline f32x2_t act2(f32x2_t v) {
  if constexpr (EPI == A_GELU) return gelu_fast2(v);
  if constexpr (EPI == A_RELU) { f32x2_t r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)}; return r; }
  return v;
}

template <int EPI, bool TIMING = false>
__global__ __launch_bounds__(512, 2) void gemm_a384_kernel(A384Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int NT = p.N / BNT;
  // Persistent, balanced schedule: the work is the list of (256-row block, 64-column tile) units in row-block-major
  // order; workgroup w of G owns the contiguous range [w U / G, (w + 1) U / G).  A range is walked as segments of
  // consecutive column tiles of one row block (A rows re-loaded at each row-block change), while the W ring keeps
  // streaming across segments.  Every workgroup gets the same number of units (+-1) whatever M is.
  const int NRB = (p.M + BM - 1) / BM;
  const long long U = (long long)NRB * NT;
  const int u_begin = (int)(U * blockIdx.x / gridDim.x), u_end = (int)(U * (blockIdx.x + 1) / gridDim.x);
  const int total = (u_end - u_begin) * NSL;  // ring slices this workgroup consumes
  int m0w = 0;                                // first row of this wave in the current segment
  const bool epi_first = wave >= 4;
  unsigned char* stg = smem + STG_OFF + wave * STG_BYTES;
  const float* bias_l = (const float*)(smem + BIAS_OFF);

  // ---- W ring producer: 16 wave-instructions of 1 KB (4 rows x 256 B) per slice, 2 per wave ----------
  // (buffer_load ... lds, not global_load_lds: the FLAT-encoded form makes hipcc treat lgkmcnt as out of order
  // and every LDS wait in the kernel becomes lgkmcnt(0), which exposes the full LDS latency in the MFMA loop)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * KD * 2), 0x00020000);
  unsigned wvoff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int inst = wave * 2 + u;
    const int row = inst * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (row & 15);
    wvoff[u] = (unsigned)((row * KD + chunk * 8) * 2);
  }
  int iss_j = u_begin % NT, iss_ks = 0;  // cursor: (column tile, k slice) of the next slice to request (issued in order)
  auto issue = [&](int i) {  // i = workgroup-local slice index (ring slot i % NS)
    const unsigned soff = __builtin_amdgcn_readfirstlane((iss_j * BNT * KD + iss_ks * SLK) * 2);
    unsigned char* dst = smem + (i % NS) * SLICE_BYTES + wave * 2048;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + u * 1024), 16,
                                               wvoff[u], soff, 0, 0);
    if (++iss_ks == NSL) { iss_ks = 0; if (++iss_j == NT) iss_j = 0; }
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < total) issue(i);

  // ---- bias -> LDS, A rows -> registers (MFMA operand layout: row l31, k = 16 s + 8 hi .. + 7) --------
  for (int i = tid; i < p.N; i += 512) ((float*)(smem + BIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
  opx8_t xf[KD / 16];
  auto load_a = [&]() {
    const op16_t* ap = p.A + (size_t)min(m0w + l31, p.M - 1) * p.lda + hi * 8;
#pragma unroll
    for (int s = 0; s < KD / 16; ++s) xf[s] = *(const opx8_t*)(ap + s * 16);
  };

  f32x16_t acc[2], prev[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; prev[t][r] = 0.f; }  // acc is re-initialised per tile

  const int xorc = l31 & 15;
  const unsigned rd_base = l31 * 256;

  // ---- epilogue addressing: buffer descriptors (wave-uniform) + one 32-bit byte offset per lane, computed
  // once; per column tile / chunk row only a scalar offset changes.  Rows >= M fall outside num_records (or get
  // an out-of-range offset) and the hardware drops their stores.
  constexpr unsigned OOB = 0x80000000u;
  unsigned voff[2] = {0, 0};
  unsigned vt_off = 0;
  unsigned stg_rd;  // per-lane byte offset into the wave's staging image for part 2
  constexpr bool IS_QKV = EPI == A_QK || EPI == A_V;
  using TRK = std::integral_constant<bool, EPI != A_V>;  // orientation of every column tile of this launch
  const unsigned c_bytes = IS_QKV ? 0u : (unsigned)((size_t)p.M * p.ldc * (EPI == A_RESID ? 4 : 2));
  const __amdgpu_buffer_rsrc_t rs_c = IS_QKV ? __builtin_amdgcn_make_buffer_rsrc(p.qkv_base, 0, p.qkv_bytes, 0x00020000)
                                                   : __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
  auto qkv_offsets = [&]() {  // per row block: the lane's token rows -> (frame, token) byte offsets
    // q / k rows: row = it*8 + (lane>>3), 16 B at column (lane&7)*8 of the head.  Frames start at multiples of 16 rows
    // (ntok_s % 16 == 0), so rows 0-15 and 16-31 of the wave's block are each contiguous in one frame:
    // voff[0] serves it = 0, 1 (+1024 B) and voff[1] serves it = 2, 3.
#pragma unroll
    for (int hblk = 0; hblk < 2; ++hblk) {
      const int m = m0w + hblk * 16 + (lane >> 3);
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      voff[hblk] = m < p.M ? (unsigned)((((size_t)b * p.heads * p.npad + tk) * 64 + (lane & 7) * 8) * 2) : OOB;
    }
    {  // v^T: d = it*16 + lane>>2, 8 tokens at m0w + (lane&3)*8 (one frame: M % 16 == ntok_s % 16 == 0)
      const int m = m0w + (lane & 3) * 8;
      const int b = m / p.ntok_s, tk = m - b * p.ntok_s;
      vt_off = m < p.M ? (unsigned)((((size_t)b * p.heads * 64 + (lane >> 2)) * p.npad + tk) * 2) : OOB;
    }
  };
  if constexpr (IS_QKV) {
    stg_rd = 0;
  } else if constexpr (EPI == A_RESID) {
    voff[0] = (unsigned)(((lane >> 4) * p.ldc + (lane & 15) * 4) * 4);
    stg_rd = (lane >> 4) * 272 + (lane & 15) * 16;
  } else {
    voff[0] = (unsigned)(((lane >> 3) * p.ldc + (lane & 7) * 8) * 2);
    stg_rd = (lane >> 3) * 144 + (lane & 7) * 16;
  }

  // ---- 16 MFMAs on one ring slice: 4 fragments in flight, one ds_read per MFMA --------------------------
  auto mfma_block = [&](int slot, int ks, auto tr_tag) {
    constexpr bool TR = decltype(tr_tag)::value;
    const unsigned char* base = smem + slot * SLICE_BYTES + rd_base;
    int xc = xorc;
    asm volatile("" : "+v"(xc));  // keep the 8 swizzled offsets from being hoisted out of the loop (8 live VGPRs otherwise)
    auto rd = [&](int s, int t) {
      return *(const opx8_t*)(base + t * 8192 + (((2 * s + hi) ^ xc) << 4));
    };
    constexpr int LA = 2;  // k-steps of fragments in flight
    opx8_t wf[2 * LA];
#pragma unroll
    for (int i = 0; i < 2 * LA; ++i) wf[i] = rd(i >> 1, i & 1);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int slot = (s % LA) * 2 + t;
        if constexpr (TR)
          acc[t] = wvn_mfma_32x32x16(wf[slot], xf[ks * 8 + s], acc[t], 0, 0, 0);
        else
          acc[t] = wvn_mfma_32x32x16(xf[ks * 8 + s], wf[slot], acc[t], 0, 0, 0);
        if (s + LA < 8) wf[slot] = rd(s + LA, t);
      }
    }
    {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * LA, 0);
#pragma unroll
      for (int i = 0; i < 16 - 2 * LA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * LA, 0);
    }
  };

  // ---- one third of the epilogue of column tile jp (accumulators in prev[]) ------------------------------
  // TR tiles : image [32 rows m][64 cols n]  (bf16 rows of 144 B, fp32 rows of 272 B)
  // !TR tiles: image [64 rows n][32 cols m]  (bf16 rows of 80 B)            -- V^T of the QKV projection
  auto epi_part = [&](int part, int jp, auto tr_tag) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = jp * BNT;
    if (part < 2) {  // accumulators already contain the bias (see init_acc)
      const int t = part;
      if constexpr (TR) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 32 * t + 8 * g + 4 * hi;
          f32x2_t a = act2<EPI>(f32x2_t{prev[t][4 * g + 0], prev[t][4 * g + 1]});
          f32x2_t b = act2<EPI>(f32x2_t{prev[t][4 * g + 2], prev[t][4 * g + 3]});
          if constexpr (EPI == A_QK) {  // q tiles carry the softmax scale (wave-uniform per column tile)
            const float qs = n0 < p.heads * 64 ? p.q_scale : 1.f;
            a *= f32x2_t{qs, qs};
            b *= f32x2_t{qs, qs};
          }
          if constexpr (EPI == A_RESID) {
            f32x4_t o = {a[0], a[1], b[0], b[1]};
            *(f32x4_t*)(stg + l31 * 272 + c * 4) = o;
          } else {
            u32x2_t o = {pack_op2(a[0], a[1]), pack_op2(b[0], b[1])};
            *(u32x2_t*)(stg + l31 * 144 + c * 2) = o;
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // tokens 8g + 4hi + e of the wave's 32; V^T is stored with bits 2 and 3 of the token index swapped
          // inside every aligned group of 16 (attention_bf16.hip): position 16 (g >> 1) + 8 hi + 4 (g & 1) + e
          const int mloc = 16 * (g >> 1) + 8 * hi + 4 * (g & 1);
          u32x2_t o = {pack_op2(prev[t][4 * g + 0], prev[t][4 * g + 1]),
                       pack_op2(prev[t][4 * g + 2], prev[t][4 * g + 3])};
          *(u32x2_t*)(stg + (32 * t + l31) * 80 + mloc * 2) = o;
        }
      }
      return;
    }
    // part 2: wave-private image -> global, 16 bytes per lane, whole rows per 8 (bf16) / 16 (fp32) lanes
    if constexpr (EPI == A_RESID) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 4 * it) * p.ldc + n0) * 4);
        f32x4_t v = *(const f32x4_t*)(stg + stg_rd + it * 4 * 272);
        const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs_c, voff[0], so, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(r[e]);  // (not __builtin_bit_cast on a vector element: clang reads element 0)
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(v[e]);
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_c, voff[0], so, 0);
        if (it & 1) __builtin_amdgcn_sched_barrier(0);  // two chunk rows in flight at a time (VGPR budget)
      }
    } else if constexpr (EPI == A_QK) {  // q / k: one column tile = one head; dst[(b*h + head)*npad + t][0..63]
      const int D = p.heads * 64;
      const int which = n0 / D, head = (n0 - which * D) >> 6;
      const unsigned so = __builtin_amdgcn_readfirstlane((which == 0 ? p.q_off : p.k_off) + head * p.npad * 64 * 2);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const u32x4_t val = *(const u32x4_t*)(stg + ((lane >> 3) + it * 8) * 144 + (lane & 7) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(val, rs_c, voff[it >> 1], so + (it & 1) * 1024, 0);
      }
    } else if constexpr (EPI == A_V) {  // v: vt[(b*h + head)*64 + d][t], 8 tokens (16 B) per lane
      const int head = n0 >> 6;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const u32x4_t val = *(const u32x4_t*)(stg + ((lane >> 2) + it * 16) * 80 + (lane & 3) * 16);
        const unsigned so = __builtin_amdgcn_readfirstlane(p.v_off + (head * 64 + it * 16) * p.npad * 2);
        __builtin_amdgcn_raw_buffer_store_b128(val, rs_c, vt_off, so, 0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const u32x4_t val = *(const u32x4_t*)(stg + stg_rd + it * 8 * 144);
        const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 8 * it) * p.ldc + n0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(val, rs_c, voff[0], so, 0);
      }
    }
  };

  // accumulators of column tile j start at the bias (TR: 4 consecutive columns per register group; !TR: the
  // lane's own column), which removes the bias add from the epilogue
  auto init_acc = [&](int j, auto tr_tag) {
    constexpr bool TR = decltype(tr_tag)::value;
    const int n0 = j * BNT;
    if constexpr (TR) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t b4 = *(const f32x4_t*)(bias_l + n0 + 32 * t + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b4[e];
        }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float b = bias_l[n0 + 32 * t + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b;
      }
    }
  };

  // ---- one slice period -----------------------------------------------------------------------------------
  // VM queue at the boundary of slice i (oldest first): DMA(i), [stores of a tile epilogue], DMA(i+1),
  // DMA(i+2) with the stores somewhere behind DMA(i) (they are issued in the ks == 2 period, after that
  // period's DMA).  `allow` = operations that may stay outstanding = everything younger than DMA(i).
  constexpr int ST = EPI == A_RESID ? 16 : 4;  // VM operations of one epilogue part 2, per lane
  long long t_wait = 0, t_mfma = 0, t_epi = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  auto period = [&](int i, int ks, int j, bool stores_in_window, auto mtr, auto etr, bool do_epi_in) {
    const bool do_epi = do_epi_in;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
    if (i + 2 < total) {
      if (ks != 2 && stores_in_window) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + ST) : "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (i + NS - 1 < total) issue(i + NS - 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
    if (epi_first) {
      if (do_epi) {
        __builtin_amdgcn_s_setprio(2);  // the VALU-heavy half wins issue arbitration over the partner's MFMA stream
        epi_part(ks, j - 1, etr);
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c2 = (long long)__builtin_amdgcn_s_memtime(); }
      mfma_block(i % NS, ks, mtr);
    } else {
      mfma_block(i % NS, ks, mtr);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TIMING) { asm volatile("s_nop 7\n s_nop 7" ::: "memory"); c2 = (long long)__builtin_amdgcn_s_memtime(); }
      if (do_epi) {
        __builtin_amdgcn_s_setprio(2);
        epi_part(ks, j - 1, etr);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TIMING) {
      c3 = (long long)__builtin_amdgcn_s_memtime();
      t_wait += c1 - c0;
      if (epi_first) { t_epi += c2 - c1; t_mfma += c3 - c2; } else { t_mfma += c2 - c1; t_epi += c3 - c2; }
    }
  };
  int si = 0;  // workgroup-local slice counter (ring slot si % NS)
  // one column tile: 3 slice periods; jj = position of the tile in its segment (the epilogue of tile jj - 1 rides along)
  auto tile = [&](int j, int jj, int j_end, auto mtr, auto etr, auto next_tr) {
#pragma unroll
    for (int ks = 0; ks < NSL; ++ks) period(si + ks, ks, j, jj >= 2, mtr, etr, jj >= 1);
    si += NSL;
#pragma unroll
    for (int t = 0; t < 2; ++t) prev[t] = acc[t];
    if (j + 1 < j_end) init_acc(j + 1, next_tr);
  };

  __syncthreads();  // bias table visible (nothing DMA'd is read before the first period's wait + barrier)
  for (int u = u_begin; u < u_end;) {
    const int rb = u / NT, j0 = u - rb * NT, j1 = min(NT, j0 + (u_end - u));
    m0w = rb * BM + wave * 32;
    load_a();
    if constexpr (IS_QKV) qkv_offsets();
    init_acc(j0, TRK{});
    for (int j = j0; j < j1; ++j) tile(j, j - j0, j1, TRK{}, TRK{}, TRK{});
    // drain: epilogue of the segment's last tile (not overlapped; a few per workgroup)
#pragma unroll
    for (int part = 0; part < 3; ++part) epi_part(part, j1 - 1, TRK{});
    u += j1 - j0;
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
      d[0] = t_wait; d[1] = t_mfma; d[2] = t_epi; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
    }
  }
}

constexpr int A384_LDS_MAX = 160 * 1024;

int a384_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int EPI, bool TIMING>
int launch_k(const A384Params& p, int lds, hipStream_t st) {
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(A384_LDS_MAX, (const void*)gemm_a384_kernel<EPI, TIMING>)) return rc;
  // one persistent workgroup per CU (144 KB of LDS each), never more workgroups than (row block, column tile) units
  const long long units = (long long)ceil_div(p.M, BM) * (p.N / BNT);
  const int grid = (int)(units < a384_num_cus() ? units : a384_num_cus());
  hipLaunchKernelGGL((gemm_a384_kernel<EPI, TIMING>), dim3(grid), dim3(512), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

template <int EPI>
int launch(const A384Params& p, hipStream_t st) {
  const int lds = BIAS_OFF + p.N * 4;
  if (lds > A384_LDS_MAX) return WVN_ERR_ARG;
  return p.dbg ? launch_k<EPI, true>(p, lds, st) : launch_k<EPI, false>(p, lds, st);
}

}  // namespace

// Eligibility: K == 384, N % 64 == 0, 16-byte aligned operands; returns WVN_ERR_ARG otherwise so the
// caller can use the generic tiled kernel.  epi uses the GemmEpilogue codes of wvn_internal.h.
int WVN_OPSYM(wvn_gemm_a384_launch)(const GemmBf16Params& g, int epi, hipStream_t st) {
  if (g.K != KD || g.ldw != KD || (g.N % BNT) != 0 || g.M <= 0 || !g.A || !g.W || (g.lda % 8) != 0) return WVN_ERR_ARG;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return WVN_ERR_ARG;
  A384Params p{};
  p.A = g.A; p.lda = g.lda; p.W = g.W; p.bias = g.bias; p.C = g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N;
  p.q = g.q; p.k = g.k; p.vt = g.vt; p.heads = g.heads; p.npad = g.npad; p.ntok_s = g.ntok_s;
  p.q_scale = g.q_scale != 0.f ? g.q_scale : 1.f;
  p.dbg = g.dbg;
  switch (epi) {
    case EPI_BF16:
    case EPI_GELU_BF16:
    case EPI_RELU_BF16:
      if (!g.C || (g.ldc % 8) != 0 || ((uintptr_t)g.C & 15)) return WVN_ERR_ARG;
      return epi == EPI_BF16 ? launch<A_BF16>(p, st) : epi == EPI_GELU_BF16 ? launch<A_GELU>(p, st) : launch<A_RELU>(p, st);
    case EPI_RESID_F32:
    case EPI_ACCUM_F32:
      if (!g.C || (g.ldc % 4) != 0 || ((uintptr_t)g.C & 15)) return WVN_ERR_ARG;
      return launch<A_RESID>(p, st);
    case EPI_QKV: {
      const uintptr_t lo = std::min({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const uintptr_t hi = std::max({(uintptr_t)g.q, (uintptr_t)g.k, (uintptr_t)g.vt});
      const size_t one = (size_t)(g.M / (g.ntok_s > 0 ? g.ntok_s : 1)) * g.heads * g.npad * 64 * 2;
      if (hi - lo + one >= (1ull << 31)) return WVN_ERR_ARG;  // one 32-bit-offset buffer descriptor must span q, k and v^T
      p.qkv_base = (op16_t*)lo; p.q_off = (unsigned)((uintptr_t)g.q - lo); p.k_off = (unsigned)((uintptr_t)g.k - lo);
      p.v_off = (unsigned)((uintptr_t)g.vt - lo); p.qkv_bytes = (unsigned)(hi - lo + one);
      if (g.N != 3 * g.heads * 64 || !g.q || !g.k || !g.vt || (g.ntok_s % 16) || (g.M % 16) || (g.npad % 16)) return WVN_ERR_ARG;
      const int D = g.heads * 64;
      p.N = 2 * D;  // q | k column tiles
      const int rc = launch<A_QK>(p, st);
      if (rc != WVN_OK) return rc;
      p.W = g.W + (size_t)2 * D * KD;  // v rows of the fused qkv weight
      p.bias = g.bias ? g.bias + 2 * D : nullptr;
      p.N = D;
      return launch<A_V>(p, st);
    }
    default: return WVN_ERR_ARG;
  }
}
