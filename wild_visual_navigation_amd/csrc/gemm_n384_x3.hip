// Row-panel SPLIT-OPERAND MFMA GEMM for the N = 384 residual updates of the ViT (fc2: K = 1536, attention projection: K = 384):
//     C[M,384] (fp32, in place) += (A[M,K] * W[384,K]^T + bias) (* ls),   A and W as hi + lo bf16 planes,
// every product hi*lo + lo*hi + hi*hi (three v_mfma_f32_32x32x16_bf16 per fragment pair, fp32 accumulation): the arithmetic of
// gemm_x3.hip in the structure of gemm_n384.hip, re-sized for two planes per operand and ONE wave per SIMD.
//
// A workgroup (4 waves) owns 128 rows and all 384 output columns: a wave keeps the 32 x 384 fp32 accumulator of its rows in
// registers (12 MFMA tiles = 192 of the 512 registers), so A -- the 4x wider hidden activation in fc2 -- is read exactly once and the
// fp32 residual read-modify-write happens once per row, after the whole K loop.  W [2 planes][384 n][16 k] (24 KB per slice, shared
// by the 4 waves) and A [2 planes][128 m][16 k] (8 KB, every wave requests and reads only its own
// 32 rows) arrive by buffer_load ... lds DMA -- in slices of ONE k-step (16 k: 32 KB) through a 4-deep LDS ring, three slices in flight.
// A k-step is 36 MFMAs per wave against 26 fragment reads; the MFMAs of two column tiles alternate (hi*lo, lo*hi,
// hi*hi of tile t interleaved with those of tile t + 1), because a filler between two MFMAs on the SAME accumulator costs ~43
// cycles and between different ones ~6 (MI355X guide), and the 16 DMA requests of the next slice ride as fillers, two per tile
// pair of the first k-step and one of the second's first four (a burst behind the barrier costs the lone wave ~60 cycles a piece).
// The epilogue (once per K loop) moves the accumulators through a wave-private LDS image, 128 columns at a time, and updates C
// with whole 512-byte rows.
#include <type_traits>

#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int NN = 384;
constexpr int NTILE = NN / 32;
constexpr int BKS = 16;                          // k per ring slice (ONE MFMA k-step: 36 MFMAs per wave)
constexpr int BM = 128;
constexpr int NS = 4;                            // slices i + 1 .. i + 3 in flight while slice i is multiplied (A comes from HBM)
constexpr int W_PLANE = NN * BKS * 2;            // 12 KB
constexpr int A_PLANE = BM * BKS * 2;            // 4 KB
constexpr int A_OFF = 2 * W_PLANE;
constexpr int STAGE_BYTES = 2 * W_PLANE + 2 * A_PLANE;   // 32 KB
constexpr int RING_BYTES = NS * STAGE_BYTES;     // 128 KB
constexpr int STG_PITCH = 132;                   // floats per staged row (128 columns + 4)
constexpr int STG_BYTES = 32 * STG_PITCH * 4;    // per wave: 16,896 (the four images overlap the ring's first 66 KB)
constexpr int BIAS_OFF = RING_BYTES;
constexpr int LS_OFF = BIAS_OFF + NN * 4;
constexpr int LDS_BYTES = LS_OFF + NN * 4;
constexpr int WP = 6, AP = 2;                    // DMA pieces (1 KB = 32 rows x 32 B) per wave and slice: W, A

struct N384X3Params {
  const bf16_t* A; size_t a_plane; int lda;      // [2][M][lda]: lo plane a_plane elements behind the hi plane
  const bf16_t* W; size_t w_plane; int ldw;      // [2][384][K]
  const float* bias; const float* ls;
  float* C; int ldc;
  int M, K;
  long long* dbg;                                // TIMING builds: per wave {wait + barrier, k-steps, epilogue, total} shader cycles
  float* stats; float eps;                       // optional: stats[m] = {mean, 1 / sqrt(var + eps)} of the updated row m (the next LayerNorm's)
  // patch embedding (EPI_PATCH): row m = (frame b, patch pp) of A lands in C row b * ntok_s + 1 + pp, and what is added to the product is
  // the position row pos[1 + pp] (+ bias) instead of the C row itself: C = A W^T + bias + pos, the cls / padding rows untouched
  const float* pos; int npatch, ntok_s, c_rows;
};

// ---- epilogue: C[rows of this wave][384] += (acc + bias) * ls, 128 columns at a time through the wave's LDS image ----
// stats != nullptr: the epilogue has every finished row in its hands (a row pair per store group, 32 lanes x 4 columns x 3 column groups), so
// it also leaves the LayerNorm statistics of the rows -- sum and sum of squares per lane, folded over the row's 32 lanes at the end --
// and the consumer (gemm_a384_x3.hip, LNA) normalises as it loads: the LayerNorm kernel between the two (1.2 GB per launch at 128 frames)
// disappears.  (One-pass variance in fp32: 384 values, relative error ~2e-5 (1 + mean^2 / var).)
__device__ inline void n384_epilogue(const f32x16_t (&acc)[NTILE], unsigned char* smem, int wave, int lane, int m0w, __amdgpu_buffer_rsrc_t rs_c,
                                     int ldc, const float* bias_l, const float* ls_l, float* stats = nullptr, float eps = 0.f, int M = 0,
                                     const float* pos = nullptr, int npatch = 0, int ntok_s = 0) {
  const int l31 = lane & 31, hi = lane >> 5;
  __syncthreads();  // every wave is done reading the ring: the staging images overlap it
  float* stg = (float*)(smem + wave * STG_BYTES);
  const unsigned cvoff = (unsigned)(((lane >> 5) * ldc + (lane & 31) * 4) * 4);
  float s1[16], s2[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) { s1[it] = 0.f; s2[it] = 0.f; }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16_t& a = acc[4 * c + tt];
        const f32x4_t o = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        *(f32x4_t*)(stg + l31 * STG_PITCH + 32 * tt + 8 * g + 4 * hi) = o;
      }
    const f32x4_t b4 = *(const f32x4_t*)(bias_l + 128 * c + (lane & 31) * 4);
    const f32x4_t l4 = *(const f32x4_t*)(ls_l + 128 * c + (lane & 31) * 4);
    // a wave alone on its SIMD: the sixteen row fetches of the column group are requested together (64 registers), then added and
    // stored -- four at a time they cost sixteen exposed round trips per row block (measured: 56 K cycles per epilogue)
    u32x4_t r[16];
    unsigned so_dst[16];
    const __amdgpu_buffer_rsrc_t rs_pos = __builtin_amdgcn_make_buffer_rsrc((void*)(pos ? pos : bias_l), 0, pos ? (unsigned)((size_t)(npatch + 1) * ldc * 4) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int m = m0w + 2 * it;   // (npatch is even, like m: a row pair never straddles two frames)
      if (pos) {   // (uniform) patch embedding: the addend is the position row, the destination the token row of (frame, patch)
        const int b = m / npatch, pp = m - b * npatch;
        so_dst[it] = __builtin_amdgcn_readfirstlane(((b * ntok_s + 1 + pp) * ldc + 128 * c) * 4);
        r[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_pos, cvoff, __builtin_amdgcn_readfirstlane(((1 + pp) * ldc + 128 * c) * 4), 0);
      } else {
        so_dst[it] = __builtin_amdgcn_readfirstlane((m * ldc + 128 * c) * 4);
        r[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, cvoff, so_dst[it], 0);
      }
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const unsigned so = so_dst[it];
      const f32x4_t v = *(const f32x4_t*)(stg + (2 * it + (lane >> 5)) * STG_PITCH + (lane & 31) * 4);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = (v[e] + b4[e]) * l4[e] + __uint_as_float(r[it][e]);
        o[e] = __float_as_uint(y);
        s1[it] += y;
        s2[it] = fmaf(y, y, s2[it]);
      }
      wvn_store_b128_guarded(o, rs_c, cvoff, so);  // rows >= M fall outside num_records: dropped
    }
  }
  if (stats) {   // (uniform) row 2 it + (lane >> 5): its 32 lanes hold the partial sums
    // fold over the 32 lanes on the VALU's data-parallel primitives: butterflies inside a row of 16 (quad_perm, row_half_mirror,
    // row_mirror: every lane then holds its row's sum), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15): lanes 16 - 31 and
    // 48 - 63 hold the totals.  (Five ds_bpermute per value through __shfl_xor: 6200 cycles per row block, 0.75 ms per step and kernel.)
    auto dpp_add = [](float v, auto ctrl, auto rmask) {
      constexpr int C = decltype(ctrl)::value, R = decltype(rmask)::value;
      return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), C, R, 0xf, false));
    };
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      float a = s1[it], b = s2[it];
      a = dpp_add(a, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{}); b = dpp_add(b, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});
      a = dpp_add(a, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{}); b = dpp_add(b, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});
      a = dpp_add(a, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}); b = dpp_add(b, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});
      a = dpp_add(a, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{}); b = dpp_add(b, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});
      a = dpp_add(a, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}); b = dpp_add(b, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
      const int m = m0w + 2 * it + (lane >> 5);
      if ((lane & 31) == 31 && m < M) {
        const float mean = a * (1.0f / NN);
        const float var = fmaxf(b * (1.0f / NN) - mean * mean, 0.f);
        *(wvn_f32x2_t*)(stats + 2 * (size_t)m) = wvn_f32x2_t{mean, 1.0f / sqrtf(var + eps)};
      }
    }
  }
}

template <bool TIMING>
__global__ __launch_bounds__(256, 1) void gemm_n384_x3_kernel(N384X3Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = p.K / BKS;
  const float* bias_l = (const float*)(smem + BIAS_OFF);
  const float* ls_l = (const float*)(smem + LS_OFF);
  for (int i = tid; i < NN; i += 256) {
    ((float*)(smem + BIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
    ((float*)(smem + LS_OFF))[i] = p.ls ? p.ls[i] : 1.f;
  }

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((p.w_plane + (size_t)NN * p.ldw) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((p.a_plane + (size_t)p.M * p.lda) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)(p.pos ? p.c_rows : p.M) * p.ldc * 4), 0x00020000);
  // W: 24 wave-instructions per slice (12 per plane: 32 rows x 32 B each), 6 per wave; LDS rows are 32 B = two 16-byte chunks, chunk c
  // of row r at position c ^ ((r >> 3) & 1): a ds_read_b128 is served 16 lanes at a time ({0-3, 12-15, 20-27}, ...), rows 8 apart
  // share their banks, and without the swizzle half of every group collides (measured: 2235 instead of ~1400 cycles per k-step)
  unsigned wvoff[WP];
#pragma unroll
  for (int u = 0; u < WP; ++u) {
    const int piece = wave * WP + u, plane = piece / 12, idx = piece - plane * 12;
    const int row = idx * 32 + (lane >> 1);
    wvoff[u] = (unsigned)((plane * p.w_plane + (size_t)row * p.ldw + (((lane & 1) ^ ((row >> 3) & 1)) * 8)) * 2);
  }
  const unsigned rdw = l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4);                         // + plane * W_PLANE + t * 1024
  const unsigned rda = A_OFF + wave * 1024 + l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4);   // + plane * A_PLANE : this wave's 32 A rows

  long long t_wait = 0, t_steps = 0, t_epi = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const int nrb = (p.M + BM - 1) / BM;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m0w = rb * BM + wave * 32;
    unsigned avoff[AP];   // [plane]: rows past M are clamped (their results are dropped)
#pragma unroll
    for (int u = 0; u < AP; ++u) {
      const int rl = lane >> 1, row = min(m0w + rl, p.M - 1);
      avoff[u] = (unsigned)((u * p.a_plane + (size_t)row * p.lda + (((lane & 1) ^ ((rl >> 3) & 1)) * 8)) * 2);
    }
    unsigned soff = 0;
    auto piece_w = [&](int i, int u) {
      unsigned char* st = smem + (i % NS) * STAGE_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(st + (wave * WP + u) * 1024), 16, wvoff[u], soff, 0, 0);
    };
    auto piece_a = [&](int i, int u) {
      unsigned char* st = smem + (i % NS) * STAGE_BYTES;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(st + A_OFF + u * A_PLANE + wave * 1024), 16, avoff[u], soff, 0, 0);
    };
    __syncthreads();  // the previous row block's staging reads are done (and the bias table is visible) before DMA reuses the LDS
#pragma unroll
    for (int i0 = 0; i0 < NS - 1; ++i0)
      if (i0 < nk) {
        soff = (unsigned)(i0 * BKS * 2);
#pragma unroll
        for (int u = 0; u < AP; ++u) piece_a(i0, u);
#pragma unroll
        for (int u = 0; u < WP; ++u) piece_w(i0, u);
      }

    f32x16_t acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int i = 0; i < nk; ++i) {
      long long c0 = 0, c1 = 0;
      if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
      // slice i landed: everything but the requests of slices i + 1 and i + 2 (8 per wave each) has retired.  (The requests go on past
      // the last slice -- unconditionally: a uniform branch around every piece would cut the k-step into basic blocks -- into ring stages
      // nobody reads again; they are drained before the epilogue re-uses the ring as staging.)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WP + AP)) : "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
      soff = __builtin_amdgcn_readfirstlane((i + NS - 1) * BKS * 2);   // slice i + 3 into the stage every wave has just left
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + (i % NS) * STAGE_BYTES;
      {
        const bf16x8_t ah = *(const bf16x8_t*)(st + rda), al = *(const bf16x8_t*)(st + rda + A_PLANE);
        bf16x8_t wh[2][2], wl[2][2];   // [pair parity][tile of the pair]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          wh[0][t] = *(const bf16x8_t*)(st + rdw + t * 1024);
          wl[0][t] = *(const bf16x8_t*)(st + rdw + W_PLANE + t * 1024);
        }
#pragma unroll
        for (int pr = 0; pr < NTILE / 2; ++pr) {
          const int cur = pr & 1, nx = cur ^ 1;
          // the requests of slice i + 3 ride as fillers: the two A pieces (HBM) first, then the six W pieces (L2)
          if (pr < AP) piece_a(i + NS - 1, pr);
          piece_w(i + NS - 1, pr);
          if (pr + 1 < NTILE / 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              wh[nx][t] = *(const bf16x8_t*)(st + rdw + (2 * pr + 2 + t) * 1024);
              wl[nx][t] = *(const bf16x8_t*)(st + rdw + W_PLANE + (2 * pr + 2 + t) * 1024);
            }
          }
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const bf16x8_t w = term == 1 ? wl[cur][t] : wh[cur][t];
              const bf16x8_t a = term == 0 ? al : ah;          // hi*lo, lo*hi, hi*hi
              acc[2 * pr + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[2 * pr + t], 0, 0, 0);
            }
#pragma unroll
          for (int n = 0; n < 6; ++n) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (TIMING) { t_wait += c1 - c0; t_steps += (long long)__builtin_amdgcn_s_memtime() - c1; }
    }

    long long e0 = 0;
    if constexpr (TIMING) e0 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA requests have landed before the ring becomes staging
    n384_epilogue(acc, smem, wave, lane, m0w, rs_c, p.ldc, bias_l, ls_l, p.stats, p.eps, p.M, p.pos, p.npatch, p.ntok_s);
    if constexpr (TIMING) t_epi += (long long)__builtin_amdgcn_s_memtime() - e0;
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
      d[0] = t_wait; d[1] = t_steps; d[2] = t_epi; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
    }
  }
}


// ---- AFRAG: A as fragment-major planes (EPI_GELU_FRAG of gemm_a384_x3.hip), W packed (wvn_pack_n384_x3_weight) --------------------------
// A never touches the LDS: fragment (R, s) of a plane is one contiguous kilobyte, lane-major, so a wave fetches its operand of k-step s
// with ONE coalesced 16-byte load per lane and plane, six k-steps ahead (48 registers).  W: the 24 KB of a k-step (both planes, rows of
// 32 B already swizzled and column-permuted by the packer) are one contiguous block: every DMA piece is one kilobyte of consecutive
// addresses (with 32 scattered rows per piece the request cost twice as much: 2235 cycles per k-step against ~1400).
constexpr int PD = 6;                            // k-steps of A fragments in flight
constexpr int FW = 6;                            // W DMA pieces per wave and k-step
constexpr int FSTAGE = 2 * W_PLANE;              // 24 KB
constexpr int FNS = 5;                           // ring depth (120 KB)
static_assert(FNS * FSTAGE <= RING_BYTES, "the fragment form's ring must fit in front of the bias table");

template <bool TIMING>
__global__ __launch_bounds__(256, 1) void gemm_n384_x3_frag_kernel(N384X3Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = p.K / BKS;                      // a multiple of PD (launcher)
  const float* bias_l = (const float*)(smem + BIAS_OFF);
  const float* ls_l = (const float*)(smem + LS_OFF);
  for (int i = tid; i < NN; i += 256) {
    ((float*)(smem + BIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
    ((float*)(smem + LS_OFF))[i] = p.ls ? p.ls[i] : 1.f;
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)2 * NN * p.K * 2), 0x00020000);
  const size_t mpad = (size_t)(p.M + 31) / 32 * 32;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((p.a_plane + mpad * p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)p.M * p.ldc * 4), 0x00020000);
  const unsigned wv0 = (unsigned)(wave * FW * 1024 + lane * 16);   // + u * 1024: piece wave * 6 + u of the k-step's 24
  const unsigned rdw = l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4);  // + plane * W_PLANE + t * 1024
  const unsigned a_lo_off = (unsigned)(p.a_plane * 2);

  long long t_wait = 0, t_steps = 0, t_epi = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const int nrb = (p.M + BM - 1) / BM;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m0w = rb * BM + wave * 32;
    const unsigned a_base = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(m0w >> 5) * nk * 1024));   // fragment (R, 0) of the hi plane
    auto piece_w = [&](int i, int u) {
      unsigned char* st = smem + (i % FNS) * FSTAGE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(st + (wave * FW + u) * 1024), 16, wv0 + u * 1024,
                                               __builtin_amdgcn_readfirstlane((unsigned)i * FSTAGE), 0, 0);
    };
    u32x4_t afh[PD], afl[PD];
    auto load_a = [&](int i, int slot) {
      const unsigned so = __builtin_amdgcn_readfirstlane(a_base + (unsigned)i * 1024);
      afh[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, so, 0);
      afl[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, so + a_lo_off, 0);
    };
    __syncthreads();  // the previous row block's staging reads are done (and the bias table is visible) before DMA reuses the LDS
    // VM queue order: the A fragments of k-steps 0 .. PD - 1 first, then the W slices 0 .. FNS - 2
#pragma unroll
    for (int j = 0; j < PD; ++j) load_a(j, j);
#pragma unroll
    for (int i0 = 0; i0 < FNS - 1; ++i0)
      if (i0 < nk) {
#pragma unroll
        for (int u = 0; u < FW; ++u) piece_w(i0, u);
      }

    f32x16_t acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int ib = 0; ib < nk; ib += PD) {
#pragma unroll
      for (int jj = 0; jj < PD; ++jj) {
        const int i = ib + jj;
        long long c0 = 0, c1 = 0;
        if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
        // W slice i landed.  Younger than its last piece, in steady state: the A pair of its period and (6 + 2) of each of the FNS - 2 =
        // 3 periods since -- 26; in the first periods fewer ((FNS - 2) * 6 = 18 at i = 0): the constant 18 is safe everywhere but the tail
        // (W requests continue past the last slice, unconditionally: the source offset then lies outside the buffer -- zeros -- and the
        //  stage is never read; drained before the epilogue)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FNS - 2) * FW) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* st = smem + (i % FNS) * FSTAGE;
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, afh[jj]), al = __builtin_bit_cast(bf16x8_t, afl[jj]);
        bf16x8_t wh[2][2], wl[2][2];   // [pair parity][tile of the pair]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          wh[0][t] = *(const bf16x8_t*)(st + rdw + t * 1024);
          wl[0][t] = *(const bf16x8_t*)(st + rdw + W_PLANE + t * 1024);
        }
#pragma unroll
        for (int pr = 0; pr < NTILE / 2; ++pr) {
          const int cur = pr & 1, nx = cur ^ 1;
          piece_w(i + FNS - 1, pr);   // slice i + 4 into the stage every wave has just left
          if (pr + 1 < NTILE / 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              wh[nx][t] = *(const bf16x8_t*)(st + rdw + (2 * pr + 2 + t) * 1024);
              wl[nx][t] = *(const bf16x8_t*)(st + rdw + W_PLANE + (2 * pr + 2 + t) * 1024);
            }
          }
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const bf16x8_t w = term == 1 ? wl[cur][t] : wh[cur][t];
              const bf16x8_t a = term == 0 ? al : ah;          // hi*lo, lo*hi, hi*hi
              acc[2 * pr + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[2 * pr + t], 0, 0, 0);
            }
#pragma unroll
          for (int n = 0; n < 6; ++n) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i + PD < nk) load_a(i + PD, jj);   // (after the last MFMA that reads the slot)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TIMING) { t_wait += c1 - c0; t_steps += (long long)__builtin_amdgcn_s_memtime() - c1; }
      }
    }
    long long e0 = 0;
    if constexpr (TIMING) e0 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA requests have landed before the ring becomes staging
    n384_epilogue(acc, smem, wave, lane, m0w, rs_c, p.ldc, bias_l, ls_l, p.stats, p.eps, p.M, p.pos, p.npatch, p.ntok_s);
    if constexpr (TIMING) t_epi += (long long)__builtin_amdgcn_s_memtime() - e0;
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
      d[0] = t_wait; d[1] = t_steps; d[2] = t_epi; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
    }
  }
}

// ---- the fragment form with a WAVE PAIR per 32 rows (round 5) ------------------------------------------------------------------------------
// Eight waves per workgroup, two per SIMD: wave (rg, ch) owns rows 32 rg .. 32 rg + 31 of the 128-row block and output columns
// 192 ch .. 192 ch + 191 -- six accumulator tiles (96 registers) instead of twelve, 18 MFMAs and THREE DMA pieces per k-step instead of 36
// and six.  The matrix work per SIMD and k-step is unchanged (36 MFMAs = 1152 cycles); what changes is that a wave which is issuing a DMA
// piece (~54 cycles each for a wave that is alone on its SIMD), waiting at the barrier or for a fragment no longer idles the matrix pipe:
// its partner's MFMAs fill it (scripts/ubench/wave_pair.hip: 1675 -> 1333 cycles per k-step for the untuned instruction streams; THIS kernel
// against the tuned one-wave form: 1546 -> 1481 with the barrier of k-step i covering slice i, -> 1355 with it covering slice i + 1 and the
// first fragments of the next k-step fetched under the tail of this one; wall time of fc2 -4 %: the denser MFMA stream clocks lower).  Price:
// both waves of a pair fetch the row group's A fragments (the second fetch hits the L2), and the LayerNorm statistics of a row are the sum
// of two waves' partials (exchanged through LDS, half 0 + half 1: a fixed order).  Same products in the same order per output element:
// C is bit-identical to gemm_n384_x3_frag_kernel's.
constexpr int PFW = 3;                            // W DMA pieces per wave and k-step
constexpr int PNS = 6;                            // ring depth: the barrier at the top of k-step i covers slice i + 1 (its first fragments are fetched under
                                                  // the tail of k-step i), slices i + 2 .. i + 4 in flight, slice i + 5 requested during k-step i
constexpr int PRING_BYTES = PNS * FSTAGE;         // 144 KB
constexpr int PSTG_PITCH = 68;                    // floats per staged row (64 columns + 4)
constexpr int PSTG_BYTES = 32 * PSTG_PITCH * 4;   // per wave: 8704 (the eight images overlap the ring's first 68 KB)
constexpr int PBIAS_OFF = PRING_BYTES;
constexpr int PLS_OFF = PBIAS_OFF + NN * 4;
constexpr int PSTAT_OFF = PLS_OFF + NN * 4;       // [2 column halves][128 rows] {sum, sum of squares}
constexpr int PAIR_LDS_BYTES = PSTAT_OFF + 2 * BM * 8;
static_assert(8 * PSTG_BYTES <= PRING_BYTES && PAIR_LDS_BYTES <= 160 * 1024, "staging images must fit in the ring, the whole in the LDS");

__device__ inline void n384_pair_epilogue(const f32x16_t (&acc)[NTILE / 2], unsigned char* smem, int wave, int lane, int rg, int ch, int m0w,
                                          __amdgpu_buffer_rsrc_t rs_c, int ldc, const float* bias_l, const float* ls_l, float* stats, float eps, int M) {
  const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, rq = lane >> 4;
  __syncthreads();  // every wave is done reading the ring: the staging images overlap it
  float* stg = (float*)(smem + wave * PSTG_BYTES);
  const unsigned cvoff = (unsigned)((rq * ldc + l15 * 4) * 4);
  const int col0 = 192 * ch;
  float s1[8], s2[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) { s1[it] = 0.f; s2[it] = 0.f; }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16_t& a = acc[2 * c + tt];
        const f32x4_t o = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        *(f32x4_t*)(stg + l31 * PSTG_PITCH + 32 * tt + 8 * g + 4 * hi) = o;
      }
    const f32x4_t b4 = *(const f32x4_t*)(bias_l + col0 + 64 * c + l15 * 4);
    const f32x4_t l4 = *(const f32x4_t*)(ls_l + col0 + 64 * c + l15 * 4);
    u32x4_t r[8];
    unsigned so_dst[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {   // the eight row-quad fetches of the column group requested together
      so_dst[it] = __builtin_amdgcn_readfirstlane(((m0w + 4 * it) * ldc + col0 + 64 * c) * 4);
      r[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, cvoff, so_dst[it], 0);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4_t v = *(const f32x4_t*)(stg + (4 * it + rq) * PSTG_PITCH + l15 * 4);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = (v[e] + b4[e]) * l4[e] + __uint_as_float(r[it][e]);
        o[e] = __float_as_uint(y);
        s1[it] += y;
        s2[it] = fmaf(y, y, s2[it]);
      }
      wvn_store_b128_guarded(o, rs_c, cvoff, so_dst[it]);  // rows >= M fall outside num_records: dropped
    }
  }
  if (stats) {   // (uniform) row 4 it + rq of the wave: its 16 lanes hold the partial sums of this column half
    float* sx = (float*)(smem + PSTAT_OFF);
    auto dpp_add = [](float v, auto ctrl) {
      constexpr int C = decltype(ctrl)::value;
      return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), C, 0xf, 0xf, false));
    };
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      float a = s1[it], b = s2[it];
      a = dpp_add(a, std::integral_constant<int, 0xB1>{}); b = dpp_add(b, std::integral_constant<int, 0xB1>{});
      a = dpp_add(a, std::integral_constant<int, 0x4E>{}); b = dpp_add(b, std::integral_constant<int, 0x4E>{});
      a = dpp_add(a, std::integral_constant<int, 0x141>{}); b = dpp_add(b, std::integral_constant<int, 0x141>{});
      a = dpp_add(a, std::integral_constant<int, 0x140>{}); b = dpp_add(b, std::integral_constant<int, 0x140>{});
      if (l15 == 0) *(wvn_f32x2_t*)(sx + ((size_t)ch * BM + rg * 32 + 4 * it + rq) * 2) = wvn_f32x2_t{a, b};
    }
    __syncthreads();
    if (ch == 0 && lane < 32) {
      const int m = m0w + lane;
      if (m < M) {
        const wvn_f32x2_t p0 = *(const wvn_f32x2_t*)(sx + ((size_t)rg * 32 + lane) * 2), p1 = *(const wvn_f32x2_t*)(sx + ((size_t)BM + rg * 32 + lane) * 2);
        const float a = p0[0] + p1[0], b = p0[1] + p1[1];
        const float mean = a * (1.0f / NN);
        const float var = fmaxf(b * (1.0f / NN) - mean * mean, 0.f);
        *(wvn_f32x2_t*)(stats + 2 * (size_t)m) = wvn_f32x2_t{mean, 1.0f / sqrtf(var + eps)};
      }
    }
  }
}

template <bool TIMING>
__global__ __launch_bounds__(512, 1) void gemm_n384_x3_frag_pair_kernel(N384X3Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, ch = wave >> 2;        // (waves w and w + 4 -- the two column halves of a row group -- share a SIMD)
  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = p.K / BKS;                      // a multiple of PD (launcher)
  const float* bias_l = (const float*)(smem + PBIAS_OFF);
  const float* ls_l = (const float*)(smem + PLS_OFF);
  for (int i = tid; i < NN; i += 512) {
    ((float*)(smem + PBIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
    ((float*)(smem + PLS_OFF))[i] = p.ls ? p.ls[i] : 1.f;
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)2 * NN * p.K * 2), 0x00020000);
  const size_t mpad = (size_t)(p.M + 31) / 32 * 32;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((p.a_plane + mpad * p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)p.M * p.ldc * 4), 0x00020000);
  const unsigned wv0 = (unsigned)(wave * PFW * 1024 + lane * 16);   // + u * 1024: piece wave * 3 + u of the k-step's 24
  const unsigned rdw = l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4) + ch * 6 * 1024;  // + plane * W_PLANE + t * 1024: this wave's six column tiles
  const unsigned a_lo_off = (unsigned)(p.a_plane * 2);

  long long t_wait = 0, t_steps = 0, t_epi = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const int nrb = (p.M + BM - 1) / BM;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m0w = rb * BM + rg * 32;
    const unsigned a_base = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(m0w >> 5) * nk * 1024));   // fragment (R, 0) of the hi plane
    auto piece_w = [&](int i, int u) {
      unsigned char* st = smem + (i % PNS) * FSTAGE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(st + (wave * PFW + u) * 1024), 16, wv0 + u * 1024,
                                               __builtin_amdgcn_readfirstlane((unsigned)i * FSTAGE), 0, 0);
    };
    u32x4_t afh[PD], afl[PD];
    auto load_a = [&](int i, int slot) {
      const unsigned so = __builtin_amdgcn_readfirstlane(a_base + (unsigned)i * 1024);
      afh[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, so, 0);
      afl[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, so + a_lo_off, 0);
    };
    __syncthreads();  // the previous row block's staging / statistics reads are done (and the bias table is visible) before DMA reuses the LDS
#pragma unroll
    for (int j = 0; j < PD; ++j) load_a(j, j);
#pragma unroll
    for (int i0 = 0; i0 < PNS - 1; ++i0)
      if (i0 < nk) {
#pragma unroll
        for (int u = 0; u < PFW; ++u) piece_w(i0, u);
      }

    f32x16_t acc[NTILE / 2];
#pragma unroll
    for (int t = 0; t < NTILE / 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // slice 0 landed (younger: the pieces of slices 1 .. PNS - 2), every wave's part of it: its first tile pair into slot 0
    bf16x8_t wh[2][2], wl[2][2];   // [slot][tile of the pair]; the first pair of k-step i sits in slot i & 1 (three pairs per k-step: the slots alternate)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PNS - 2) * PFW) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      wh[0][t] = *(const bf16x8_t*)(smem + rdw + t * 1024);
      wl[0][t] = *(const bf16x8_t*)(smem + rdw + W_PLANE + t * 1024);
    }
    for (int ib = 0; ib < nk; ib += PD) {
#pragma unroll
      for (int jj = 0; jj < PD; ++jj) {
        const int i = ib + jj;
        long long c0 = 0, c1 = 0;
        if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
        // slice i + 1 landed: younger than its last piece are at least the (PNS - 3) * PFW pieces of the three slices since.  The barrier also says
        // that every wave has left k-step i - 1: the stage of slice i - 1 is free for slice i + 5
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PNS - 3) * PFW) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* st = smem + (i % PNS) * FSTAGE;
        const unsigned char* st1 = smem + ((i + 1) % PNS) * FSTAGE;
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, afh[jj]), al = __builtin_bit_cast(bf16x8_t, afl[jj]);
        const int s0 = jj & 1;   // (PD is even: the slot parity of a k-step is a constant of the unrolled body)
#pragma unroll
        for (int pr = 0; pr < NTILE / 4; ++pr) {
          const int cur = (s0 + pr) & 1, nx = cur ^ 1;
          piece_w(i + PNS - 1, pr);   // slice i + 5 into the stage every wave has left: one piece per tile pair
          if (pr + 1 < NTILE / 4) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              wh[nx][t] = *(const bf16x8_t*)(st + rdw + (2 * pr + 2 + t) * 1024);
              wl[nx][t] = *(const bf16x8_t*)(st + rdw + W_PLANE + (2 * pr + 2 + t) * 1024);
            }
          } else {   // the FIRST pair of the next k-step (its slice is visible since this k-step's barrier): no LDS round trip behind the next barrier
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              wh[nx][t] = *(const bf16x8_t*)(st1 + rdw + t * 1024);
              wl[nx][t] = *(const bf16x8_t*)(st1 + rdw + W_PLANE + t * 1024);
            }
          }
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const bf16x8_t w = term == 1 ? wl[cur][t] : wh[cur][t];
              const bf16x8_t a = term == 0 ? al : ah;          // hi*lo, lo*hi, hi*hi
              acc[2 * pr + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, acc[2 * pr + t], 0, 0, 0);
            }
#pragma unroll
          for (int n = 0; n < 6; ++n) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i + PD < nk) load_a(i + PD, jj);   // (after the last MFMA that reads the slot)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TIMING) { t_wait += c1 - c0; t_steps += (long long)__builtin_amdgcn_s_memtime() - c1; }
      }
    }
    long long e0 = 0;
    if constexpr (TIMING) e0 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA requests have landed before the ring becomes staging
    n384_pair_epilogue(acc, smem, wave, lane, rg, ch, m0w, rs_c, p.ldc, bias_l, ls_l, p.stats, p.eps, p.M);
    if constexpr (TIMING) t_epi += (long long)__builtin_amdgcn_s_memtime() - e0;
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg && ch == 0) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + rg) * 4;
      d[0] = t_wait; d[1] = t_steps; d[2] = t_epi; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
    }
  }
}

// ---- MX correction terms (round 6): hi * hi on fp16 MFMAs, the two correction products on v_mfma_scale_f32_32x32x64_f8f6f4 ------------------
// The <= 1e-3 mode needs every linear's operand rounding compensated (scripts/error_budget_r6.py); the bf16 x 3 form above pays three full-rate
// MFMAs per fragment pair for it.  Here an operand v is  h = fp16(v),  l8 = e5m2((v - h) * 2^12),  h8 = e5m2(v)  and
//     sum a w  ~=  sum a_h w_h  +  2^-12 (sum a_h8 w_l8 + sum a_l8 w_h8):
// per 64 k of a 32 x 32 tile four v_mfma_f32_32x32x16_f16 (128 cycles) and TWO scaled 8-bit MFMAs of K = 64 (2 x 64 cycles) instead of twelve
// bf16 MFMAs (384): two thirds of the matrix-pipe time.  e5m2 IS the top byte of an fp16, so the 8-bit planes need no block scales: the
// residues carry ONE constant factor 2^12 (undone by the instruction's E8M0 scale operand, 115 = 2^-12, identical in every lane and block:
// the hardware's assignment of scale bytes to 32-k blocks -- registers 0-3 of both half-waves = block 0, scripts/ubench/mx_formats.hip --
// never matters), and |l8| <= |v| keeps them inside the format wherever v fits fp16.  What the correction products lose is the 2-bit
// significand of their operands: (2^-12 / sqrt 3) * ~7 % of a product -- tokens 2.9e-4 / 3.4e-4 from the fp32 oracle in the emulation
// (synthetic / the reference's real frame; the bf16 x 3 form: 1.1e-4 / 2.3e-4, both dominated by the fp16 attention operands).
//
// Structure = gemm_n384_x3_frag_pair_kernel (wave pair per 32 rows, 6-deep ring of 24 KB stages, three DMA pieces per wave and stage), with
// FOUR stage kinds per 64 k (the packed weight of backbone.pack_n384_mx lists them in this order, every stage 24 x 1 KB blocks [x][tile]):
//   0: W_h of k-steps 0 | 1 (x = k-step)  -> 2 x 6 fp16 MFMAs per wave      2: W_l8 (x = 16-byte half of the lane's 32 bytes) -> 6 scaled MFMAs on a_h8
//   1: W_h of k-steps 2 | 3                                                 3: W_h8                                           -> 6 scaled MFMAs on a_l8
// -- 384 matrix-pipe cycles per wave and stage whatever its kind (576 in the bf16 x 3 form), the same LDS reads (12 per wave and stage) and
// DMA pieces.  A: fragment-major again, per row group R and 64-k step c four 1 KB fp16 fragments (hi plane, as EPI_GELU_FRAG) and two 1 KB
// halves in each of the two 8-bit planes: lane (row, h) holds bytes (s', j) = k-step s' of the four, element j of its eight -- the order the
// producers' accumulators hold them; two 64-k steps in flight (64 registers).
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
constexpr int MXD = 2;                            // 64-k steps of A in flight
constexpr int MX_SC_ONE = 0x7f7f7f7f, MX_SC_RES = 0x73737373;   // E8M0 scale bytes: 2^0, 2^-12 (all four bytes alike: op_sel never matters)

struct N384MXExtra { unsigned a_l8_off; };   // byte offset of the l8 plane behind the fp16 plane (one buffer descriptor)

// VAR (timing experiments only, results are garbage for VAR != 0): 1 = no W DMA inside the loop, 2 = no A loads inside the loop, 3 = neither,
// 4 = no barrier / wait at the stage boundaries
template <bool TIMING, int VAR = 0>
__global__ __launch_bounds__(512, 1) void gemm_n384_mx_pair_kernel(N384X3Params p, N384MXExtra ex) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, ch = wave >> 2;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = p.K / BKS;                      // stages: four per 64 k; a multiple of 8 (launcher)
  const int nmac = p.K / 64;
  const float* bias_l = (const float*)(smem + PBIAS_OFF);
  const float* ls_l = (const float*)(smem + PLS_OFF);
  for (int i = tid; i < NN; i += 512) {
    ((float*)(smem + PBIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;
    ((float*)(smem + PLS_OFF))[i] = p.ls ? p.ls[i] : 1.f;
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)4 * NN * p.K), 0x00020000);
  const size_t mpad = (size_t)(p.M + 31) / 32 * 32;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)(ex.a_l8_off + mpad * p.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)p.M * p.ldc * 4), 0x00020000);
  const unsigned wv0 = (unsigned)(wave * PFW * 1024 + lane * 16);
  const unsigned rdw = l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4) + ch * 6 * 1024;  // + x * W_PLANE + t * 1024: this wave's six column tiles

  long long t_wait = 0, t_steps = 0, t_epi = 0;
  const long long t_start = TIMING ? (long long)__builtin_amdgcn_s_memtime() : 0;
  const int nrb = (p.M + BM - 1) / BM;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m0w = rb * BM + rg * 32;
    const unsigned a_base = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(m0w >> 5) * p.K * 64));    // fragment (R, 0) of the fp16 plane
    const unsigned a8_base = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(m0w >> 5) * p.K * 32));   // (R, 0) of an 8-bit plane
    auto piece_w = [&](int i, int u) {
      unsigned char* st = smem + (i % PNS) * FSTAGE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(st + (wave * PFW + u) * 1024), 16, wv0 + u * 1024,
                                               __builtin_amdgcn_readfirstlane((unsigned)i * FSTAGE), 0, 0);
    };
    u32x4_t ah[MXD][4], a8[MXD][2][2];   // a8[slot][0 = h8 | 1 = l8][half]
    auto load_ah = [&](int c, int slot) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        ah[slot][s] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, __builtin_amdgcn_readfirstlane(a_base + (unsigned)(c * 4 + s) * 1024), 0);
    };
    auto load_a8 = [&](int c, int slot) {   // the l8 plane (the h8 operand is derived from the fp16 fragments: derive_h8)
#pragma unroll
      for (int x = 0; x < 2; ++x)
        a8[slot][1][x] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, lane * 16, __builtin_amdgcn_readfirstlane(a8_base + ex.a_l8_off + (unsigned)(c * 2 + x) * 1024), 0);
    };
    // a_h8 = e5m2(a_h): sixteen v_cvt_scalef32_pk_bf8_f16 per 64 k and lane, in VALU slots this kernel does not use otherwise -- instead of a third plane
    // in HBM (a quarter of the A stream, which is what the stage waits for: profiles/r06_mx_kernels.md) and a conversion + two stores per tile in the producers
    auto derive_h8 = [&](int slot) {
      typedef __attribute__((ext_vector_type(2))) short s16x2_t;
      typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
#pragma unroll
      for (int sfr = 0; sfr < 4; ++sfr) {
        uint32_t d[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t pr = ah[slot][sfr][e];   // (scalar copy: clang's bit_cast of a vector element reads element 0)
          const s16x2_t o = __builtin_bit_cast(s16x2_t, d[e >> 1]);
          d[e >> 1] = __builtin_bit_cast(uint32_t, (e & 1) ? __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f, true)
                                                           : __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(o, __builtin_bit_cast(h2_t, pr), 1.0f, false));
        }
        a8[slot][0][sfr >> 1][2 * (sfr & 1)] = d[0];
        a8[slot][0][sfr >> 1][2 * (sfr & 1) + 1] = d[1];
      }
    };
    __syncthreads();  // the previous row block's staging / statistics reads are done (and the bias table is visible) before DMA reuses the LDS
    // (the order is pinned: hipcc counts the operations behind a register load to size the vmcnt wait in front of its first use, and takes the
    //  minimum over the loop's two entry edges -- left to the scheduler, the operands of stage 0 were requested LAST and every trip waited for
    //  all but the 15 youngest operations, i.e. for A fragments requested two stages earlier from HBM)
#pragma unroll
    for (int c = 0; c < MXD; ++c) {
      load_ah(c, c);
      __builtin_amdgcn_sched_barrier(0);
      load_a8(c, c);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i0 = 0; i0 < PNS - 1; ++i0)
      if (i0 < nk) {
#pragma unroll
        for (int u = 0; u < PFW; ++u) piece_w(i0, u);
      }

    f32x16_t acc[NTILE / 2];
#pragma unroll
    for (int t = 0; t < NTILE / 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    u32x4_t wq[2][2][2];   // [slot][x][tile of the pair]; the first pair of stage i sits in slot i & 1 (three pairs per stage: the slots alternate)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PNS - 2) * PFW) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int t = 0; t < 2; ++t) wq[0][x][t] = *(const u32x4_t*)(smem + rdw + x * W_PLANE + t * 1024);
    for (int ib = 0; ib < nk; ib += 4 * MXD) {
#pragma unroll
      for (int jj = 0; jj < 4 * MXD; ++jj) {
        const int i = ib + jj;
        const int kind = jj & 3, slot = jj >> 2;   // (compile-time constants of the unrolled body)
        long long c0 = 0, c1 = 0;
        if constexpr (TIMING) c0 = (long long)__builtin_amdgcn_s_memtime();
        // (the barrier also says that every wave has left stage i - 1: its ring slot is free)
        // slice i + 1 landed: younger than its last piece are EXACTLY the (PNS - 3) * PFW pieces of the three stages since and the A loads those
        // stages ended with (kind 1: four, kinds 2 / 3: two each; requested unconditionally -- past the last 64-k step the offsets fall outside the
        // buffer and return zeros into registers nobody reads -- so that the count is a constant of the stage kind: waiting for "all but nine"
        // also waited for A fragments requested from HBM two stages earlier, 125 cycles per stage)
        if constexpr (VAR != 4) {
          // A loads per stage kind: 0 / 4 (fp16 fragments) / 0 / 2 (l8 halves); of the three stages before a stage of kind 0 / 1 / 2 / 3: 6 / 2 / 6 / 4 -- behind
          // the prologue (whose A loads all precede its pieces) stages 0 / 1 / 2 see 0 / 0 / 4 (uniform branches at a stage boundary, where the barrier ends
          // the scheduling region anyway)
          constexpr int W3 = (PNS - 3) * PFW;
          if (jj == 0) { if (i == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 6) : "memory"); }
          else if (jj == 1) { if (i == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 2) : "memory"); }
          else if (jj == 2) { if (i == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 4) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 6) : "memory"); }
          else if ((jj & 3) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 6) : "memory");
          else if ((jj & 3) == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 2) : "memory");
          else if ((jj & 3) == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 6) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W3 + 4) : "memory");
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (TIMING) c1 = (long long)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* st = smem + (i % PNS) * FSTAGE;
        const unsigned char* st1 = smem + ((i + 1) % PNS) * FSTAGE;
        const int s0 = jj & 1;
#pragma unroll
        for (int pr = 0; pr < NTILE / 4; ++pr) {
          const int cur = (s0 + pr) & 1, nx = cur ^ 1;
          if constexpr (!(VAR & 1) || VAR == 4) piece_w(i + PNS - 1, pr);   // stage i + 5 into the slot every wave has left: one piece per tile pair
          const unsigned char* src = pr + 1 < NTILE / 4 ? st + (2 * pr + 2) * 1024 : st1;   // (the last pair's partner: the FIRST pair of the next stage)
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int t = 0; t < 2; ++t) wq[nx][x][t] = *(const u32x4_t*)(src + rdw + x * W_PLANE + t * 1024);
          if (kind == 1 && pr == 0) derive_h8(slot);   // (the fp16 fragments of this 64-k step are in: stage kind 0 has used them; VALU work beside this stage's MFMAs)
          if (kind < 2) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                acc[2 * pr + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wq[cur][x][t]), __builtin_bit_cast(f16x8_t, ah[slot][2 * kind + x]),
                                                                         acc[2 * pr + t], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
          } else {
            const u32x4_t a0 = a8[slot][kind - 2][0], a1 = a8[slot][kind - 2][1];
            const i32x8_t av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const u32x4_t w0 = wq[cur][0][t], w1 = wq[cur][1][t];
              const i32x8_t wv = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
              // kind 2: W_l8 (carries 2^12) x a_h8;  kind 3: W_h8 x a_l8 (carries 2^12)
              acc[2 * pr + t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, av, acc[2 * pr + t], 1, 1, 0, kind == 2 ? MX_SC_RES : MX_SC_ONE, 0,
                                                                                kind == 2 ? MX_SC_ONE : MX_SC_RES);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // the operands of 64-k step c + MXD into the registers this stage has just finished with
        const int c2 = (i >> 2) + MXD;
        if constexpr (!(VAR & 2) || VAR == 4) {
          if (kind == 1) load_ah(c2, slot);
          if (kind == 3) load_a8(c2, slot);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TIMING) { t_wait += c1 - c0; t_steps += (long long)__builtin_amdgcn_s_memtime() - c1; }
      }
    }
    long long e0 = 0;
    if constexpr (TIMING) e0 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA requests have landed before the ring becomes staging
    n384_pair_epilogue(acc, smem, wave, lane, rg, ch, m0w, rs_c, p.ldc, bias_l, ls_l, p.stats, p.eps, p.M);
    if constexpr (TIMING) t_epi += (long long)__builtin_amdgcn_s_memtime() - e0;
  }
  if constexpr (TIMING) {
    if (lane == 0 && p.dbg && ch == 0) {
      long long* d = p.dbg + ((size_t)blockIdx.x * 4 + rg) * 4;
      d[0] = t_wait; d[1] = t_steps; d[2] = t_epi; d[3] = (long long)__builtin_amdgcn_s_memtime() - t_start;
    }
  }
}

int g_n384_mx_var = 0;   // timing experiments of the instrumented MX build (wvn_debug_n384_pair(16 + var))
int g_n384_pair = 1;   // 1: the wave-pair form of the fragment kernel (default since its barrier covers the NEXT slice: 1355 against 1546 cycles per k-step and
                       // SIMD, fc2 -4 % wall time), 0: one wave per SIMD (round 4; wvn_debug_n384_pair).  scripts/bench_n384_pair.py, profiles/r05_wave_pair.md

int n384x3_num_cus() {
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

// Eligibility: N == 384, K % 32 == 0, residual epilogue (optional LayerScale), stacked planes, 16-byte aligned operands, 31-bit byte
// offsets; WVN_ERR_ARG otherwise (the caller uses the tiled gemm_x3 kernel).
int wvn_gemm_n384_x3_launch(const GemmBf16Params& g, int epi, hipStream_t st) {
  const bool patch = epi == EPI_PATCH;   // C = A W^T + bias + pos into the token rows (see N384X3Params)
  if (epi != EPI_RESID_F32 && epi != EPI_ACCUM_F32 && !patch) return WVN_ERR_ARG;
  if (patch && (!g.pos || g.npatch <= 0 || (g.npatch & 1) || g.ntok_s < g.npatch + 1 || g.ldc != NN || g.ls || (g.M % g.npatch) != 0)) return WVN_ERR_ARG;
  if (g.N != NN || g.K <= 0 || (g.K % BKS) != 0 || g.M <= 0 || !g.A || !g.A_lo || !g.W || !g.W_lo || !g.C) return WVN_ERR_ARG;
  if ((g.lda % 8) || (g.ldw % 8) || (g.ldc % 4) || g.ldw != g.K) return WVN_ERR_ARG;
  if (((uintptr_t)g.A | (uintptr_t)g.A_lo | (uintptr_t)g.W | (uintptr_t)g.W_lo | (uintptr_t)g.C) & 15) return WVN_ERR_ARG;
  if (g.A_lo <= g.A || g.W_lo <= g.W) return WVN_ERR_ARG;
  const size_t a_plane = (size_t)(g.A_lo - g.A), w_plane = (size_t)(g.W_lo - g.W);
  const size_t c_rows = patch ? (size_t)(g.M / g.npatch) * g.ntok_s : (size_t)g.M;
  if ((a_plane + (size_t)g.M * g.lda) * 2 >= (1ull << 32) || c_rows * g.ldc * 4 >= (1ull << 32) || (w_plane + (size_t)NN * g.ldw) * 2 >= (1ull << 32))
    return WVN_ERR_ARG;
  N384X3Params p{};
  p.A = g.A; p.a_plane = a_plane; p.lda = g.lda; p.W = g.W; p.w_plane = w_plane; p.ldw = g.ldw; p.bias = g.bias; p.ls = g.ls;
  p.C = (float*)g.C; p.ldc = g.ldc; p.M = g.M; p.K = g.K; p.dbg = g.dbg; p.stats = g.ln_stats_out; p.eps = g.ln_eps;
  if (patch) { p.pos = g.pos; p.npatch = g.npatch; p.ntok_s = g.ntok_s; p.c_rows = (int)c_rows; p.stats = nullptr; }
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(LDS_BYTES, (const void*)gemm_n384_x3_kernel<false>, (const void*)gemm_n384_x3_kernel<true>)) return rc;
  const int ncu = n384x3_num_cus(), nrb = ceil_div(g.M, BM);
  const int grid = nrb < ncu ? nrb : ncu;
  if (p.dbg) hipLaunchKernelGGL(gemm_n384_x3_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, st, p);
  else hipLaunchKernelGGL(gemm_n384_x3_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// A: fragment-major planes ([2][ceil(M / 32)][K / 16][64 lanes][8], lo plane a_plane elements behind the hi plane); W: the packed weight of
// wvn_pack_n384_x3_weight (2 * 384 * K elements).  K % 96 == 0 (six k-steps of A in flight, unrolled).
int wvn_gemm_n384_x3_frag_launch(const GemmBf16Params& g, int epi, hipStream_t st) {
  if (epi != EPI_RESID_F32 && epi != EPI_ACCUM_F32) return WVN_ERR_ARG;
  if (g.N != NN || g.K <= 0 || (g.K % (BKS * PD)) != 0 || g.M <= 0 || !g.A || !g.A_lo || !g.W || !g.C || (g.ldc % 4)) return WVN_ERR_ARG;
  if (((uintptr_t)g.A | (uintptr_t)g.A_lo | (uintptr_t)g.W | (uintptr_t)g.C) & 15) return WVN_ERR_ARG;
  if (g.A_lo <= g.A) return WVN_ERR_ARG;
  const size_t a_plane = (size_t)(g.A_lo - g.A), mpad = (size_t)(g.M + 31) / 32 * 32;
  if ((a_plane + mpad * g.K) * 2 >= (1ull << 32) || (size_t)g.M * g.ldc * 4 >= (1ull << 32) || (size_t)2 * NN * g.K * 2 >= (1ull << 31)) return WVN_ERR_ARG;
  N384X3Params p{};
  p.A = g.A; p.a_plane = a_plane; p.lda = g.K; p.W = g.W; p.w_plane = 0; p.ldw = g.K; p.bias = g.bias; p.ls = g.ls;
  p.C = (float*)g.C; p.ldc = g.ldc; p.M = g.M; p.K = g.K; p.dbg = g.dbg; p.stats = g.ln_stats_out; p.eps = g.ln_eps;
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(PAIR_LDS_BYTES, (const void*)gemm_n384_x3_frag_kernel<false>, (const void*)gemm_n384_x3_frag_kernel<true>,
                                (const void*)gemm_n384_x3_frag_pair_kernel<false>, (const void*)gemm_n384_x3_frag_pair_kernel<true>)) return rc;
  const int ncu = n384x3_num_cus(), nrb = ceil_div(g.M, BM);
  const int grid = nrb < ncu ? nrb : ncu;
  if (g_n384_pair) {
    if (p.dbg) hipLaunchKernelGGL(gemm_n384_x3_frag_pair_kernel<true>, dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p);
    else hipLaunchKernelGGL(gemm_n384_x3_frag_pair_kernel<false>, dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p);
  } else if (p.dbg) hipLaunchKernelGGL(gemm_n384_x3_frag_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, st, p);
  else hipLaunchKernelGGL(gemm_n384_x3_frag_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
// MX form (see gemm_n384_mx_pair_kernel): A = fragment-major fp16 plane (g.A) + the l8 plane (g.A_lo, behind g.A within 4 GB; the h8 operand is derived in
// registers); W = backbone.pack_n384_mx's stage list (4 * 384 * K bytes).  K % 128 == 0.
int wvn_gemm_n384_mx_launch(const GemmBf16Params& g, int epi, hipStream_t st) {
  if (epi != EPI_RESID_F32 && epi != EPI_ACCUM_F32) return WVN_ERR_ARG;
  if (g.N != NN || g.K <= 0 || (g.K % 128) != 0 || g.M <= 0 || !g.A || !g.A_lo || !g.W || !g.C || (g.ldc % 4)) return WVN_ERR_ARG;
  if (((uintptr_t)g.A | (uintptr_t)g.A_lo | (uintptr_t)g.W | (uintptr_t)g.C) & 15) return WVN_ERR_ARG;
  const size_t mpad = (size_t)(g.M + 31) / 32 * 32;
  const uintptr_t a0 = (uintptr_t)g.A, al = (uintptr_t)g.A_lo;
  if (al < a0 + mpad * g.K * 2) return WVN_ERR_ARG;
  const uintptr_t top = al + mpad * g.K;
  if (top - a0 >= (1ull << 32) || (size_t)g.M * g.ldc * 4 >= (1ull << 32) || (size_t)4 * NN * g.K >= (1ull << 31)) return WVN_ERR_ARG;
  N384X3Params p{};
  p.A = g.A; p.a_plane = 0; p.lda = g.K; p.W = g.W; p.w_plane = 0; p.ldw = g.K; p.bias = g.bias; p.ls = g.ls;
  p.C = (float*)g.C; p.ldc = g.ldc; p.M = g.M; p.K = g.K; p.dbg = g.dbg; p.stats = g.ln_stats_out; p.eps = g.ln_eps;
  N384MXExtra ex{(unsigned)(al - a0)};
  static LdsOptIn lds_opt_in;
  if (const int rc = lds_opt_in(PAIR_LDS_BYTES, (const void*)gemm_n384_mx_pair_kernel<false>, (const void*)gemm_n384_mx_pair_kernel<true>,
                                (const void*)gemm_n384_mx_pair_kernel<true, 1>, (const void*)gemm_n384_mx_pair_kernel<true, 2>,
                                (const void*)gemm_n384_mx_pair_kernel<true, 3>, (const void*)gemm_n384_mx_pair_kernel<true, 4>)) return rc;
  const int ncu = n384x3_num_cus(), nrb = ceil_div(g.M, BM);
  const int grid = nrb < ncu ? nrb : ncu;
  if (p.dbg && g_n384_mx_var == 1) hipLaunchKernelGGL((gemm_n384_mx_pair_kernel<true, 1>), dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p, ex);
  else if (p.dbg && g_n384_mx_var == 2) hipLaunchKernelGGL((gemm_n384_mx_pair_kernel<true, 2>), dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p, ex);
  else if (p.dbg && g_n384_mx_var == 3) hipLaunchKernelGGL((gemm_n384_mx_pair_kernel<true, 3>), dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p, ex);
  else if (p.dbg && g_n384_mx_var == 4) hipLaunchKernelGGL((gemm_n384_mx_pair_kernel<true, 4>), dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p, ex);
  else if (p.dbg) hipLaunchKernelGGL(gemm_n384_mx_pair_kernel<true>, dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p, ex);
  else hipLaunchKernelGGL(gemm_n384_mx_pair_kernel<false>, dim3(grid), dim3(512), PAIR_LDS_BYTES, st, p, ex);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
void wvn_gemm_n384_x3_set_pair(int on) {
  if (on >= 16) { g_n384_mx_var = on - 16; return; }
  g_n384_pair = on ? 1 : 0;
}
