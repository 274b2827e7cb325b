// fp8 (OCP e4m3) MFMA GEMM with both operands streamed into LDS by direct-to-LDS loads: the long-K linear of ViT-Base (fc2: K = 3072, N = 768) in WVN_PREC_FP8.
//     C (fp32) (+)= ((A_q[M,K] * W_q[N,K]^T) * {sa[m] | MX block scales of A} * sw[n] + bias[n]) (* ls[n])
//
// Why: gemm_fp8.hip moves every K-tile global -> registers -> LDS behind two workgroup barriers and keeps 186 - 245 registers: two workgroups per CU, 0.14 - 0.15 of
// the matrix pipe on this shape, and 1032 tiles on 512 workgroup slots are two rounds and EIGHT tiles.  Here
//   * the same 128 x 128 output tile and 2 x 2 waves of 64 x 64, but K-tiles of 64 bytes (16 KB of operands) through a THREE-stage ring filled by
//     buffer_load ... lds: no staging registers, no ds_write, ONE barrier per K-tile, the K-tile two ahead requested while the current one is multiplied;
//   * LDS rows of 64 B with the four 16-byte chunks XOR-swizzled by (row >> 2) & 3 -- applied on the SOURCE side of the DMA (lane -> (row, physical chunk)), so a piece
//     is 16 rows x 64 B and a fragment read (lane = row, bytes [16 hi, +16) and [32 + 16 hi, +16): the mapping under which the instruction's scale block kb is
//     k [32 kb, 32 kb + 32)) is conflict-free;
//   * <= 168 registers and 50 KB of LDS: THREE workgroups per CU = 768 slots: the 1032 tiles are two rounds, and three waves per SIMD cover one another's DMA issue;
//   * AMX: the activation's E8M0 block scales (gemm_a768_fp8.hip's GELU epilogue) -- a lane's 16 scale bytes of eight K-tiles arrive as one 16-byte load per
//     64 x 32-row block, one group ahead.
// K % 512 == 0, N % 128 == 0, lda / ldw % 16 == 0; epilogues EPI_F32 and EPI_RESID_F32 (+ LayerScale); WVN_ERR_ARG otherwise (the caller uses gemm_fp8.hip).
#include <stdlib.h>

#include "common.h"
#include "wvn_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NS = 3;
constexpr int OP_BYTES = BM * BK;            // 8 KB: one operand of one K-tile
constexpr int STAGE = 2 * OP_BYTES;          // A | W
constexpr int LDS_BYTES = NS * STAGE;        // 48 KB
constexpr int GRP = 8;                       // K-tiles per scale group (16 scale bytes per row)

struct DmaParams {
  const unsigned char* A; int lda; const unsigned char* W; int ldw;
  const float* sa; const unsigned char* a_scales; const float* sw; const float* bias; const float* ls;
  float* C; int ldc; int M, N, K;
};

template <bool RESID, bool AMX>
__global__ __launch_bounds__(256, 3) void gemm_fp8_dma_kernel(DmaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = p.K / BK;

  // ---- producer: a K-tile of an operand = eight pieces of 16 rows x 64 B; wave w copies pieces 2 w, 2 w + 1 of A and of W.  Lane -> (row = 16 piece + (lane >> 2),
  // physical chunk lane & 3) holding logical chunk (lane & 3) ^ ((row >> 2) & 3); rows past M / N read zeros through the descriptor's bound ----
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((size_t)p.M * p.lda), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.ldw), 0x00020000);
  unsigned voa[2], vow[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 16 * (2 * wave + u) + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    voa[u] = (unsigned)((size_t)(m0 + row) * p.lda + c * 16);
    vow[u] = (unsigned)((size_t)(n0 + row) * p.ldw + c * 16);
  }
  auto issue = [&](int kt) __attribute__((always_inline)) {
    unsigned char* dst = smem + (kt % NS) * STAGE + (2 * wave) * 1024;
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(kt * BK));
    // (the instruction's immediate offset would be added to the GLOBAL address as well as the LDS one: every piece gets its own LDS base instead)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)dst, 16, voa[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(dst + 1024), 16, voa[1], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + OP_BYTES), 16, vow[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr)(dst + OP_BYTES + 1024), 16, vow[1], so, 0, 0);
  };

  // ---- consumer addressing: lane (row, hi) reads logical chunks hi and 2 + hi of its row ----
  unsigned ra[2][2], rw[2][2];   // [i | j][first | second half of the operand]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + l31, f = (r >> 2) & 3;
    ra[i][0] = (unsigned)(r * 64 + ((hi ^ f) << 4));
    ra[i][1] = (unsigned)(r * 64 + (((2 + hi) ^ f) << 4));
    const int q = wn * 64 + i * 32 + l31, g = (q >> 2) & 3;
    rw[i][0] = (unsigned)(OP_BYTES + q * 64 + ((hi ^ g) << 4));
    rw[i][1] = (unsigned)(OP_BYTES + q * 64 + (((2 + hi) ^ g) << 4));
  }

  // AMX: the 16 scale bytes of a group of eight K-tiles for the lane's rows (i = 0 | 1), one group ahead
  u32x4_t scn[2] = {{0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu}, {0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu}}, scc[2] = {scn[0], scn[1]};
  const int nblk = p.K / 32;
  const unsigned char* srow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) srow[i] = AMX ? p.a_scales + (size_t)min(m0 + wm * 64 + i * 32 + l31, p.M - 1) * nblk : nullptr;
  auto load_scales = [&](int grp) __attribute__((always_inline)) {
    if constexpr (AMX) {
#pragma unroll
      for (int i = 0; i < 2; ++i) scn[i] = *(const u32x4_t*)(srow[i] + grp * 16);
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_scales(0);
  issue(0);
  issue(1);
  if constexpr (AMX) { scc[0] = scn[0]; scc[1] = scn[1]; }

  const int ngrp = nk / GRP;
  for (int g = 0; g < ngrp; ++g) {
#pragma unroll
    for (int t = 0; t < GRP; ++t) {
      const int kt = g * GRP + t;
      // this wave's pieces of K-tile kt have landed: everything but the four pieces of kt + 1 (and, right behind a group's first K-tile, the two scale loads
      // requested between them)
      if (AMX && t == 1 && g + 1 < ngrp) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t == 0 && g + 1 < ngrp) load_scales(g + 1);
      issue(kt + 2);   // (past the last K-tile: the descriptor returns zeros into a stage nobody reads)
      const unsigned char* st = smem + (kt % NS) * STAGE;
      i32x8_t af[2], bf[2];
      int sca[2] = {0x7f7f7f7f, 0x7f7f7f7f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4_t lo = *(const u32x4_t*)(st + ra[i][0]), h4 = *(const u32x4_t*)(st + ra[i][1]);
        af[i] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)h4[0], (int)h4[1], (int)h4[2], (int)h4[3]};
        const u32x4_t lw = *(const u32x4_t*)(st + rw[i][0]), hw = *(const u32x4_t*)(st + rw[i][1]);
        bf[i] = i32x8_t{(int)lw[0], (int)lw[1], (int)lw[2], (int)lw[3], (int)hw[0], (int)hw[1], (int)hw[2], (int)hw[3]};
        if constexpr (AMX) sca[i] = (int)(scc[i][t >> 1] >> (8 * (2 * (t & 1) + hi)));   // byte 2 t + hi of the group's sixteen
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[j], af[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, sca[i]);
      if (t == GRP - 1) { if constexpr (AMX) { scc[0] = scn[0]; scc[1] = scn[1]; } }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the two surplus K-tiles land before the wave ends)

  // ---- epilogue: lane = row m, register 4 q + e of tile (i, j) = column n0 + wn 64 + 32 j + 8 q + 4 hi + e; 16-byte pieces, read-modify-write for RESID ----
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)p.M * p.ldc * 4), 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    const float sl = AMX ? 1.f : p.sa[min(m, p.M - 1)];
    const unsigned voff = m < p.M ? (unsigned)(((size_t)m * p.ldc + 4 * hi) * 4) : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4_t old[4];
      if constexpr (RESID) {
#pragma unroll
        for (int q = 0; q < 4; ++q) old[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, voff, __builtin_amdgcn_readfirstlane((unsigned)((n0 + wn * 64 + 32 * j + 8 * q) * 4)), 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + wn * 64 + 32 * j + 8 * q + 4 * hi;
        const f32x4_t s4 = *(const f32x4_t*)(p.sw + c);
        f32x4_t b4 = {0.f, 0.f, 0.f, 0.f}, l4 = {1.f, 1.f, 1.f, 1.f};
        if (p.bias) b4 = *(const f32x4_t*)(p.bias + c);
        if (p.ls) l4 = *(const f32x4_t*)(p.ls + c);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = fmaf(acc[i][j][4 * q + e], sl * s4[e], b4[e]) * l4[e];
          if constexpr (RESID) v += __uint_as_float(old[q][e]);
          o[e] = __float_as_uint(v);
        }
        wvn_store_b128_guarded(o, rs_c, voff, __builtin_amdgcn_readfirstlane((unsigned)((n0 + wn * 64 + 32 * j + 8 * q) * 4)));
      }
    }
  }
}

template <bool RESID, bool AMX>
int launch(const DmaParams& p, hipStream_t st) {
  const int tiles = ceil_div(p.M, BM) * (p.N / BN);
  hipLaunchKernelGGL((gemm_fp8_dma_kernel<RESID, AMX>), dim3(tiles), dim3(256), LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

}  // namespace

int wvn_gemm_fp8_dma_launch(const GemmFp8Params& g, int epi, hipStream_t st) {
  if (getenv("WVN_NO_FP8_DMA")) return WVN_ERR_ARG;
  if (epi != EPI_F32 && epi != EPI_RESID_F32) return WVN_ERR_ARG;
  if (!g.A || !g.W || !g.sw || !g.C || g.M <= 0 || (g.K % (BK * GRP)) != 0 || (g.N % BN) != 0 || (g.lda % 16) != 0 || (g.ldw % 16) != 0 || (g.ldc % 4) != 0) return WVN_ERR_ARG;
  if ((((uintptr_t)g.A | (uintptr_t)g.W | (uintptr_t)g.C | (uintptr_t)g.sw) & 15) || (g.bias && ((uintptr_t)g.bias & 15)) || (g.ls && ((uintptr_t)g.ls & 15))) return WVN_ERR_ARG;
  if ((size_t)g.M * g.lda >= (1ull << 32) || (size_t)g.N * g.ldw >= (1ull << 32) || (size_t)g.M * g.ldc * 4 >= (1ull << 31)) return WVN_ERR_ARG;
  if (!g.a_scales && !g.sa) return WVN_ERR_ARG;
  if (g.a_scales && (((uintptr_t)g.a_scales & 15) || ((g.K / 32) % 16) != 0)) return WVN_ERR_ARG;   // (a row's scale bytes are read 16 at a time)
  if (epi == EPI_F32 && g.ls) return WVN_ERR_ARG;
  DmaParams p{};
  p.A = g.A; p.lda = g.lda; p.W = g.W; p.ldw = g.ldw; p.sa = g.sa; p.a_scales = g.a_scales; p.sw = g.sw; p.bias = g.bias; p.ls = g.ls;
  p.C = (float*)g.C; p.ldc = g.ldc; p.M = g.M; p.N = g.N; p.K = g.K;
  if (epi == EPI_RESID_F32) return g.a_scales ? launch<true, true>(p, st) : launch<true, false>(p, st);
  return g.a_scales ? launch<false, true>(p, st) : launch<false, false>(p, st);
}
