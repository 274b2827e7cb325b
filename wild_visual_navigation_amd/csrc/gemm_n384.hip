// Row-panel bf16 MFMA GEMM for the N = 384 residual update with a long K (fc2 of the ViT MLP, K = 1536):
//   C[M,384] (fp32, in place) += A[M,K] * W[384,K]^T + bias
//
// A workgroup (8 waves) owns 256 rows and ALL 384 output columns: every wave keeps the 32 x 384 fp32 accumulator of its
// 32 rows in registers (12 MFMA tiles = 192 VGPRs), so A (the 4x wider hidden activation, by far the larger operand) is
// read exactly once and the fp32 residual read-modify-write happens once per row, after the whole K loop.  K is streamed
// in slices of 32: per slice the W part [384 n][32 k] (24 KB, shared by the 8 waves) and the A part [256 m][32 k] (16 KB,
// each wave DMAs and reads only its own 32 rows) arrive by buffer_load ... lds DMA into a 3-deep LDS ring -- no staging
// VGPRs, no ds_write.  LDS rows are 64 B; 16-byte chunks are XOR-swizzled with (row >> 2) & 3 on the DMA source address and
// on the read, so every fragment read is one conflict-free ds_read_b128.  One s_barrier and a counted s_waitcnt vmcnt per
// slice (one slice stays in flight across the barrier); 24 MFMAs per wave per slice.
// The epilogue (once per 48 slices at K = 1536) moves the accumulators through a wave-private LDS image, 128 columns at a
// time, and updates C with whole 512-byte rows (16-byte buffer loads / stores).
//
// MFMA: v_mfma_f32_32x32x16_bf16, orientation mfma(Wfrag, Afrag): lane = output row, registers = 4 consecutive columns.
#include <type_traits>

#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int NN = 384;                 // output columns (12 MFMA tiles)
constexpr int NTILE = NN / 32;
constexpr int BKS = 32;                 // k per ring slice (2 MFMA k-steps)
constexpr int BM = 256;
constexpr int NS = 3;
constexpr int W_BYTES = NN * BKS * 2;   // 24 KB
constexpr int A_BYTES = BM * BKS * 2;   // 16 KB
constexpr int STAGE_BYTES = W_BYTES + A_BYTES;

constexpr int STG_PITCH = 132;                             // floats per staged row (128 columns + 4: conflict-light)
constexpr int STG_BYTES = 32 * STG_PITCH * 4;              // per wave: 16,896
constexpr int BIAS_OFF = 8 * STG_BYTES;                    // 135,168 (> RING_BYTES: staging and ring share the front)
constexpr int LDS_BYTES = BIAS_OFF + NN * 4;

struct N384Params {
  const op16_t* A; int lda;
  const op16_t* W; int ldw;   // [384][K]
  const float* bias;
  float* C; int ldc;          // fp32, updated in place
  int M, K;
};

__global__ __launch_bounds__(512, 2) void gemm_n384_kernel(N384Params p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nk = p.K / BKS;
  const float* bias_l = (const float*)(smem + BIAS_OFF);
  for (int i = tid; i < NN; i += 512) ((float*)(smem + BIAS_OFF))[i] = p.bias ? p.bias[i] : 0.f;

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)NN * p.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((size_t)p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)p.M * p.ldc * 4), 0x00020000);
  // W: 24 wave-instructions of 1 KB (16 rows x 64 B) per slice, 3 per wave; LDS chunk lane & 3 of row r holds source chunk
  // (lane & 3) ^ ((r >> 2) & 3)
  unsigned wvoff[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int row = (wave * 3 + u) * 16 + (lane >> 2);
    wvoff[u] = (unsigned)((row * p.ldw + (((lane & 3) ^ ((row >> 2) & 3)) * 8)) * 2);
  }
  const int xorw = (l31 >> 2) & 3;
  const unsigned rdw = l31 * 64;                         // + t * 2048 per column tile
  const unsigned rda = W_BYTES + wave * 2048 + l31 * 64;  // this wave's 32 A rows

  const int nrb = (p.M + BM - 1) / BM;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m0w = rb * BM + wave * 32;
    // A: this wave's 32 rows, 2 wave-instructions (16 rows x 64 B); rows past M are clamped (their results are dropped)
    unsigned avoff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rl = u * 16 + (lane >> 2);
      const int row = min(m0w + rl, p.M - 1);
      avoff[u] = (unsigned)(((size_t)row * p.lda + (((lane & 3) ^ ((rl >> 2) & 3)) * 8)) * 2);
    }
    auto issue = [&](int i) {
      unsigned char* st = smem + (i % NS) * STAGE_BYTES;
      const unsigned soff = __builtin_amdgcn_readfirstlane(i * BKS * 2);
#pragma unroll
      for (int u = 0; u < 3; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(st + (wave * 3 + u) * 1024), 16,
                                                 wvoff[u], soff, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(st + W_BYTES + wave * 2048 + u * 1024),
                                                 16, avoff[u], soff, 0, 0);
    };
    __syncthreads();  // previous row block's staging reads are done (and the bias table is visible) before DMA reuses the LDS
    issue(0);
    if (nk > 1) issue(1);

    // accumulators start at zero and the bias is added after the K loop, then the residual: (acc + bias) + C is the
    // association of the tiled kernel too, so a row's result does not depend on which of the two kernels computed it
    f32x16_t acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int i = 0; i < nk; ++i) {
      if (i + 1 < nk) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // slice i landed; slice i + 1 (5 DMAs) may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 2 < nk) issue(i + 2);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* st = smem + (i % NS) * STAGE_BYTES;
      int xc = xorw;
      asm volatile("" : "+v"(xc));
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const opx8_t af = *(const opx8_t*)(st + rda + (((2 * s + hi) ^ xc) << 4));
        auto rd = [&](int t) { return *(const opx8_t*)(st + rdw + t * 2048 + (((2 * s + hi) ^ xc) << 4)); };
        opx8_t wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = rd(t);
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
          acc[t] = wvn_mfma_32x32x16(wf[t & 3], af, acc[t], 0, 0, 0);
          if (t + 4 < NTILE) wf[t & 3] = rd(t + 4);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
        for (int t = 0; t < NTILE - 4; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: C[rows of this wave][384] += acc, 128 columns at a time through the wave's LDS image -----------------
    __syncthreads();  // every wave is done reading the ring: the staging images overlap it
    float* stg = (float*)(smem + wave * STG_BYTES);
    const unsigned cvoff = (unsigned)(((lane >> 5) * p.ldc + (lane & 31) * 4) * 4);
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
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const unsigned so = __builtin_amdgcn_readfirstlane(((m0w + 2 * it) * p.ldc + 128 * c) * 4);
        f32x4_t v = *(const f32x4_t*)(stg + (2 * it + (lane >> 5)) * STG_PITCH + (lane & 31) * 4);
        const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs_c, cvoff, so, 0);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __float_as_uint((v[e] + b4[e]) + __uint_as_float(r[e]));
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_c, cvoff, so, 0);  // rows >= M fall outside num_records: dropped
        if ((it & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

int n384_num_cus() {
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

// Eligibility: N == 384, K % 32 == 0, residual epilogue, 16-byte aligned operands, 32-bit byte offsets;
// WVN_ERR_ARG otherwise (the caller then uses the generic tiled kernel).  *rows_done = number of leading rows handled
// here (a multiple of 256 unless it is M); the caller finishes rows [*rows_done, M).
// force != 0 (tests): take every row block here, whatever their number
int WVN_OPSYM(wvn_gemm_n384_launch)(const GemmBf16Params& g, int epi, hipStream_t st, int* rows_done, int force) {
  if (epi != EPI_RESID_F32 && epi != EPI_ACCUM_F32) return WVN_ERR_ARG;
  if (g.N != NN || g.K <= 0 || (g.K % BKS) != 0 || g.M <= 0 || !g.A || !g.W || !g.C) return WVN_ERR_ARG;
  if ((g.lda % 8) || (g.ldw % 8) || (g.ldc % 4)) return WVN_ERR_ARG;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15) || ((uintptr_t)g.C & 15)) return WVN_ERR_ARG;
  if ((size_t)g.M * g.lda * 2 >= (1ull << 32) || (size_t)g.M * g.ldc * 4 >= (1ull << 32) || (size_t)NN * g.ldw * 2 >= (1ull << 32))
    return WVN_ERR_ARG;
  // Row blocks are dealt round-robin to one persistent workgroup per CU; a last round that would occupy less than a
  // quarter of the CUs is left to the caller's tiled kernel instead (rows are independent): 786 row blocks on 256 CUs
  // are 3 full rounds here + 18 row blocks there, not 4 rounds.
  const int ncu = n384_num_cus();
  int nrb_all = ceil_div(g.M, BM), m_here = g.M;
  // One workgroup owns 256 rows x all 384 columns, so M / 256 workgroups is all the parallelism there is: below about three
  // quarters of a round the tiled kernel (3 column tiles per row block) is faster (measured: 1-8 frames of 3152 rows 0.33 /
  // 0.49 / 0.74 ms there vs 0.80 / 0.81 / 0.91 ms here per 12 launches; 16 frames 1.32 vs 1.26).  Same association, same bits.
  if (!force && nrb_all * 4 < ncu * 3) return WVN_ERR_ARG;
  const int full = (nrb_all / ncu) * ncu, rem = nrb_all - full;
  if (rows_done && full > 0 && rem > 0 && rem * 4 <= ncu) m_here = full * BM;
  if (rows_done) *rows_done = m_here;
  N384Params p{};
  p.A = g.A; p.lda = g.lda; p.W = g.W; p.ldw = g.ldw; p.bias = g.bias; p.C = (float*)g.C; p.ldc = g.ldc; p.M = m_here; p.K = g.K;
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(LDS_BYTES, (const void*)gemm_n384_kernel)) return rc;
  const int nrb = ceil_div(m_here, BM);
  const int grid = nrb < ncu ? nrb : ncu;
  hipLaunchKernelGGL(gemm_n384_kernel, dim3(grid), dim3(512), LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
