// Attention output projection with residual for D = 384 on gfx950:   x[M,384] (fp32, in place) += (A[M,384] W^T + b) (* ls)
//
// 59 GFLOP against 775 MB of unavoidable traffic (155 MB of bf16 attention rows in, 620 MB of residual stream read-modify-written):
// an HBM-bound kernel that the K = 384 GEMM kernels run at 3.6 TB/s (216 us) because they stream W (288 KB per 256 rows, nine
// times the rows' own bytes through LDS DMA, rings and barriers) and expose the read-modify-write latency in an epilogue; a plain
// `x += 1` over the same 620 MB runs at 6.5 TB/s.  So this kernel is built for memory-level parallelism instead:
//   * W does not stream: a workgroup owns one HALF of the output columns and keeps that half of W (192 x 384 bf16 = 144 KB,
//     XOR-swizzled) resident in LDS for its whole life -- loaded once by DMA, no ring, no barrier in the main loop;
//   * 8 waves per CU (two per SIMD), each taking 32-row groups on its own: the rows as 24 MFMA operand fragments straight from
//     global memory, three column tiles at a time (48 accumulator registers), W fragments from LDS, and the residual
//     read-modify-written straight from the accumulator layout (lane = column, register = row: every load / store instruction
//     covers whole 128-byte lines; the loads are issued before the 72 MFMAs that produce their addends, so their latency hides
//     behind the matrix work and behind the other seven waves);
//   * the two workgroups that own the two column halves of the same rows run on the same XCD at the same time (blockIdx b and
//     b + 8), so the second read of the attention rows is an L2 hit, not HBM traffic.
#include "operand.h"
#include "wvn_internal.h"

namespace {

constexpr int KD = 384;                 // K and N
constexpr int NH = 192;                 // output columns per workgroup (half)
constexpr int WROW = KD * 2;            // bytes per W row in LDS (768 = 48 chunks of 16 B)
constexpr int W_BYTES = NH * WROW;      // 147,456
constexpr int TAB_OFF = W_BYTES;        // bias [192], LayerScale [192] (fp32)

struct ProjParams {
  const op16_t* A; int lda;
  const op16_t* W;          // [384][384]
  const float* bias;        // [384] or nullptr
  const float* ls;          // [384] or nullptr
  float* X; int ldx;
  int M;
};

__device__ inline void store_b128_guarded(u32x4_t v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0);   // (mlp_fused.hip: the gfx950 store-data hazard LLVM does not pad)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(512, 1) void proj_resid_kernel(ProjParams p) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  // workgroup b: column half (b >> 3) & 1 of pair (b >> 4) * 8 + (b & 7); b and b + 8 (same XCD) share their rows
  const int half = ((int)blockIdx.x >> 3) & 1;
  const int pair = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7), npairs = (int)gridDim.x >> 1;
  const int n_base = half * NH;

  // ---- this half of W -> LDS, once: 144 DMA pieces of 1 KB (64 lanes x 16 B), 18 per wave.  Row n (768 B) chunk c lands at chunk
  // (c & ~15) | ((c ^ n) & 15): fragment reads of 16 rows x one chunk column then hit 16 different bank groups ----
  {
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)(KD * KD * 2), 0x00020000);
    for (int i = 0; i < 18; ++i) {
      const int piece = wave * 18 + i;                 // 1 KB = LDS bytes [piece * 1024, +1024)
      const int lin = piece * 64 + lane;               // 16-byte LDS chunk index
      const int n = lin / 48, c_lds = lin - n * 48;
      const int c_src = (c_lds & ~15) | ((c_lds ^ n) & 15);
      const unsigned voff = (unsigned)(((n_base + n) * KD + c_src * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, voff, 0, 0, 0);
    }
  }
  float* bias_l = (float*)(smem + TAB_OFF);
  for (int i = tid; i < NH; i += 512) {
    bias_l[i] = p.bias ? p.bias[n_base + i] : 0.f;
    bias_l[NH + i] = p.ls ? p.ls[n_base + i] : 1.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)((size_t)p.M * p.ldx * 4), 0x00020000);
  // W fragment of column tile T (32 columns), k-step s: row n = 32 T + l31, chunk 2 s + hi
  const unsigned wrow = l31 * WROW;
  const int wx = l31 & 15;
  auto wfrag = [&](int T, int s) -> opx8_t {
    const int c = 2 * s + hi;
    return *(const opx8_t*)(smem + T * 32 * WROW + wrow + ((((c & ~15) | ((c ^ wx) & 15))) << 4));
  };
  const int ngroups = (p.M + 31) / 32;
  for (int rg = pair * 8 + wave; rg < ngroups; rg += npairs * 8) {
    const int m0 = rg * 32;
    const int m = min(m0 + l31, p.M - 1);                      // rows past M: clamped reads, dropped stores
    // rows -> MFMA operand fragments (row l31, k = 16 s + 8 hi .. + 7)
    opx8_t af[KD / 16];
    {
      const op16_t* ap = p.A + (size_t)m * p.lda + hi * 8;
#pragma unroll
      for (int s = 0; s < KD / 16; ++s) af[s] = *(const opx8_t*)(ap + s * 16);
    }
    // Residual addressing in the STRAIGHT accumulator layout (A operand = rows, B operand = W): lane = column n_base + 32 T + l31,
    // register 4 g + e = row m0 + 8 g + 4 hi + e.  One dword per lane, but an instruction covers two whole 128-byte lines (32
    // consecutive columns of two rows); the transposed layout's 16-byte pieces touch 32 lines per instruction for the same 1 KB,
    // and the L1's line rate, not HBM, then bounds the kernel (200 us; 216 for the staged epilogues of the older kernels).
    const unsigned xlane = (unsigned)(((m0 + 4 * hi) * p.ldx + n_base + l31) * 4);   // + (8 g + e) * ldx * 4 (scalar) + 128 T (immediate)
    unsigned srow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) srow[r] = __builtin_amdgcn_readfirstlane((unsigned)((8 * (r >> 2) + (r & 3)) * p.ldx * 4));
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __builtin_amdgcn_sched_barrier(0);   // (keeps the second pass's 48 loads out of the first pass: they would not fit the registers)
      // residual addends of this pass's three column tiles: requested now, needed after the MFMAs
      uint32_t xr[3][16];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[t][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, xlane + 128u * (3 * pass + t), srow[r], 0);
      f32x16_t acc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float bn = bias_l[32 * (3 * pass + t) + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bn;
      }
#pragma unroll
      for (int s = 0; s < KD / 16; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc[t] = wvn_mfma_32x32x16(af[s], wfrag(3 * pass + t, s), acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float ln = bias_l[NH + 32 * (3 * pass + t) + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float o = acc[t][r] * ln + __uint_as_float(xr[t][r]);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rs_x, xlane + 128u * (3 * pass + t), srow[r], 0);   // rows >= M: dropped
        }
      }
    }
  }
}

int proj_num_cus() {
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

// Eligibility: N == K == 384, 16-byte aligned operands, 32-bit byte offsets, a CU count that is a multiple of 16 (the pairing of
// the two column halves on one XCD).  WVN_ERR_ARG otherwise.
int WVN_OPSYM(wvn_proj_resid_launch)(const op16_t* A, int lda, const op16_t* W, const float* bias, const float* ls, float* x, int ldx, int M,
                          hipStream_t st) {
  if (!A || !W || !x || M <= 0 || (lda % 8) != 0 || (ldx % 4) != 0) return WVN_ERR_ARG;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)x) & 15) != 0) return WVN_ERR_ARG;
  if ((size_t)M * ldx * 4 >= (1ull << 32)) return WVN_ERR_ARG;
  const int ncu = proj_num_cus();
  if (ncu % 16) return WVN_ERR_ARG;
  const int lds = TAB_OFF + 2 * NH * 4;
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(160 * 1024, (const void*)proj_resid_kernel)) return rc;
  ProjParams p{};
  p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.ls = ls; p.X = x; p.ldx = ldx; p.M = M;
  hipLaunchKernelGGL(proj_resid_kernel, dim3(ncu), dim3(512), lds, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
