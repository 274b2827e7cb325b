// Micro-benchmark: what does it cost ONE wave per SIMD (the structure of the persistent GEMM kernels here) to ISSUE a 1 KB global -> LDS
// transfer, and how many MFMAs fit in its shadow?  Three ways to bring 1 KB per wave into LDS:
//   0  buffer_load ... lds (16 bytes per lane, LDS-DMA: what csrc/gemm_*_x3.hip, mlp_fused.hip, qkv_fused.hip use)
//   1  buffer_load_dwordx4 into registers (the ds_write_b128 that would follow is timed separately as form 2)
//   2  buffer_load_dwordx4 into registers + ds_write_b128 of the PREVIOUS iteration's registers
// Each form is timed alone (cycles per piece) and interleaved with MFMAs (one piece per 6 x v_mfma_f32_32x32x16_bf16 = 192 cycles of
// matrix pipe: the k-step of gemm_a384_x3): the excess over 192 is what the piece costs the wave.
//   hipcc --offload-arch=gfx950 -O3 -o dma_issue dma_issue.hip && ./dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int FORM, bool WITH_MFMA>
__global__ __launch_bounds__(256, 1) void k(const unsigned char* src, float* out, long long* cyc, int iters, unsigned bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  bf16x8_t a[2], b;
  f32x16_t acc[2];
  for (int c = 0; c < 2; ++c) {
    a[c] = *(const bf16x8_t*)(src + (c * 64 + lane) * 16);
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  }
  b = *(const bf16x8_t*)(src + (128 + lane) * 16);
  u32x4_t cur[8], prev[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { cur[u] = u32x4_t{0, 0, 0, 0}; prev[u] = cur[u]; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((i * 8 + u) * 4 + wave) * 1024u) % (bytes - 1024u));
      unsigned char* dst = lds + ((u * 4 + wave) * 1024);
      if constexpr (FORM == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, so, 0, 0);
      } else if constexpr (FORM == 1) {
        cur[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so, 0);
      } else if constexpr (FORM == 2) {
        *(u32x4_t*)(dst + lane * 16) = prev[u];   // the piece requested one round earlier (its data has arrived: one wait per round below)
        cur[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so, 0);
      }
      if constexpr (WITH_MFMA) {
#pragma unroll
        for (int m = 0; m < 6; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 1], b, acc[m & 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(cur[u])); prev[u] = cur[u]; }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int u = 0; u < 8; ++u) s += prev[u][0] * 1e-30f;
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  s += lds[threadIdx.x];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int FORM, bool WITH_MFMA>
void run(const char* name, const unsigned char* src, float* out, long long* cyc, unsigned bytes) {
  const int iters = 2000, blocks = 256;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<FORM, WITH_MFMA>), dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters, bytes);
    (void)hipDeviceSynchronize();
  }
  std::vector<long long> h(blocks * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (auto v : h) m += (double)v;
  m /= h.size();
  // s_memtime counts at 100 MHz on this chip: convert with the measured ratio of the MFMA-only loop (192 cycles per 6 MFMAs)
  printf("%-58s %8.1f memtime ticks per piece%s\n", name, m / (iters * 8.0), WITH_MFMA ? " (with 6 MFMAs)" : "");
}

int main() {
  const unsigned bytes = 1u << 20;   // (L2-resident: the question is the ISSUE cost, not the memory system)
  unsigned char* src; float* out; long long* cyc;
  (void)hipMalloc(&src, bytes); (void)hipMemset(src, 0, bytes);
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
  run<1, true>("(warm-up)", src, out, cyc, bytes);
  run<3, true>("6 MFMAs, no transfer", src, out, cyc, bytes);
  run<0, false>("buffer_load ... lds, alone", src, out, cyc, bytes);
  run<1, false>("buffer_load_dwordx4 -> VGPR, alone", src, out, cyc, bytes);
  run<2, false>("buffer_load_dwordx4 -> VGPR + ds_write_b128, alone", src, out, cyc, bytes);
  run<0, true>("buffer_load ... lds", src, out, cyc, bytes);
  run<1, true>("buffer_load_dwordx4 -> VGPR", src, out, cyc, bytes);
  run<2, true>("buffer_load_dwordx4 -> VGPR + ds_write_b128", src, out, cyc, bytes);
  return 0;
}
