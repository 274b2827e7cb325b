// Probe (round 6): the pieces the MX correction terms of the <= 1e-3 mode rest on, checked on the hardware before any kernel is built on them.
//   1. v_cvt_scalef32_pk_bf8_f32 / _f16 and v_cvt_pk_bf8_f32: rounding, saturation, what the scale operand does (divide or multiply);
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 with bf8 (e5m2) operands: which k a (lane, byte) pair is, that the A and B operands pair byte for
//      byte, what a scale byte means and that it is taken per LANE (row, 32-k block);
//   3. the rate of the c8 instruction mix with bf8 formats and non-unit scales (must equal profiles/r05_mx_terms_ubench.md's e4m3 rows).
//   hipcc --offload-arch=gfx950 -O3 -o mx_formats mx_formats.hip && ./mx_formats
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) short s2;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void cvt_probe(const float* x, unsigned* o, float sc, int n, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // MODE.FP16_OVFL (csrc/common.h: wvn_fp16_saturate)
  const int i = threadIdx.x;
  if (i >= n) return;
  s2 old = {0, 0};
  const auto r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(old, x[2 * i], x[2 * i + 1], sc, false);
  o[i] = (unsigned)(unsigned short)r[0];
  const h2 hh = {(_Float16)x[2 * i], (_Float16)x[2 * i + 1]};
  const auto r2 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(old, hh, sc, false);
  o[64 + i] = (unsigned)(unsigned short)r2[0];
  o[128 + i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(x[2 * i], x[2 * i + 1], 0, false) & 0xffffu;
  const auto r4 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[2 * i], x[2 * i + 1], sc, false);
  o[192 + i] = (unsigned)(unsigned short)r4[0];
}

// D[i][j] = sum_k A[i][k] B[j][k]; a / b: [32][64] bytes; lane (r = l & 31, h = l >> 5) supplies bytes a[r][32 h .. 32 h + 31] in register order
__global__ void mfma_probe(const unsigned char* a, const unsigned char* b, const int* sca, const int* scb, float* d, int fmt) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8_t av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = *(const int*)(a + r * 64 + 32 * h + 4 * i);
    bv[i] = *(const int*)(b + r * 64 + 32 * h + 4 * i);
  }
  f32x16_t acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (fmt == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 1, 1, 0, sca[l], 0, scb[l]);
  else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sca[l], 0, scb[l]);
  for (int i = 0; i < 16; ++i) d[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];   // row i of A, column r of B
}

template <int MODE>
__global__ void rate(const unsigned* src, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16_t acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  i32x8_t a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (int)src[lane * 8 + i]; b8[i] = (int)src[512 + lane * 8 + i]; }
  const f16x8_t ah = __builtin_bit_cast(f16x8_t, *(const __attribute__((ext_vector_type(4))) int*)&a8), bh = __builtin_bit_cast(f16x8_t, *(const __attribute__((ext_vector_type(4))) int*)&b8);
  const int s115 = 0x73737373, s127 = 0x7f7f7f7f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
      if constexpr (MODE == 0) {   // e4m3, unit scales (round 5's row)
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, s127, 0, s127);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, s127, 0, s127);
      } else {                     // e5m2, one residue operand scaled by 2^-12 each
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 1, 1, 0, s127, 0, s115);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 1, 1, 0, s115, 0, s127);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static float bf8_to_f(unsigned char v) {   // e5m2 = the top byte of an fp16
  const unsigned short h = (unsigned short)v << 8;
  _Float16 f;
  memcpy(&f, &h, 2);
  return (float)f;
}
static float fp8_to_f(unsigned char v) {   // OCP e4m3
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}

int main() {
  // ---- 1. conversions ----
  const float xs[] = {1.0f, 1.1f, 1.125f, 1.2f, 1.375f, 1.3f, -3.7f, 0.001f, 65000.f, 1e6f, 1e-6f, 3e-5f, 4096.f, 1.f / 4096, 0.3f / 4096, -0.77f / 4096,
                      57344.f, 61440.f, 0.f, -0.f, 2.5f, 3.5f, 1.625f, 1.875f};
  const int n = sizeof(xs) / 4 / 2;
  float* dx; unsigned* dob;
  (void)hipMalloc(&dx, sizeof(xs)); (void)hipMalloc(&dob, 256 * 4);
  (void)hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
  for (int ovfl : {1}) {   // what the library's kernels see: MODE.FP16_OVFL set
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dob, 1.0f, n, ovfl);
    unsigned ho[256];
    (void)hipMemcpy(ho, dob, sizeof(ho), hipMemcpyDeviceToHost);
    printf("with MODE.FP16_OVFL = 1, scale operand 1:\n");
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 2; ++e)
        if (fabsf(xs[2 * i + e]) >= 4096.f || xs[2 * i + e] == 1.0f) {
          const unsigned char b0 = (ho[i] >> (8 * e)) & 255, b1 = (ho[64 + i] >> (8 * e)) & 255, b2 = (ho[128 + i] >> (8 * e)) & 255;
          printf("  x %13.6g : scalef32_pk_bf8_f32 %02x = %-12g  scalef32_pk_bf8_f16 %02x = %-12g  pk_bf8_f32 %02x = %g\n", xs[2 * i + e], b0, bf8_to_f(b0), b1, bf8_to_f(b1), b2, bf8_to_f(b2));
        }
  }
  for (float sc : {1.0f, 1.0f / 4096, 4096.f}) {
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dob, sc, n, 0);
    unsigned ho[256];
    (void)hipMemcpy(ho, dob, sizeof(ho), hipMemcpyDeviceToHost);
    printf("scale operand %g:\n", sc);
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 2; ++e) {
        const unsigned char b0 = (ho[i] >> (8 * e)) & 255, b1 = (ho[64 + i] >> (8 * e)) & 255, b2 = (ho[128 + i] >> (8 * e)) & 255, b3 = (ho[192 + i] >> (8 * e)) & 255;
        printf("  x %13.6g : scalef32_pk_bf8_f32 %02x = %-12g  scalef32_pk_bf8_f16 %02x = %-12g  pk_bf8_f32 (no scale) %02x = %-12g  scalef32_pk_fp8_f32 %02x = %g\n", xs[2 * i + e], b0,
               bf8_to_f(b0), b1, bf8_to_f(b1), b2, bf8_to_f(b2), b3, fp8_to_f(b3));
      }
  }
  // ---- 2. the scaled MFMA ----
  for (int fmt : {1, 0}) {
    std::vector<unsigned char> a(32 * 64), b(32 * 64);
    std::vector<float> af(32 * 64), bf(32 * 64);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int i = 0; i < 32 * 64; ++i) {
      // small-integer-ish values, exactly representable: sign, exponent in a narrow band, two mantissa bits
      const unsigned char va = fmt == 1 ? (unsigned char)(((rnd() & 1) << 7) | ((13 + rnd() % 5) << 2) | (rnd() & 3)) : (unsigned char)(((rnd() & 1) << 7) | ((5 + rnd() % 5) << 3) | (rnd() & 7));
      const unsigned char vb = fmt == 1 ? (unsigned char)(((rnd() & 1) << 7) | ((13 + rnd() % 5) << 2) | (rnd() & 3)) : (unsigned char)(((rnd() & 1) << 7) | ((5 + rnd() % 5) << 3) | (rnd() & 7));
      a[i] = va; b[i] = vb;
      af[i] = fmt == 1 ? bf8_to_f(va) : fp8_to_f(va);
      bf[i] = fmt == 1 ? bf8_to_f(vb) : fp8_to_f(vb);
    }
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) {   // per-lane scale bytes: row (l & 31) and k block (l >> 5) dependent; junk in the upper bytes (opsel 0 must ignore them)
      sa[l] = (127 - (l & 3) - 2 * (l >> 5)) | 0x11223300;
      sb[l] = (127 - 12 + ((l >> 2) & 1) + (l >> 5)) | 0x44556600;
    }
    unsigned char *da, *db; int *dsa, *dsb; float* dd;
    (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dd, 4096);
    (void)hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, fmt);
    std::vector<float> d(1024);
    (void)hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
    // three readings of (which k block a byte belongs to, which lane supplies its scale):
    //   H1: a lane's 32 bytes are one block, scaled by the lane's own byte;  H2: registers 0-3 of BOTH half-waves are block 0, registers 4-7 block 1,
    //   and the scale of (row r, block kb) comes from lane r + 32 kb;  U: uniform scales (the pairing of A and B bytes alone)
    for (int hyp = 0; hyp < 3; ++hyp) {
      if (hyp == 2) {
        for (int l = 0; l < 64; ++l) { sa[l] = 0x7f7f7f7f - 0x03030303; sb[l] = 0x73737373; }
        (void)hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, fmt);
        (void)hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
      }
      double worst = 0, big = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int h = 0; h < 2; ++h)
            for (int by = 0; by < 32; ++by) {
              const int kb = hyp == 1 ? by / 16 : h;   // the block of byte `by` of half-wave h
              ref += (double)af[i * 64 + 32 * h + by] * bf[j * 64 + 32 * h + by] * ldexp(1.0, ((sa[i + 32 * kb] & 255) - 127) + ((sb[j + 32 * kb] & 255) - 127));
            }
          worst = fmax(worst, fabs(ref - d[i * 32 + j]));
          big = fmax(big, fabs(ref));
        }
      printf("scaled MFMA, format %s, %s: max |D - ref| = %.3g (max |ref| %.3g)  -> %s\n", fmt == 1 ? "bf8 e5m2" : "fp8 e4m3",
             hyp == 0 ? "H1 (a lane's bytes = one block, its own scale byte)" : hyp == 1 ? "H2 (registers 0-3 = block 0 from lane r, 4-7 = block 1 from lane r + 32)" : "uniform scales 2^-3 x 2^-12",
             worst, big, worst <= 1e-5 * big ? "CONFIRMED" : "mismatch");
    }
  }
  // ---- 3. rate ----
  unsigned* src; float* out; long long* cyc;
  (void)hipMalloc(&src, 1 << 16); (void)hipMemset(src, 0x3c, 1 << 16);
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
  for (int rep = 0; rep < 2; ++rep)
    for (int w : {1, 2}) {
      for (int mode = 0; mode < 2; ++mode) {
        const int iters = 4000, blocks = 256, threads = 256 * w;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL((rate<0>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
        else hipLaunchKernelGGL((rate<1>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * threads / 64);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0;
        for (auto v : h) m += (double)v;
        m /= h.size();
        const double tiles = (double)blocks * (threads / 64) * iters * 4;
        if (rep) printf("c8 mix, %s, %d wave(s)/SIMD: %.1f ticks per 64-k tile step and wave, %.3f ms, %.0f algorithmic TFLOP/s\n", mode ? "bf8 with 2^-12 scales" : "e4m3 unit scales", w,
                        m / (iters * 4.0), ms, tiles * 32.0 * 32 * 64 * 2 / (ms * 1e-3) / 1e12);
      }
    }
  return 0;
}
