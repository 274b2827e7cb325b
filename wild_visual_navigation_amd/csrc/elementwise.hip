// HBM-bound elementwise / normalisation kernels of the backbone (gfx950).
//   patchify  : ImageNet-normalise + im2col of PxP patches (fused; one pass over the frame)
//   cls rows  : token 0 of every frame = cls_token + pos[0]
//   layernorm : one 64-lane wave per token row, two-pass statistics in registers, fp32 math
//   upsample  : bilinear align_corners=True  token-major [B,G*G,D] -> NCHW [B,D,H,H]
#include <type_traits>

#include "common.h"
#include "wvn_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// patchify: img [B,3,S,S] fp32 in [0,1] -> patches [B*G*G, 3*P*P], k = c*P*P + py*P + px
// (the flattening order of the conv weight [D,3,P,P]).  dino_interface.py:52 normalisation fused.
// One thread per (patch, c, py): reads P contiguous floats, writes P contiguous outputs.
// ---------------------------------------------------------------------------------------------
// ldp: row stride of the patch matrix in elements (>= 3*P*P; the pad columns, if any, are zeroed by the caller).
// out_lo != nullptr (T = bf16_t only): exact mode, out receives the hi plane and out_lo the lo plane of the same value.
// Frame ingest: with gather tables (rows / cols != nullptr) the network input pixel (y, x) is pixel
// (rows[y], cols[x]) of the [B,3,Hs,Ws] SOURCE frame -- T.Resize(NEAREST) + T.CenterCrop (dino_interface.py:52-59,
// image_projector.py:56-59) never materialise; TIN = unsigned char: raw 8-bit pixels, x / 255 fused as well.
struct Gather { const int* rows; const int* cols; int Hs, Ws; };
template <typename TIN>
__device__ inline float load_pixel(const TIN* p) {
  if constexpr (sizeof(TIN) == 1) return (float)*p / 255.0f;  // == torch's x.float() / 255
  else return *p;
}
template <typename T, int P, typename TIN = float>
__global__ void patchify_kernel(const TIN* __restrict__ img, T* __restrict__ out, T* __restrict__ out_lo, int ldp, int B, int S,
                                Gather gt = Gather{nullptr, nullptr, 0, 0}) {
  wvn_fp16_saturate();
  const int G = S / P;
  const long long total = (long long)B * G * G * 3 * P;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  // order threads so that consecutive threads walk gx fastest -> contiguous image reads per row
  int gx = (int)(i % G);
  long long r = i / G;
  int py = (int)(r % P); r /= P;
  int gy = (int)(r % G); r /= G;
  int c = (int)(r % 3);
  int b = (int)(r / 3);
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const int Hs = gt.rows ? gt.Hs : S, Ws = gt.rows ? gt.Ws : S;
  const TIN* src = img + (((size_t)b * 3 + c) * Hs + (gt.rows ? gt.rows[gy * P + py] : gy * P + py)) * Ws;
  const size_t doff = ((size_t)b * G * G + gy * G + gx) * ldp + c * P * P + py * P;
  T* dst = out + doff;
#pragma unroll
  for (int px = 0; px < P; ++px) {
    const int x = gx * P + px;
    const float v = (load_pixel(src + (gt.cols ? gt.cols[x] : x)) - mean[c]) / stdv[c];
    ElemIO<T>::store(dst + px, v);
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (out_lo) out_lo[doff + px] = f32_to_bf16(v - bf16_to_f32(f32_to_bf16(v)));
    }
  }
}

// bf16, P = 8 fast path: one workgroup per (patch row gy, frame).  The 3 x 8 image rows the patch row needs are read
// as float4 (fully coalesced, 43 KB), normalised, converted, and scattered into an LDS image that already has the output
// order [gx][c][py][px]; the 21 KB image then leaves as contiguous 16-byte stores (the element-order kernel above writes
// 16-byte fragments 384 B apart).
template <typename TIN, bool F16 = false>  // float: pixels in [0,1]; unsigned char: raw 8-bit pixels, x / 255 is fused (frame ingest without the
                         // 4x larger fp32 upload: quick_start.py:160-161 / ros_converter.py:113-126 do the division on the host side)
                         // F16: fp16 instead of bf16 operands (WVN_PREC_F16)
__global__ __launch_bounds__(256) void patchify8_bf16_rows_kernel(const TIN* __restrict__ img, bf16_t* __restrict__ out, int S) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) bf16_t prow[];  // [G][192]
  const int G = S / 8, gy = blockIdx.x, b = blockIdx.y;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const int q = S / 4;  // 4-pixel groups per image row
  for (int i = threadIdx.x; i < 3 * 8 * q; i += 256) {
    const int c = i / (8 * q), r = i - c * 8 * q, py = r / q, x4 = r - py * q;
    const size_t src = (((size_t)b * 3 + c) * S + gy * 8 + py) * S + x4 * 4;
    f32x4_t v;
    if constexpr (sizeof(TIN) == 1) {
      const uchar4 u = *(const uchar4*)(img + src);
      v = f32x4_t{(float)u.x / 255.0f, (float)u.y / 255.0f, (float)u.z / 255.0f, (float)u.w / 255.0f};  // == torch's x.float() / 255
    } else {
      v = *(const f32x4_t*)(img + src);
    }
    const int gx = x4 >> 1, px = (x4 & 1) * 4;
    const float m = mean[c], sd = stdv[c];
    auto pk = [](float a, float b2) { return F16 ? pack_f16x2(a, b2) : pack_bf16x2(a, b2); };
    u32x2_t o = {pk((v[0] - m) / sd, (v[1] - m) / sd), pk((v[2] - m) / sd, (v[3] - m) / sd)};
    *(u32x2_t*)(prow + gx * 192 + c * 64 + py * 8 + px) = o;
  }
  __syncthreads();
  u32x4_t* dst = (u32x4_t*)(out + ((size_t)b * G * G + (size_t)gy * G) * 192);
  for (int i = threadIdx.x; i < G * 192 / 8; i += 256) dst[i] = ((const u32x4_t*)prow)[i];
}

// The same patch-row panel gathered from a larger source frame through the ingest tables (one element per thread and step:
// a NEAREST down-sampling touches isolated pixels, there is nothing to vectorise on the read side).
// PLANES (the split-operand modes): hi = bf16(v) -> out, lo = bf16(v - hi) -> out_lo, a second LDS image (the element-order kernel above wrote the
// two planes of the <= 1e-3 modes in 16-byte fragments 384 B apart: 628 us per 128-frame launch pair against ~200 here)
template <typename TIN, bool F16, bool PLANES = false>
__global__ __launch_bounds__(256) void patchify8_gather_rows_kernel(const TIN* __restrict__ img, bf16_t* __restrict__ out, int S,
                                                                    Gather gt, bf16_t* __restrict__ out_lo = nullptr) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) bf16_t prow[];  // [G][192] (PLANES: two of them)
  const int G = S / 8, gy = blockIdx.x, b = blockIdx.y;
  bf16_t* plo = prow + G * 192;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  // a thread owns network columns x = tid, tid + 256, ...: its column-table entry is loaded once, the eight row-table entries are wave-uniform, and the 24
  // pixel loads of a column (3 channels x 8 rows) are independent -- no integer division per element (the element-order loop this replaces spent most of its
  // 437 us per 64 frames on `i / (8 S)` and `r / S` with a runtime S: round 6)
  int ry[8];
#pragma unroll
  for (int py = 0; py < 8; ++py) ry[py] = gt.rows[gy * 8 + py];
  for (int x = threadIdx.x; x < S; x += 256) {
    const int cx = gt.cols[x];
    const int ob = (x >> 3) * 192 + (x & 7);
    float raw[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int py = 0; py < 8; ++py) raw[c][py] = load_pixel(img + (((size_t)b * 3 + c) * gt.Hs + ry[py]) * gt.Ws + cx);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int py = 0; py < 8; ++py) {
        const float v = (raw[c][py] - mean[c]) / stdv[c];
        const int o = ob + c * 64 + py * 8;
        const bf16_t h = F16 ? f32_to_f16(v) : f32_to_bf16(v);
        prow[o] = h;
        if constexpr (PLANES) plo[o] = f32_to_bf16(v - bf16_to_f32(h));
      }
  }
  __syncthreads();
  u32x4_t* dst = (u32x4_t*)(out + ((size_t)b * G * G + (size_t)gy * G) * 192);
  for (int i = threadIdx.x; i < G * 192 / 8; i += 256) dst[i] = ((const u32x4_t*)prow)[i];
  if constexpr (PLANES) {
    u32x4_t* dl = (u32x4_t*)(out_lo + ((size_t)b * G * G + (size_t)gy * G) * 192);
    for (int i = threadIdx.x; i < G * 192 / 8; i += 256) dl[i] = ((const u32x4_t*)plo)[i];
  }
}

// NEAREST resize + crop as an image (ImageProjector.resize_image, image_projector.py:199-200): out[b,c,y,x] = in[b,c,rows[y],cols[x]]
template <typename T>
__global__ void gather_image_kernel(const T* __restrict__ in, T* __restrict__ out, long long planes, int Ho, int Wo, Gather gt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * Ho * Wo) return;
  const int x = (int)(i % Wo);
  const long long r = i / Wo;
  const int y = (int)(r % Ho);
  const long long pl = r / Ho;
  out[i] = in[(pl * gt.Hs + gt.rows[y]) * gt.Ws + gt.cols[x]];
}

__global__ void cls_rows_kernel(const float* __restrict__ cls_pos, float* __restrict__ x, int B, int ntok, int D) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  int b = i / D, d = i - b * D;
  x[(size_t)b * ntok * D + d] = cls_pos[d];
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per output row.  D <= 64*VPT_MAX.  Statistics exactly as torch (biased
// variance, eps inside the sqrt), all in fp32.
// ---------------------------------------------------------------------------------------------
template <typename T, int VPT>  // VPT = D / 64 values per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int ldy,
                                                        float* __restrict__ y2, int ldy2, int rows_out, int D,
                                                        float eps, int drop_cls, int ntok, int ntok_s,
                                                        T* __restrict__ y_lo) {
  wvn_fp16_saturate();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows_out) return;
  size_t in_row = row;
  if (drop_cls) {
    int np = ntok - 1;
    int b = row / np, pp = row - b * np;
    in_row = (size_t)b * ntok_s + 1 + pp;
  }
  const float* xr = x + in_row * D;
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    v[i] = xr[lane + 64 * i];
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    float d = v[i] - mean;
    q += d * d;
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int c = lane + 64 * i;
    float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
    if (y) ElemIO<T>::store(y + (size_t)row * ldy + c, o);
    if constexpr (std::is_same<T, bf16_t>::value) {  // exact mode: lo plane of the same value
      if (y_lo) y_lo[(size_t)row * ldy + c] = f32_to_bf16(o - bf16_to_f32(f32_to_bf16(o)));
    }
    if (y2) y2[(size_t)row * ldy2 + c] = o;
  }
}

// D = 384 with bf16 output (LN1 / LN2 of every ViT-S block: 24 launches per forward, HBM-bound at 1536 B read + 768 B
// written per row).  16 lanes per row, each lane owns 3 groups of 8 consecutive columns: every access is a 16-byte load or
// store, a wave handles 4 rows per pass and LN_RPW rows in all (gamma / beta stay in registers).  Same two-pass fp32
// statistics as the generic kernel.
constexpr int LN_RPW = 16;
template <bool PLANES, bool F16 = false>  // PLANES: exact mode, y_lo receives the lo plane (x - bf16(x)) of every output value; F16: fp16 output
__global__ __launch_bounds__(256) void layernorm384_bf16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                bf16_t* __restrict__ y_lo, int ldy, int rows, float eps) {
  wvn_fp16_saturate();
  const int lane = threadIdx.x & 63, sub = lane & 15, rsel = lane >> 4;
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  f32x4_t gm[6], bt[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gm[2 * i] = *(const f32x4_t*)(gamma + 128 * i + 8 * sub);
    gm[2 * i + 1] = *(const f32x4_t*)(gamma + 128 * i + 8 * sub + 4);
    bt[2 * i] = *(const f32x4_t*)(beta + 128 * i + 8 * sub);
    bt[2 * i + 1] = *(const f32x4_t*)(beta + 128 * i + 8 * sub + 4);
  }
  auto sum16 = [](float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
#pragma unroll 2
  for (int it = 0; it < LN_RPW / 4; ++it) {
    const int row = wg * LN_RPW + it * 4 + rsel;
    const bool ok = row < rows;
    const float* xr = x + (size_t)(ok ? row : rows - 1) * 384 + 8 * sub;
    f32x4_t v[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[2 * i] = *(const f32x4_t*)(xr + 128 * i);
      v[2 * i + 1] = *(const f32x4_t*)(xr + 128 * i + 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = sum16(s) / 384.f;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    const float rstd = 1.0f / sqrtf(sum16(q) / 384.f + eps);
    if (ok) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[2 * i + (e >> 2)][e & 3] - mean) * rstd * gm[2 * i + (e >> 2)][e & 3] + bt[2 * i + (e >> 2)][e & 3];
        auto pk = [](float a, float b2) { return F16 ? pack_f16x2(a, b2) : pack_bf16x2(a, b2); };
        const u32x4_t u = {pk(o[0], o[1]), pk(o[2], o[3]), pk(o[4], o[5]), pk(o[6], o[7])};
        *(u32x4_t*)(y + (size_t)row * ldy + 128 * i + 8 * sub) = u;
        if constexpr (PLANES) {
          u32x4_t w;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            w[e] = pack_bf16x2(o[2 * e] - __uint_as_float(u[e] << 16), o[2 * e + 1] - __uint_as_float(u[e] & 0xffff0000u));
          *(u32x4_t*)(y_lo + (size_t)row * ldy + 128 * i + 8 * sub) = w;
        }
      }
    }
  }
}

template <bool F16>
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ dst, int ldd,
                                     int rows, int cols) {
  wvn_fp16_saturate();
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  const float v = src[(size_t)r * lds_ + c];
  dst[(size_t)r * ldd + c] = F16 ? f32_to_f16(v) : f32_to_bf16(v);
}

// fp32 [rows, cols] -> hi / lo bf16 planes (exact-mode operands of gemm_x3.hip)
__global__ void split_planes_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                    int ldd, int rows, int cols) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  const float v = src[(size_t)r * lds_ + c];
  const bf16_t h = f32_to_bf16(v);
  hi[(size_t)r * ldd + c] = h;
  lo[(size_t)r * ldd + c] = f32_to_bf16(v - bf16_to_f32(h));
}

// ---------------------------------------------------------------------------------------------
// Bilinear (align_corners=True) upsample of a token-major feature map to NCHW, the tensor
// DinoInterface.inference returns (dino_interface.py:87-90).  HBM-write bound (308 MB/frame at
// 448^2 x 384): one block per (b, output row y, 32-channel slab) stages the two source rows it
// needs (2 x G x 32 floats) in LDS and emits full 448-float rows with coalesced stores.
// ATen semantics: src = dst * (G-1)/(H-1) (fp32), i0 = (int)src, i1 = i0 + (i0 < G-1), w1 = src - i0.
// ---------------------------------------------------------------------------------------------
constexpr int UP_CH = 32;
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ tok, float* __restrict__ out, int B,
                                                       int G, int D, int H) {
  extern __shared__ float rows[];  // [2][G][UP_CH+1]
  const int y = blockIdx.x, cs = blockIdx.y * UP_CH, b = blockIdx.z;
  const float scale = lerp_scale(G, H);
  const LerpTap ty = lerp_tap(y, G, scale);
  const int y0 = ty.i0, y1 = ty.i1;
  const int tid = threadIdx.x;
  const int nload = 2 * G * UP_CH;
  for (int i = tid; i < nload; i += blockDim.x) {
    int c = i % UP_CH;
    int gx = (i / UP_CH) % G;
    int r = i / (UP_CH * G);
    int gy = r ? y1 : y0;
    int ch = cs + c;
    rows[(r * G + gx) * (UP_CH + 1) + c] = (ch < D) ? tok[((size_t)b * G * G + gy * G + gx) * D + ch] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < UP_CH * H; i += blockDim.x) {
    int xo = i % H, c = i / H;
    if (cs + c >= D) continue;
    const LerpTap tx = lerp_tap(xo, G, scale);
    out[(((size_t)b * D + cs + c) * H + y) * H + xo] =
        bilerp_fixed(rows[(0 * G + tx.i0) * (UP_CH + 1) + c], rows[(0 * G + tx.i1) * (UP_CH + 1) + c],
                     rows[(1 * G + tx.i0) * (UP_CH + 1) + c], rows[(1 * G + tx.i1) * (UP_CH + 1) + c], tx.w0, tx.w1, ty.w0, ty.w1);
  }
}

// nearest upsample of an integer label grid [B,G,G] -> [B,H,H]  (stego_interface.py:108-109;
// ATen nearest: src = min((int)floorf(dst * (float)G / H), G-1))
__global__ void upsample_nearest_i32_kernel(const int* __restrict__ lab, int* __restrict__ out, int B, int G, int H) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * H) return;
  int x = (int)(i % H);
  int y = (int)((i / H) % H);
  int b = (int)(i / ((long long)H * H));
  const float sc = (float)G / (float)H;
  int sy = min((int)floorf((float)y * sc), G - 1);
  int sx = min((int)floorf((float)x * sc), G - 1);
  out[i] = lab[((size_t)b * G + sy) * G + sx];
}

// Any even patch size (DINOv2's 14), 16-bit output, optionally through the ingest tables: one workgroup per (frame, patch row).  The strip of 3 x P image rows is
// normalised and rounded on the way into LDS with coalesced reads along x; then every patch row of the matrix (3 P P contiguous values) leaves as consecutive
// 4-byte stores.  (The per-thread form above writes 2 P bytes per thread at a stride of a whole patch row: 195 us for 16 frames of 518^2, this one ~25.)
template <typename TIN, bool F16>
__global__ __launch_bounds__(256) void patchify_strip_kernel(const TIN* __restrict__ img, bf16_t* __restrict__ out, int ldp, int S, int P, Gather gt) {
  wvn_fp16_saturate();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ps[];
  bf16_t* strip = (bf16_t*)smem_ps;                     // [3][P][S]
  const int gy = blockIdx.x, b = blockIdx.y, G = S / P;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const int Hs = gt.rows ? gt.Hs : S, Ws = gt.rows ? gt.Ws : S;
  const int nrow = 3 * P;
  for (int r = threadIdx.x >> 6; r < nrow; r += 4) {    // a wave per image row of the strip
    const int c = r / P, py = r - c * P;
    const int y = gy * P + py;
    const TIN* src = img + (((size_t)b * 3 + c) * Hs + (gt.rows ? gt.rows[y] : y)) * Ws;
    for (int x = threadIdx.x & 63; x < S; x += 64) {
      const float v = (load_pixel(src + (gt.cols ? gt.cols[x] : x)) - mean[c]) / stdv[c];
      strip[r * S + x] = F16 ? (bf16_t)f32_to_f16(v) : f32_to_bf16(v);
    }
  }
  __syncthreads();
  const int KP = 3 * P * P, half = KP / 2;              // pairs (e, e + 1): P even -> both in the same image row
  for (int i = threadIdx.x; i < G * half; i += 256) {
    const int gx = i / half, e = 2 * (i - gx * half);
    const int r = e / P, px = e - r * P;                // r = c P + py
    const unsigned v = *(const unsigned*)(strip + r * S + gx * P + px);
    *(unsigned*)(out + ((size_t)b * G * G + (size_t)gy * G + gx) * ldp + e) = v;
  }
}

}  // namespace

// out_mode: 0 fp32, 1 bf16, 2 hi/lo bf16 planes (exact mode; lo plane = patches_lo), 3 fp16.  ldp: row stride of the patch
// matrix in elements (0 = 3*P*P).  P in {8, 14, 16}.  ing != nullptr: the frames are [B,3,src_h,src_w] and network pixel (y, x)
// is frame pixel (rows[y], cols[x]) -- NEAREST resize + centre crop fused into the patch gather.
template <typename TIN>
static int patchify_any(const TIN* img, void* patches, void* patches_lo, int out_mode, int ldp, int B, int S, int P, const Gather& gt,
                        hipStream_t st) {
  const int G = S / P, KP = 3 * P * P;
  const long long total = (long long)B * G * G * 3 * P;
  dim3 grid((unsigned)((total + 255) / 256));
  bf16_t* lo = out_mode == 2 ? (bf16_t*)patches_lo : nullptr;
  const bool rows_ok = P == 8 && ldp == KP && (S % 8) == 0 && (((uintptr_t)patches) & 15) == 0 && (S / 8) * 192 * 2 <= 64 * 1024;
  const size_t shm = (size_t)(S / 8) * 192 * 2;
  if (rows_ok && out_mode == 2 && gt.rows && (((uintptr_t)patches_lo) & 15) == 0 && 2 * shm <= 64 * 1024) {   // hi / lo planes through the ingest tables
    hipLaunchKernelGGL((patchify8_gather_rows_kernel<TIN, false, true>), dim3(S / 8, B), dim3(256), 2 * shm, st, img, (bf16_t*)patches, S, gt, lo);
    return WVN_OK;
  }
  if (rows_ok && (out_mode == 1 || out_mode == 3)) {
    if (gt.rows) {
      if (out_mode == 3) hipLaunchKernelGGL((patchify8_gather_rows_kernel<TIN, true>), dim3(S / 8, B), dim3(256), shm, st, img, (bf16_t*)patches, S, gt, (bf16_t*)nullptr);
      else hipLaunchKernelGGL((patchify8_gather_rows_kernel<TIN, false>), dim3(S / 8, B), dim3(256), shm, st, img, (bf16_t*)patches, S, gt, (bf16_t*)nullptr);
      return WVN_OK;
    }
    if ((((uintptr_t)img) & (sizeof(TIN) == 1 ? 3 : 15)) == 0) {
      if (out_mode == 3) hipLaunchKernelGGL((patchify8_bf16_rows_kernel<TIN, true>), dim3(S / 8, B), dim3(256), shm, st, img, (bf16_t*)patches, S);
      else hipLaunchKernelGGL((patchify8_bf16_rows_kernel<TIN, false>), dim3(S / 8, B), dim3(256), shm, st, img, (bf16_t*)patches, S);
      return WVN_OK;
    }
  }
  // 16-bit output of any even patch size through the strip kernel (a patch row of the matrix 4-byte aligned: ldp even)
  if ((out_mode == 1 || out_mode == 3) && (P % 2) == 0 && (ldp % 2) == 0 && (((uintptr_t)patches) & 3) == 0 && (size_t)3 * P * S * 2 <= 64 * 1024 && !getenv("WVN_NO_PATCHIFY_STRIP")) {
    const size_t sh = (size_t)3 * P * S * 2;
    if (out_mode == 3) hipLaunchKernelGGL((patchify_strip_kernel<TIN, true>), dim3(G, B), dim3(256), sh, st, img, (bf16_t*)patches, ldp, S, P, gt);
    else hipLaunchKernelGGL((patchify_strip_kernel<TIN, false>), dim3(G, B), dim3(256), sh, st, img, (bf16_t*)patches, ldp, S, P, gt);
    return WVN_OK;
  }
#define WVN_PATCHIFY_P(PP)                                                                                                              \
  do {                                                                                                                                 \
    if (out_mode == 3) hipLaunchKernelGGL((patchify_kernel<f16raw_t, PP, TIN>), grid, dim3(256), 0, st, img, (f16raw_t*)patches,        \
                                          (f16raw_t*)nullptr, ldp, B, S, gt);                                                          \
    else if (out_mode) hipLaunchKernelGGL((patchify_kernel<bf16_t, PP, TIN>), grid, dim3(256), 0, st, img, (bf16_t*)patches, lo, ldp,   \
                                          B, S, gt);                                                                                   \
    else hipLaunchKernelGGL((patchify_kernel<float, PP, TIN>), grid, dim3(256), 0, st, img, (float*)patches, (float*)nullptr, ldp, B,   \
                            S, gt);                                                                                                    \
  } while (0)
  if (P == 8) WVN_PATCHIFY_P(8);
  else if (P == 16) WVN_PATCHIFY_P(16);
  else if (P == 14) WVN_PATCHIFY_P(14);
  else return WVN_ERR_ARG;
#undef WVN_PATCHIFY_P
  return WVN_OK;
}

int wvn_patchify_launch(const void* img_v, int img_u8, void* patches, void* patches_lo, int out_mode, int ldp, int B, int S, int P,
                        hipStream_t st, const WvnIngest* ing) {
  const int KP = 3 * P * P;
  if (ldp == 0) ldp = KP;
  if (!img_v || !patches || B <= 0 || P <= 0 || S % P != 0 || out_mode < 0 || out_mode > 3 || ldp < KP ||
      (out_mode == 2 && !patches_lo))
    return WVN_ERR_ARG;
  Gather gt{nullptr, nullptr, 0, 0};
  if (ing) {
    if (!ing->rows || !ing->cols || ing->src_h <= 0 || ing->src_w <= 0) return WVN_ERR_ARG;
    gt = Gather{ing->rows, ing->cols, ing->src_h, ing->src_w};
  }
  const int rc = img_u8 ? patchify_any((const unsigned char*)img_v, patches, patches_lo, out_mode, ldp, B, S, P, gt, st)
                        : patchify_any((const float*)img_v, patches, patches_lo, out_mode, ldp, B, S, P, gt, st);
  if (rc != WVN_OK) return rc;
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// out[planes, out_h, out_w] = in[planes, src_h, src_w] gathered through the ingest tables; elem_bytes 1 (uint8) or 4 (fp32 / int32)
int wvn_gather_image_launch(const void* in, void* out, long long planes, int out_h, int out_w, int elem_bytes, const WvnIngest* ing,
                            hipStream_t st) {
  if (!in || !out || !ing || !ing->rows || !ing->cols || planes <= 0 || out_h <= 0 || out_w <= 0 || ing->src_h <= 0 || ing->src_w <= 0)
    return WVN_ERR_ARG;
  const Gather gt{ing->rows, ing->cols, ing->src_h, ing->src_w};
  const long long n = planes * out_h * out_w;
  dim3 grid((unsigned)((n + 255) / 256));
  if (elem_bytes == 1) hipLaunchKernelGGL(gather_image_kernel<unsigned char>, grid, dim3(256), 0, st, (const unsigned char*)in, (unsigned char*)out, planes, out_h, out_w, gt);
  else if (elem_bytes == 4) hipLaunchKernelGGL(gather_image_kernel<unsigned int>, grid, dim3(256), 0, st, (const unsigned int*)in, (unsigned int*)out, planes, out_h, out_w, gt);
  else return WVN_ERR_ARG;
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// zero the byte range [col0, col0 + ncol) of every row (4-byte granularity): padding hygiene of the ViT workspace
__global__ void pad_zero_kernel(unsigned char* __restrict__ base, long long nrows, long long row_stride, long long col0,
                                int ncol_words) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * ncol_words) return;
  const long long r = i / ncol_words;
  const int w = (int)(i - r * ncol_words);
  *(unsigned int*)(base + r * row_stride + col0 + (long long)w * 4) = 0u;
}

int wvn_pad_zero_launch(void* base, long long nrows, long long row_stride_bytes, long long col0_bytes,
                        long long ncol_bytes, hipStream_t st) {
  if (!base || (row_stride_bytes & 3) || (col0_bytes & 3) || (ncol_bytes & 3)) return WVN_ERR_ARG;
  if (nrows <= 0 || ncol_bytes <= 0) return WVN_OK;
  const long long n = nrows * (ncol_bytes / 4);
  hipLaunchKernelGGL(pad_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned char*)base, nrows,
                     row_stride_bytes, col0_bytes, (int)(ncol_bytes / 4));
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_cls_rows_launch(const float* cls_pos, float* x, int B, int ntok, int D, hipStream_t st) {  // ntok = rows per frame
  hipLaunchKernelGGL(cls_rows_kernel, dim3(ceil_div(B * D, 256)), dim3(256), 0, st, cls_pos, x, B, ntok, D);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

template <typename T>
static int ln_dispatch(const float* x, const float* g, const float* b, T* y, int ldy, float* y2, int ldy2, int rows_out,
                       int D, float eps, int drop_cls, int ntok, int ntok_s, T* y_lo, hipStream_t st) {
  dim3 grid(ceil_div(rows_out, 4)), block(256);
  switch (D / 64) {
    case 6: hipLaunchKernelGGL((layernorm_kernel<T, 6>), grid, block, 0, st, x, g, b, y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, y_lo); break;
    case 12: hipLaunchKernelGGL((layernorm_kernel<T, 12>), grid, block, 0, st, x, g, b, y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, y_lo); break;
    case 16: hipLaunchKernelGGL((layernorm_kernel<T, 16>), grid, block, 0, st, x, g, b, y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, y_lo); break;
    case 1: hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, st, x, g, b, y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, y_lo); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, st, x, g, b, y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, y_lo); break;
    default: return WVN_ERR_ARG;
  }
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// y_fmt: 0 fp32, 1 bf16, 2 fp16 output rows.  y_lo != nullptr (with y_fmt == 1): exact mode, y / y_lo receive the hi / lo planes
int wvn_layernorm_launch(const float* x, const float* gamma, const float* beta, void* y, int y_fmt, int ldy,
                         float* y2, int ldy2, int rows_out, int D, float eps, int drop_cls, int ntok,
                         int ntok_s, hipStream_t st, void* y_lo) {
  if (!x || !gamma || !beta || (D % 64) != 0 || rows_out <= 0 || y_fmt < 0 || y_fmt > 2 || (y_lo && (y_fmt != 1 || !y))) return WVN_ERR_ARG;
  if (y_fmt && y && !y2 && !drop_cls && D == 384 && (ldy % 8) == 0 &&
      ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)y_lo) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0) {
    const dim3 grid(ceil_div(rows_out, 4 * LN_RPW));
    if (y_lo)
      hipLaunchKernelGGL((layernorm384_bf16_kernel<true>), grid, dim3(256), 0, st, x, gamma, beta, (bf16_t*)y, (bf16_t*)y_lo, ldy, rows_out, eps);
    else if (y_fmt == 2)
      hipLaunchKernelGGL((layernorm384_bf16_kernel<false, true>), grid, dim3(256), 0, st, x, gamma, beta, (bf16_t*)y, (bf16_t*)nullptr, ldy, rows_out, eps);
    else
      hipLaunchKernelGGL((layernorm384_bf16_kernel<false>), grid, dim3(256), 0, st, x, gamma, beta, (bf16_t*)y, (bf16_t*)nullptr, ldy, rows_out, eps);
    WVN_LAUNCH_CHECK();
    return WVN_OK;
  }
  if (y_fmt == 2) return ln_dispatch<f16raw_t>(x, gamma, beta, (f16raw_t*)y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, (f16raw_t*)nullptr, st);
  if (y_fmt == 1) return ln_dispatch<bf16_t>(x, gamma, beta, (bf16_t*)y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, (bf16_t*)y_lo, st);
  return ln_dispatch<float>(x, gamma, beta, (float*)y, ldy, y2, ldy2, rows_out, D, eps, drop_cls, ntok, ntok_s, (float*)nullptr, st);
}

int wvn_cast_f32_bf16_launch(const float* src, int lds_, bf16_t* dst, int ldd, int rows, int cols, hipStream_t st, int f16) {
  long long n = (long long)rows * cols;
  if (f16) hipLaunchKernelGGL(cast_f32_bf16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
  else hipLaunchKernelGGL(cast_f32_bf16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_split_planes_launch(const float* src, int lds_, bf16_t* hi, bf16_t* lo, int ldd, int rows, int cols, hipStream_t st) {
  if (!src || !hi || !lo || rows <= 0 || cols <= 0) return WVN_ERR_ARG;
  long long n = (long long)rows * cols;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds_, hi, lo, ldd, rows, cols);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_upsample_bilinear_launch(const float* tok, float* out, int B, int G, int D, int H, hipStream_t st) {
  if (!tok || !out) return WVN_ERR_ARG;
  size_t shm = (size_t)2 * G * (UP_CH + 1) * sizeof(float);
  hipLaunchKernelGGL(upsample_kernel, dim3(H, ceil_div(D, UP_CH), B), dim3(256), shm, st, tok, out, B, G, D, H);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_upsample_nearest_i32_launch(const int* lab, int* out, int B, int G, int H, hipStream_t st) {
  long long n = (long long)B * H * H;
  hipLaunchKernelGGL(upsample_nearest_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lab, out, B, G, H);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// (test hook) out[i] = fp16(in[i]) through the path's converters under wvn_fp16_saturate
__global__ void f16_saturate_probe_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, int n) {
  wvn_fp16_saturate();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t pk = pack_f16x2(in[i], in[i]);
    out[i] = (i & 1) ? (uint16_t)(pk >> 16) : f32_to_f16(in[i]);
  }
}
int wvn_f16_saturate_probe_launch(const float* in, uint16_t* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(f16_saturate_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, st, in, out, n);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
