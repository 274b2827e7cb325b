// Fused per-pixel traversability inference: what the live node runs for every camera frame with
// prediction_per_pixel (wvn_feature_extractor_node.py:319-363, quick_start.py:183-210):
//
//   dense = bilinear_align_corners(tokens [G,G,384] -> [H,W,384])       (dino_interface.py:87-90, 308 MB/frame at 448^2)
//   out   = SimpleMLP(dense rows)   384 -> 256 -> 32 -> 1+384            (simple_mlp.py:10-39, 47.7 GFLOP/frame)
//   trav  = sigmoid(out[:,0]);  conf = confidence(mean((out[:,1:] - dense)^2))   (confidence_generator.py:182-193)
//
// Here the dense tensor is never built and the 384->256 layer never runs per pixel:
//   * layer 1 is linear before its ReLU, and bilinear weights are a convex combination, so
//       W1 * interp(tokens) + b1 == interp(W1 * tokens) + b1:
//     Z = tokens * W1^T is one [G*G,384]x[384,256] GEMM at TOKEN resolution (64x fewer rows), written next to the tokens
//     (row layout  [ Z 256 | x 384 ]  bf16, "zx");
//   * the interpolation itself is an MFMA: a 16x16-pixel tile touches at most 4x4 tokens (checked on the host), so
//       interp^T [channels x pixels] = zx_window^T [channels x 16 tokens] * weights [16 tokens x pixels]
//     is ONE K=16 step of v_mfma_f32_32x32x16_bf16 per 32 channels, with the bilinear weights (4 non-zeros per pixel)
//     built in registers.  Weights are split hi+lo (two MFMAs) so that they carry 16 mantissa bits: the interpolation
//     stays a partition of unity to 2^-17 instead of 2^-9;
//   * everything is kept TRANSPOSED (lane = pixel, registers = channels).  The accumulator layout of one MFMA is then
//     exactly the B-operand layout of the next one up to a fixed permutation of k, which is folded into the packed
//     weight images (wvn_pixel_mlp_pack) -- activations never leave registers, there is no LDS round trip between layers;
//   * the reconstruction error comes out of the MFMA chain directly: acc = b3 + W3 h2 - interp(x) (negated weights), so the
//     epilogue per 32 channels is 16 squares.
// Per 32 pixels (one wave): 16 (Z interp) + 16 (layer 2) + 12 x 4 (layer 3 + x interp) + 2 (traversability row) = 82 MFMAs
// = 2.7 MFLOP of MFMA work instead of 7.6 MFLOP, and ~20 KB of L2 reads per 256 pixels instead of 1.5 KB of HBM per pixel.
//
// bf16 operands, fp32 accumulation: this is the speed mode.  The exact (fp32 FMA) mode is wvn_upsample_bilinear +
// wvn_mlp_forward + wvn_mlp_confidence.
#include <cstdlib>

#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr int H1 = 256, H2 = 32;
constexpr int W2_BYTES = 16 * 2 * 32 * 16;          // [16 k-steps][2 lane halves][32 rows][8 bf16] = 16,384
constexpr int TILE = 16;                            // pixels per tile edge; a wave owns 2 rows x 16 columns

// D = MLP input size: 384 (DINO ViT-S features) or 90 (STEGO code, the live node's default feature_type).  For D = 90 the
// x part of a zx row is zero-padded to 128 columns (K of the layer-1 GEMM) and three 32-channel blocks are interpolated.
template <int D>
struct Cfg {
  static constexpr int DREAL = D;
  static constexpr int NT = (D + 31) / 32;              // 32-channel reconstruction tiles (12 / 3)
  static constexpr int DX = D == 384 ? 384 : 128;       // x columns of a zx row
  static constexpr int ZXC = H1 + DX;                   // zx row length the caller provides (640 / 384)
  static constexpr int NCH = H1 + 32 * NT;              // channels staged and interpolated per token (640 / 352)
  static constexpr int PLANE = NCH * 16 + 64;           // bytes of one 8-token plane [NCH ch][8 tok] (+64: planes 16 banks apart)
  static constexpr int TOK_BYTES = 2 * PLANE;
  static constexpr int W3_TILES = NT + 1;               // + 1 tile whose row 0 is the traversability unit
  static constexpr int W3_BYTES = W3_TILES * 2 * 2 * 32 * 16;
  static constexpr int NBIAS = H1 + H2 + W3_TILES * 32; // b1 | b2 | b3[1:] (padded) | (b3[0], 31 zeros)
  static constexpr int W1_BYTES = H1 * DX * 2;          // packed blob starts with W1 as bf16 [256][DX] (the Z GEMM's weight)
  static constexpr int WIMG_BYTES = W2_BYTES + W3_BYTES + NBIAS * 4;  // what the kernel copies to LDS
  static constexpr int OFF_W2 = TOK_BYTES;
  static constexpr int OFF_W3 = OFF_W2 + W2_BYTES;
  static constexpr int OFF_BIAS = OFF_W3 + W3_BYTES;
  static constexpr int LDS_BYTES = OFF_BIAS + NBIAS * 4; // 66,432 at D = 384: two workgroups per CU
  static constexpr int NFETCH = 16 * (NCH / 8);          // 16-byte chunks of one token window
  static constexpr int NPRE = (NFETCH + 511) / 512;
  // exact mode (hi + lo): lo planes after the hi planes, then W2H | W3H | W2L | W3L | bias
  static constexpr int XTOKL = TOK_BYTES;
  static constexpr int XOFFW = 2 * TOK_BYTES;
  static constexpr int XW2H = XOFFW, XW3H = XW2H + W2_BYTES, XW2L = XW3H + W3_BYTES, XW3L = XW2L + W2_BYTES;
  static constexpr int XBIAS = XW3L + W3_BYTES;
  static constexpr int XWIMG_BYTES = 2 * (W2_BYTES + W3_BYTES) + NBIAS * 4;
  static constexpr int XLDS_BYTES = XBIAS + NBIAS * 4;   // 130,048 at D = 384
  static constexpr int XNPRE = (2 * NFETCH + 511) / 512;
};

struct PixParams {
  const bf16_t* zx; int ldzx;       // [B*G*G][ldzx]: columns [0,256) = Z, [256,640) = tokens
  const unsigned char* wimg;        // W2 image | W3 image | biases (WIMG_BYTES)
  float* trav; float* conf; float* loss;
  int B, G, Ho, Wo, nty, ntx;
  float sy, sx;                     // (G-1)/(Ho-1), (G-1)/(Wo-1)
  float mean, std, std_factor;
  const float* conf_dev;            // optional {mean, std, std_factor} in device memory (overrides the three scalars)
};

// confidence_generator.py:182-193 (same arithmetic as mlp.hip's row kernel)
__device__ inline float pix_confidence(float x, float mean, float std, float f) {
  const float shifted = mean + std * f;
  float lo = shifted - std;
  lo = (lo > 0.f || isnan(lo)) ? lo : 0.f;
  const float hi = shifted + std;
  float xc = fminf(fmaxf(x, lo), hi);
  if (isnan(lo) || isnan(hi) || isnan(x)) xc = NAN;
  return 1.f - (xc - lo) / (hi - lo);
}

__device__ inline f32x16_t bias16(const float* b, int h) {  // accumulator register 4i+j <-> row 8i + 4h + j
  f32x16_t a;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4_t v = *(const f32x4_t*)(b + 8 * i + 4 * h);
    a[4 * i + 0] = v[0]; a[4 * i + 1] = v[1]; a[4 * i + 2] = v[2]; a[4 * i + 3] = v[3];
  }
  return a;
}

// relu + bf16 pack of 8 accumulator registers: element e of the B fragment = register r0 + e
__device__ inline bf16x8_t relu_pack8(const f32x16_t& a, int r0) {
  u32x4_t u;
#pragma unroll
  for (int q = 0; q < 4; ++q) u[q] = pack_bf16x2(fmaxf(a[r0 + 2 * q], 0.f), fmaxf(a[r0 + 2 * q + 1], 0.f));
  return __builtin_bit_cast(bf16x8_t, u);
}

template <int WSPLIT, int D>
__global__ __launch_bounds__(512, 2) void pixel_mlp_kernel(PixParams p) {
  using K = Cfg<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_frame = p.nty * p.ntx;
  const int ntiles = p.B * tiles_per_frame;

  for (int i = tid; i < K::WIMG_BYTES / 16; i += 512) *(u32x4_t*)(smem + K::OFF_W2 + i * 16) = ((const u32x4_t*)p.wimg)[i];
  const float* bias_l = (const float*)(smem + K::OFF_BIAS);

  // this thread's share of a token window: chunks tid, tid + 512, tid + 1024 of [16 tokens][80 x 16 B]
  u32x4_t pre[K::NPRE];
  auto fetch = [&](int tile) {
    const int b = tile / tiles_per_frame, r = tile - b * tiles_per_frame;
    const int tyi = r / p.ntx, txi = r - tyi * p.ntx;
    const int by = (int)(p.sy * (float)(tyi * TILE)), bx = (int)(p.sx * (float)(txi * TILE));
#pragma unroll
    for (int k = 0; k < K::NPRE; ++k) {
      const int idx = tid + 512 * k;
      if (idx < K::NFETCH) {
        const int tok = idx & 15, chunk = idx >> 4;
        const int gy = min(by + (tok >> 2), p.G - 1), gx = min(bx + (tok & 3), p.G - 1);
        pre[k] = *(const u32x4_t*)(p.zx + ((size_t)b * p.G * p.G + (size_t)gy * p.G + gx) * p.ldzx + chunk * 8);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int k = 0; k < K::NPRE; ++k) {
      const int idx = tid + 512 * k;
      if (idx < K::NFETCH) {
        const int tok = idx & 15, chunk = idx >> 4;
        unsigned char* dst = smem + (tok >> 3) * K::PLANE + chunk * 128 + (tok & 7) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) *(bf16_t*)(dst + e * 16) = (bf16_t)(pre[k][e >> 1] >> ((e & 1) * 16));
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    stash();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);

    const int b = tile / tiles_per_frame, r = tile - b * tiles_per_frame;
    const int tyi = r / p.ntx, txi = r - tyi * p.ntx;
    const int by = (int)(p.sy * (float)(tyi * TILE)), bx = (int)(p.sx * (float)(txi * TILE));
    const int py = tyi * TILE + 2 * wave + (n >> 4), px = txi * TILE + (n & 15);
    // ATen's align_corners bilinear: src = dst * (G-1)/(H-1), i0 = (int)src, w1 = src - i0, w0 = 1 - w1
    const float fsy = p.sy * (float)min(py, p.Ho - 1), fsx = p.sx * (float)min(px, p.Wo - 1);
    const int gy0 = (int)fsy, gx0 = (int)fsx;
    const float wy1 = fsy - (float)gy0, wx1 = fsx - (float)gx0;
    const int ty0 = gy0 - by, tx0 = gx0 - bx;
    float wyv[2], wxv[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) wyv[q] = ((2 * h + q) == ty0 ? 1.f - wy1 : 0.f) + ((2 * h + q) == ty0 + 1 ? wy1 : 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) wxv[q] = (q == tx0 ? 1.f - wx1 : 0.f) + (q == tx0 + 1 ? wx1 : 0.f);
    // B fragment of the interpolation: k slot 8h + e = token (row 2h + (e >> 2), column e & 3) of the 4x4 window
    u32x4_t whi_u, wlo_u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float w0 = wyv[q >> 1] * wxv[(2 * q) & 3], w1 = wyv[q >> 1] * wxv[(2 * q + 1) & 3];
      const uint32_t hi = pack_bf16x2(w0, w1);
      whi_u[q] = hi;
      wlo_u[q] = pack_bf16x2(w0 - __uint_as_float(hi << 16), w1 - __uint_as_float(hi & 0xffff0000u));
    }
    const bf16x8_t whi = __builtin_bit_cast(bf16x8_t, whi_u), wlo = __builtin_bit_cast(bf16x8_t, wlo_u);
    const bf16x8_t nhi = __builtin_bit_cast(bf16x8_t, whi_u ^ 0x80008000u), nlo = __builtin_bit_cast(bf16x8_t, wlo_u ^ 0x80008000u);

    const unsigned char* tokl = smem + h * K::PLANE + n * 16;         // + 512 per 32-channel block
    const unsigned char* w2l = smem + K::OFF_W2 + (h * 32 + n) * 16;  // + 1024 per k-step
    const unsigned char* w3l = smem + K::OFF_W3 + (h * 32 + n) * 16;  // + 1024 per (tile, k-step)

    // ---- layers 1 + 2: h1 block = relu(b1 + interp(Z block)); a2 += W2[:, block] * h1 block
    f32x16_t a2 = bias16(bias_l + H1, h);
#pragma unroll
    for (int blk = 0; blk < H1 / 32; ++blk) {
      f32x16_t az = bias16(bias_l + 32 * blk, h);
      const bf16x8_t tf = *(const bf16x8_t*)(tokl + blk * 512);
      az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, whi, az, 0, 0, 0);
      if (WSPLIT) az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, wlo, az, 0, 0, 0);
      const bf16x8_t h0 = relu_pack8(az, 0), h1v = relu_pack8(az, 8);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(w2l + (2 * blk) * 1024), h0, a2, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(w2l + (2 * blk + 1) * 1024), h1v, a2, 0, 0, 0);
    }
    const bf16x8_t g0 = relu_pack8(a2, 0), g1 = relu_pack8(a2, 8);

    // ---- layer 3 + reconstruction error, 32 channels at a time: a3 = b3 + W3 h2 - interp(x)
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < K::NT; ++t) {
      f32x16_t a3 = bias16(bias_l + H1 + H2 + 32 * t, h);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(w3l + (2 * t) * 1024), g0, a3, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(w3l + (2 * t + 1) * 1024), g1, a3, 0, 0, 0);
      const bf16x8_t tf = *(const bf16x8_t*)(tokl + (H1 / 32 + t) * 512);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, nhi, a3, 0, 0, 0);
      if (WSPLIT) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, nlo, a3, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 16; ++q) lsum = fmaf(a3[q], a3[q], lsum);
    }
    f32x16_t at = bias16(bias_l + H1 + H2 + 32 * K::NT, h);
    at = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(w3l + (2 * (K::NT)) * 1024), g0, at, 0, 0, 0);
    at = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)(w3l + (2 * (K::NT) + 1) * 1024), g1, at, 0, 0, 0);

    lsum += __shfl_xor(lsum, 32, 64);
    if (h == 0 && py < p.Ho && px < p.Wo) {
      const size_t o = ((size_t)b * p.Ho + py) * p.Wo + px;
      const float lr = lsum / (float)K::DREAL;  // (padded channels contribute exact zeros)
      if (p.trav) p.trav[o] = sigmoid_f(at[0]);
      if (p.loss) p.loss[o] = lr;
      if (p.conf) {
        const float cm = p.conf_dev ? p.conf_dev[0] : p.mean, cs = p.conf_dev ? p.conf_dev[1] : p.std;
        const float cf = p.conf_dev ? p.conf_dev[2] : p.std_factor;
        p.conf[o] = pix_confidence(lr, cm, cs, cf);
      }
    }
    __syncthreads();  // every wave is done with this tile's token image
  }
}

// fp32 flat parameters [W1 | b1 | W2 | b2 | W3 | b3] (Linear layout) -> packed blob.
// k permutations (see the file header): the B fragment of k-step u built from an accumulator holds, in slot (h, e), row
//   16u + 8(e >> 2) + 4h + (e & 3)   of the 32-row block the accumulator covers.
template <int D>
__global__ void pixel_mlp_pack_kernel(const float* __restrict__ prm, unsigned char* __restrict__ out) {
  using K = Cfg<D>;
  const float* W1 = prm;
  const float* b1 = W1 + H1 * D;
  const float* W2 = b1 + H1;
  const float* b2 = W2 + H2 * H1;
  const float* W3 = b2 + H2;
  const float* b3 = W3 + (1 + D) * H2;
  bf16_t* w1o = (bf16_t*)out;
  bf16_t* w2o = (bf16_t*)(out + K::W1_BYTES);
  bf16_t* w3o = (bf16_t*)(out + K::W1_BYTES + W2_BYTES);
  float* bo = (float*)(out + K::W1_BYTES + W2_BYTES + K::W3_BYTES);
  const int gsz = gridDim.x * blockDim.x, g0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = g0; i < H1 * K::DX; i += gsz) {  // [256][DX], zero beyond column D
    const int r = i / K::DX, c = i - r * K::DX;
    w1o[i] = c < D ? f32_to_bf16(W1[r * D + c]) : (bf16_t)0;
  }
  for (int i = g0; i < W2_BYTES / 2; i += gsz) {  // [s][h][m][e]
    const int e = i & 7, m = (i >> 3) & 31, hh = (i >> 8) & 1, s = i >> 9;
    const int c = 32 * (s >> 1) + 16 * (s & 1) + 8 * (e >> 2) + 4 * hh + (e & 3);
    w2o[i] = f32_to_bf16(W2[m * H1 + c]);
  }
  for (int i = g0; i < K::W3_BYTES / 2; i += gsz) {  // [t][u][h][m][e]
    const int e = i & 7, m = (i >> 3) & 31, hh = (i >> 8) & 1, u = (i >> 9) & 1, t = i >> 10;
    const int r = 16 * u + 8 * (e >> 2) + 4 * hh + (e & 3);
    const int row = t < K::NT ? (32 * t + m < D ? 1 + 32 * t + m : -1) : (m == 0 ? 0 : -1);
    w3o[i] = row < 0 ? (bf16_t)0 : f32_to_bf16(W3[row * H2 + r]);
  }
  for (int i = g0; i < K::NBIAS; i += gsz) {
    float v;
    if (i < H1) v = b1[i];
    else if (i < H1 + H2) v = b2[i - H1];
    else if (i < H1 + H2 + 32 * K::NT) v = (i - H1 - H2 < D) ? b3[1 + i - H1 - H2] : 0.f;
    else v = (i == H1 + H2 + 32 * K::NT) ? b3[0] : 0.f;
    bo[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact-mode variant: every bf16 MFMA operand is split into hi + lo (v = hi + lo to 16 mantissa bits) and every product
// is formed as hi*hi + hi*lo + lo*hi (three MFMAs, fp32 accumulation): the result matches the fp32 reference sequence to
// ~1e-5 relative, inside the 1e-3 bar of the exact mode, at 186 instead of 82 MFMAs per 32 pixels.  The token-resolution
// layer-1 GEMM (Z = tokens * W1^T) runs on the exact fp32 FMA path.  Layout: zxh / zxl = hi / lo parts of [ Z | x ].
// ---------------------------------------------------------------------------------------------------------------------
struct PixX3Params {
  const bf16_t* zxh; const bf16_t* zxl; int ldzx;
  const unsigned char* wimg;
  float* trav; float* conf; float* loss;
  int B, G, Ho, Wo, nty, ntx;
  float sy, sx;
  float mean, std, std_factor;
  const float* conf_dev;
};

__device__ inline uint32_t split_lo(float v0, float v1, uint32_t hi) {
  return pack_bf16x2(v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xffff0000u));
}
// relu + hi / lo split of 8 accumulator registers
__device__ inline void relu_split8(const f32x16_t& a, int r0, bf16x8_t& hi, bf16x8_t& lo) {
  u32x4_t uh, ul;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float v0 = fmaxf(a[r0 + 2 * q], 0.f), v1 = fmaxf(a[r0 + 2 * q + 1], 0.f);
    uh[q] = pack_bf16x2(v0, v1);
    ul[q] = split_lo(v0, v1, uh[q]);
  }
  hi = __builtin_bit_cast(bf16x8_t, uh);
  lo = __builtin_bit_cast(bf16x8_t, ul);
}
#define MFMA3(acc, ah, al, bh, bl)                                       \
  do {                                                                   \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0); \
  } while (0)

template <int D>
__global__ __launch_bounds__(512, 1) void pixel_mlp_x3_kernel(PixX3Params p) {
  using K = Cfg<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_frame = p.nty * p.ntx;
  const int ntiles = p.B * tiles_per_frame;

  for (int i = tid; i < K::XWIMG_BYTES / 16; i += 512) *(u32x4_t*)(smem + K::XOFFW + i * 16) = ((const u32x4_t*)p.wimg)[i];
  const float* bias_l = (const float*)(smem + K::XBIAS);

  u32x4_t pre[K::XNPRE];
  auto fetch = [&](int tile) {
    const int b = tile / tiles_per_frame, r = tile - b * tiles_per_frame;
    const int tyi = r / p.ntx, txi = r - tyi * p.ntx;
    const int by = (int)(p.sy * (float)(tyi * TILE)), bx = (int)(p.sx * (float)(txi * TILE));
#pragma unroll
    for (int k = 0; k < K::XNPRE; ++k) {
      const int idx = min(tid + 512 * k, 2 * K::NFETCH - 1);  // (clamped duplicates re-read / rewrite the same bytes)
      const int part = idx >= K::NFETCH, id = idx - part * K::NFETCH;
      const int tok = id & 15, chunk = id >> 4;
      const int gy = min(by + (tok >> 2), p.G - 1), gx = min(bx + (tok & 3), p.G - 1);
      const bf16_t* src = part ? p.zxl : p.zxh;
      pre[k] = *(const u32x4_t*)(src + ((size_t)b * p.G * p.G + (size_t)gy * p.G + gx) * p.ldzx + chunk * 8);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int k = 0; k < K::XNPRE; ++k) {
      const int idx = min(tid + 512 * k, 2 * K::NFETCH - 1);
      const int part = idx >= K::NFETCH, id = idx - part * K::NFETCH;
      const int tok = id & 15, chunk = id >> 4;
      unsigned char* dst = smem + part * K::XTOKL + (tok >> 3) * K::PLANE + chunk * 128 + (tok & 7) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) *(bf16_t*)(dst + e * 16) = (bf16_t)(pre[k][e >> 1] >> ((e & 1) * 16));
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    stash();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);

    const int b = tile / tiles_per_frame, r = tile - b * tiles_per_frame;
    const int tyi = r / p.ntx, txi = r - tyi * p.ntx;
    const int by = (int)(p.sy * (float)(tyi * TILE)), bx = (int)(p.sx * (float)(txi * TILE));
    const int py = tyi * TILE + 2 * wave + (n >> 4), px = txi * TILE + (n & 15);
    const float fsy = p.sy * (float)min(py, p.Ho - 1), fsx = p.sx * (float)min(px, p.Wo - 1);
    const int gy0 = (int)fsy, gx0 = (int)fsx;
    const float wy1 = fsy - (float)gy0, wx1 = fsx - (float)gx0;
    const int ty0 = gy0 - by, tx0 = gx0 - bx;
    float wyv[2], wxv[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) wyv[q] = ((2 * h + q) == ty0 ? 1.f - wy1 : 0.f) + ((2 * h + q) == ty0 + 1 ? wy1 : 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) wxv[q] = (q == tx0 ? 1.f - wx1 : 0.f) + (q == tx0 + 1 ? wx1 : 0.f);
    u32x4_t whi_u, wlo_u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float w0 = wyv[q >> 1] * wxv[(2 * q) & 3], w1 = wyv[q >> 1] * wxv[(2 * q + 1) & 3];
      whi_u[q] = pack_bf16x2(w0, w1);
      wlo_u[q] = split_lo(w0, w1, whi_u[q]);
    }
    const bf16x8_t whi = __builtin_bit_cast(bf16x8_t, whi_u), wlo = __builtin_bit_cast(bf16x8_t, wlo_u);
    const bf16x8_t nhi = __builtin_bit_cast(bf16x8_t, whi_u ^ 0x80008000u), nlo = __builtin_bit_cast(bf16x8_t, wlo_u ^ 0x80008000u);

    const unsigned char* tokh = smem + h * K::PLANE + n * 16;
    const unsigned char* tokl = tokh + K::XTOKL;
    const unsigned lw = (h * 32 + n) * 16;  // lane's fragment inside one [k-step] image block of 1024 bytes

    f32x16_t a2 = bias16(bias_l + H1, h);
#pragma unroll
    for (int blk = 0; blk < H1 / 32; ++blk) {
      f32x16_t az = bias16(bias_l + 32 * blk, h);
      const bf16x8_t th = *(const bf16x8_t*)(tokh + blk * 512), tl = *(const bf16x8_t*)(tokl + blk * 512);
      MFMA3(az, th, tl, whi, wlo);
      bf16x8_t hh[2], hl[2];
      relu_split8(az, 0, hh[0], hl[0]);
      relu_split8(az, 8, hh[1], hl[1]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t wh = *(const bf16x8_t*)(smem + K::XW2H + (2 * blk + u) * 1024 + lw);
        const bf16x8_t wl = *(const bf16x8_t*)(smem + K::XW2L + (2 * blk + u) * 1024 + lw);
        MFMA3(a2, wh, wl, hh[u], hl[u]);
      }
    }
    bf16x8_t gh[2], gl[2];
    relu_split8(a2, 0, gh[0], gl[0]);
    relu_split8(a2, 8, gh[1], gl[1]);

    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < K::NT; ++t) {
      f32x16_t a3 = bias16(bias_l + H1 + H2 + 32 * t, h);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t wh = *(const bf16x8_t*)(smem + K::XW3H + (2 * t + u) * 1024 + lw);
        const bf16x8_t wl = *(const bf16x8_t*)(smem + K::XW3L + (2 * t + u) * 1024 + lw);
        MFMA3(a3, wh, wl, gh[u], gl[u]);
      }
      const bf16x8_t th = *(const bf16x8_t*)(tokh + (H1 / 32 + t) * 512), tl = *(const bf16x8_t*)(tokl + (H1 / 32 + t) * 512);
      MFMA3(a3, th, tl, nhi, nlo);
#pragma unroll
      for (int q = 0; q < 16; ++q) lsum = fmaf(a3[q], a3[q], lsum);
    }
    f32x16_t at = bias16(bias_l + H1 + H2 + 32 * K::NT, h);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bf16x8_t wh = *(const bf16x8_t*)(smem + K::XW3H + (2 * (K::NT) + u) * 1024 + lw);
      const bf16x8_t wl = *(const bf16x8_t*)(smem + K::XW3L + (2 * (K::NT) + u) * 1024 + lw);
      MFMA3(at, wh, wl, gh[u], gl[u]);
    }

    lsum += __shfl_xor(lsum, 32, 64);
    if (h == 0 && py < p.Ho && px < p.Wo) {
      const size_t o = ((size_t)b * p.Ho + py) * p.Wo + px;
      const float lr = lsum / (float)K::DREAL;
      if (p.trav) p.trav[o] = sigmoid_f(at[0]);
      if (p.loss) p.loss[o] = lr;
      if (p.conf) {
        const float cm = p.conf_dev ? p.conf_dev[0] : p.mean, cs = p.conf_dev ? p.conf_dev[1] : p.std;
        const float cf = p.conf_dev ? p.conf_dev[2] : p.std_factor;
        p.conf[o] = pix_confidence(lr, cm, cs, cf);
      }
    }
    __syncthreads();
  }
}
#undef MFMA3

// packed blob of the exact mode: W2 hi | W3 hi | W2 lo | W3 lo | biases (same fragment order as the bf16 pack)
template <int D>
__global__ void pixel_mlp_pack_x3_kernel(const float* __restrict__ prm, unsigned char* __restrict__ out) {
  using K = Cfg<D>;
  const float* W1 = prm;
  const float* b1 = W1 + H1 * D;
  const float* W2 = b1 + H1;
  const float* b2 = W2 + H2 * H1;
  const float* W3 = b2 + H2;
  const float* b3 = W3 + (1 + D) * H2;
  bf16_t* w2h = (bf16_t*)out;
  bf16_t* w3h = (bf16_t*)(out + W2_BYTES);
  bf16_t* w2l = (bf16_t*)(out + W2_BYTES + K::W3_BYTES);
  bf16_t* w3l = (bf16_t*)(out + 2 * W2_BYTES + K::W3_BYTES);
  float* bo = (float*)(out + 2 * (W2_BYTES + K::W3_BYTES));
  const int gsz = gridDim.x * blockDim.x, g0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = g0; i < W2_BYTES / 2; i += gsz) {
    const int e = i & 7, m = (i >> 3) & 31, hh = (i >> 8) & 1, s = i >> 9;
    const int c = 32 * (s >> 1) + 16 * (s & 1) + 8 * (e >> 2) + 4 * hh + (e & 3);
    const float v = W2[m * H1 + c];
    const bf16_t hi = f32_to_bf16(v);
    w2h[i] = hi;
    w2l[i] = f32_to_bf16(v - bf16_to_f32(hi));
  }
  for (int i = g0; i < K::W3_BYTES / 2; i += gsz) {
    const int e = i & 7, m = (i >> 3) & 31, hh = (i >> 8) & 1, u = (i >> 9) & 1, t = i >> 10;
    const int r = 16 * u + 8 * (e >> 2) + 4 * hh + (e & 3);
    const int row = t < K::NT ? (32 * t + m < D ? 1 + 32 * t + m : -1) : (m == 0 ? 0 : -1);
    const float v = row < 0 ? 0.f : W3[row * H2 + r];
    const bf16_t hi = f32_to_bf16(v);
    w3h[i] = hi;
    w3l[i] = f32_to_bf16(v - bf16_to_f32(hi));
  }
  for (int i = g0; i < K::NBIAS; i += gsz) {
    float v;
    if (i < H1) v = b1[i];
    else if (i < H1 + H2) v = b2[i - H1];
    else if (i < H1 + H2 + 32 * K::NT) v = (i - H1 - H2 < D) ? b3[1 + i - H1 - H2] : 0.f;
    else v = (i == H1 + H2 + 32 * K::NT) ? b3[0] : 0.f;
    bo[i] = v;
  }
}

// zf [rows][256] fp32 (layer-1 pre-activations) and tokens [rows][ldt] fp32 -> hi / lo rows [ Z | x ] of 640 bf16
template <int D>
__global__ void pixel_split_rows_kernel(const float* __restrict__ zf, const float* __restrict__ tok, int ldt,
                                        bf16_t* __restrict__ zxh, bf16_t* __restrict__ zxl, long long rows) {
  using K = Cfg<D>;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K::NCH) return;
  const long long r = i / K::NCH;
  const int c = (int)(i - r * K::NCH);
  const float v = c < H1 ? zf[r * H1 + c] : (c - H1 < D ? tok[r * ldt + (c - H1)] : 0.f);
  const bf16_t hi = f32_to_bf16(v);
  zxh[i] = hi;
  zxl[i] = f32_to_bf16(v - bf16_to_f32(hi));
}

int pix_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int D>
int pix_infer(const void* packed, void* zx, int ldzx, int B, int G, int out_h, int out_w, float mean, float std,
              float std_factor, const float* conf_state, float* trav, float* conf, float* loss, hipStream_t st) {
  using K = Cfg<D>;
  if (ldzx < K::ZXC || (ldzx % 8) || ((uintptr_t)zx & 15) || ((uintptr_t)packed & 15)) return WVN_ERR_ARG;
  const float sy = (float)(G - 1) / (float)(out_h - 1), sx = (float)(G - 1) / (float)(out_w - 1);
  // a 16-pixel span must stay inside 3 consecutive source cells (4 tokens): 15 * scale < 2
  if (15.f * sy > 1.99f || 15.f * sx > 1.99f) return WVN_ERR_ARG;
  // Z = x * W1^T, bf16, into columns [0,256) of the same rows (K = the zero-padded x width)
  GemmBf16Params g{};
  g.A = (const bf16_t*)zx + H1; g.lda = ldzx;
  g.W = (const bf16_t*)packed; g.ldw = K::DX;
  g.bias = nullptr;
  g.C = zx; g.ldc = ldzx;
  g.M = B * G * G; g.N = H1; g.K = K::DX;
  int rc = wvn_gemm_bf16_launch(g, EPI_BF16, st);
  if (rc != WVN_OK) return rc;

  PixParams p{};
  p.zx = (const bf16_t*)zx; p.ldzx = ldzx;
  p.wimg = (const unsigned char*)packed + K::W1_BYTES;
  p.trav = trav; p.conf = conf; p.loss = loss;
  p.B = B; p.G = G; p.Ho = out_h; p.Wo = out_w;
  p.nty = ceil_div(out_h, TILE); p.ntx = ceil_div(out_w, TILE);
  p.sy = sy; p.sx = sx; p.mean = mean; p.std = std; p.std_factor = std_factor; p.conf_dev = conf_state;
  auto kern = pixel_mlp_kernel<1, D>;  // bilinear weights split hi + lo (16 mantissa bits)
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(K::LDS_BYTES, (const void*)kern)) return rc;
  const int ntiles = B * p.nty * p.ntx;
  const int cap = 2 * pix_num_cus();
  hipLaunchKernelGGL(kern, dim3(ntiles < cap ? ntiles : cap), dim3(512), K::LDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

template <int D>
size_t pix_exact_ws(int B, int G) {
  const size_t rows = (size_t)B * G * G;
  return rows * H1 * 4 + 2 * rows * Cfg<D>::NCH * 2 + 256;
}

template <int D>
int pix_infer_exact(const float* params, const void* packed, const float* tokens, int ldt, int B, int G, int out_h, int out_w,
                    float mean, float std, float std_factor, const float* conf_state, float* trav, float* conf, float* loss,
                    void* workspace, size_t workspace_bytes, hipStream_t st) {
  using K = Cfg<D>;
  if (ldt < D || ((uintptr_t)workspace & 15) || ((uintptr_t)packed & 15)) return WVN_ERR_ARG;
  if (workspace_bytes < pix_exact_ws<D>(B, G)) return WVN_ERR_WORKSPACE;
  const float sy = (float)(G - 1) / (float)(out_h - 1), sx = (float)(G - 1) / (float)(out_w - 1);
  if (15.f * sy > 1.99f || 15.f * sx > 1.99f) return WVN_ERR_ARG;
  const long long rows = (long long)B * G * G;
  float* zf = (float*)workspace;
  bf16_t* zxh = (bf16_t*)(zf + rows * H1);
  bf16_t* zxl = zxh + rows * K::NCH;
  // layer 1 at token resolution on the exact fp32 path: Z = x * W1^T (bias is added after the interpolation)
  GemmF32Params g{};
  g.A = tokens; g.lda = ldt; g.transA = 0;
  g.B = params; g.ldb = D; g.transB = 1;   // W1 [256][D], Linear layout, first in the flat parameter buffer
  g.bias = nullptr;
  g.C = zf; g.ldc = H1; g.M = (int)rows; g.N = H1; g.K = D; g.batch = 1; g.splitk = 1;
  int rc = wvn_gemm_f32_launch(g, F32_EPI_NONE, st);
  if (rc != WVN_OK) return rc;
  const long long nel = rows * K::NCH;
  hipLaunchKernelGGL(pixel_split_rows_kernel<D>, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, zf, tokens, ldt, zxh, zxl, rows);
  WVN_LAUNCH_CHECK();

  PixX3Params p{};
  p.zxh = zxh; p.zxl = zxl; p.ldzx = K::NCH;
  p.wimg = (const unsigned char*)packed;
  p.trav = trav; p.conf = conf; p.loss = loss;
  p.B = B; p.G = G; p.Ho = out_h; p.Wo = out_w;
  p.nty = ceil_div(out_h, TILE); p.ntx = ceil_div(out_w, TILE);
  p.sy = sy; p.sx = sx; p.mean = mean; p.std = std; p.std_factor = std_factor; p.conf_dev = conf_state;
  static LdsOptIn lds_opt_in;   // per device (common.h)
  if (const int rc = lds_opt_in(K::XLDS_BYTES, (const void*)pixel_mlp_x3_kernel<D>)) return rc;
  const int ntiles = B * p.nty * p.ntx;
  const int cap = pix_num_cus();
  hipLaunchKernelGGL(pixel_mlp_x3_kernel<D>, dim3(ntiles < cap ? ntiles : cap), dim3(512), K::XLDS_BYTES, st, p);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

bool pix_supported(int D, int h1, int h2) { return (D == 384 || D == 90) && h1 == H1 && h2 == H2; }

}  // namespace

// D = 384 (DINO ViT-S features) or 90 (STEGO code); 0 / WVN_ERR_ARG for anything else
size_t wvn_pixel_mlp_pack_bytes_impl(int D) {
  return D == 384 ? (size_t)Cfg<384>::W1_BYTES + Cfg<384>::WIMG_BYTES : D == 90 ? (size_t)Cfg<90>::W1_BYTES + Cfg<90>::WIMG_BYTES : 0;
}
int wvn_pixel_mlp_zx_cols_impl(int D) { return D == 384 ? Cfg<384>::ZXC : D == 90 ? Cfg<90>::ZXC : 0; }

int wvn_pixel_mlp_pack_launch(int D, int h1, int h2, const float* params, void* packed, hipStream_t st) {
  if (!pix_supported(D, h1, h2) || !params || !packed || ((uintptr_t)packed & 15)) return WVN_ERR_ARG;
  if (D == 384) hipLaunchKernelGGL(pixel_mlp_pack_kernel<384>, dim3(96), dim3(256), 0, st, params, (unsigned char*)packed);
  else hipLaunchKernelGGL(pixel_mlp_pack_kernel<90>, dim3(96), dim3(256), 0, st, params, (unsigned char*)packed);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_pixel_mlp_infer_launch(int D, int h1, int h2, const void* packed, void* zx, int ldzx, int B, int G, int out_h,
                               int out_w, float mean, float std, float std_factor, const float* conf_state, float* trav,
                               float* conf, float* loss, hipStream_t st) {
  if (!pix_supported(D, h1, h2) || !packed || !zx || B <= 0 || G < 2 || out_h < 2 || out_w < 2) return WVN_ERR_ARG;
  return D == 384 ? pix_infer<384>(packed, zx, ldzx, B, G, out_h, out_w, mean, std, std_factor, conf_state, trav, conf, loss, st)
                  : pix_infer<90>(packed, zx, ldzx, B, G, out_h, out_w, mean, std, std_factor, conf_state, trav, conf, loss, st);
}

size_t wvn_pixel_mlp_exact_pack_bytes_impl(int D) {
  return D == 384 ? (size_t)Cfg<384>::XWIMG_BYTES : D == 90 ? (size_t)Cfg<90>::XWIMG_BYTES : 0;
}
size_t wvn_pixel_mlp_exact_workspace_bytes_impl(int D, int B, int G) {
  return D == 384 ? pix_exact_ws<384>(B, G) : D == 90 ? pix_exact_ws<90>(B, G) : 0;
}

int wvn_pixel_mlp_exact_pack_launch(int D, int h1, int h2, const float* params, void* packed, hipStream_t st) {
  if (!pix_supported(D, h1, h2) || !params || !packed || ((uintptr_t)packed & 15)) return WVN_ERR_ARG;
  if (D == 384) hipLaunchKernelGGL(pixel_mlp_pack_x3_kernel<384>, dim3(96), dim3(256), 0, st, params, (unsigned char*)packed);
  else hipLaunchKernelGGL(pixel_mlp_pack_x3_kernel<90>, dim3(96), dim3(256), 0, st, params, (unsigned char*)packed);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_pixel_mlp_infer_exact_launch(int D, int h1, int h2, const float* params, const void* packed, const float* tokens,
                                     int ldt, int B, int G, int out_h, int out_w, float mean, float std, float std_factor,
                                     const float* conf_state, float* trav, float* conf, float* loss, void* workspace,
                                     size_t workspace_bytes, hipStream_t st) {
  if (!pix_supported(D, h1, h2) || !params || !packed || !tokens || !workspace || B <= 0 || G < 2 || out_h < 2 || out_w < 2)
    return WVN_ERR_ARG;
  return D == 384 ? pix_infer_exact<384>(params, packed, tokens, ldt, B, G, out_h, out_w, mean, std, std_factor, conf_state, trav,
                                         conf, loss, workspace, workspace_bytes, st)
                  : pix_infer_exact<90>(params, packed, tokens, ldt, B, G, out_h, out_w, mean, std, std_factor, conf_state, trav,
                                        conf, loss, workspace, workspace_bytes, st);
}
