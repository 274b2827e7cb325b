// Segment kernels (HBM-bound integer / scatter work; wave-level reductions, no GEMM reshaping).  Every float result here is
// deterministic: sums either run in a fixed order or accumulate in 64-bit fixed point (integer atomics are order-independent).
//   segpool_weights + segpool_reduce : FeatureExtractor.sparsify_features fused with the bilinear
//        up-sampling of DinoInterface.inference -- the [B,D,H,H] dense map (308 MB/frame) is never
//        materialised.  mean_{pixels in s} bilinear(F)(pixel) = sum_p W[s,p] F[p] / |s| with
//        W[s,p] = sum of the bilinear tap weights that pixels of s put on patch p.
//   label_pool      : MissionNode.update_supervision_signal (nodes.py:400-440)
//   centers         : SegmentExtractor.centers (segment_extractor.py:70-92), exact integer sums
//   adjacency       : SegmentExtractor.adjacency_list (segment_extractor.py:39-67), bit-exact
#include "common.h"
#include "wvn_internal.h"

namespace {

__device__ inline float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// Weights accumulate as 2^-40 fixed point in 64-bit integers: integer addition is associative, so the result does not
// depend on the order in which the atomics land (fp32 atomics made the general pooling run-to-run different in the last
// bits).  Tap weights are in [0, 64] per run and a segment has < 2^18 pixels: no overflow; quantisation 1e-12.
constexpr double WFIX = 1099511627776.0;  // 2^40
__device__ inline void wfix_add(unsigned long long* p, float v) {
  atomicAdd(p, (unsigned long long)__double2ll_rn((double)v * WFIX));
}

// One lane per pixel, a wave covers 64 consecutive pixels of the flattened frame.  Runs of lanes
// with the same (segment, row, left tap) are reduced in-wave (prefix-sum differences, fixed lane order) so that one
// lane per run issues the 4 weight atomics + 1 count atomic (~8x fewer L2 atomics at P=8).
__global__ __launch_bounds__(256) void segpool_weights_kernel(const int* __restrict__ seg, unsigned long long* __restrict__ W,
                                                              int* __restrict__ cnt, int H, int Wd, int G, int S) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int npix = H * Wd;
  const bool inb = pix < npix;
  int s = -1, y = 0, x = 0;
  if (inb) {
    s = seg[(size_t)b * npix + pix];
    y = pix / Wd;
    x = pix - y * Wd;
  }
  if (s >= S) s = -1;
  // align_corners=True taps; the reference resizes to (H, H) using H for both dims (dino_interface.py:88)
  const float scale = (H > 1) ? (float)(G - 1) / (float)(H - 1) : 0.f;
  const float sx = scale * (float)x, sy = scale * (float)y;
  const int x0 = (int)sx, y0 = (int)sy;
  const int x1 = x0 + (x0 < G - 1 ? 1 : 0), y1 = y0 + (y0 < G - 1 ? 1 : 0);
  const float wx1 = sx - (float)x0, wx0 = 1.f - wx1;
  const float wy1 = sy - (float)y0, wy0 = 1.f - wy1;

  const long long key = (s < 0) ? -1ll : (((long long)s * H + y) * G + x0);
  const long long prev = __shfl_up(key, 1, 64);
  const bool head = (lane == 0) || (key != prev);
  const unsigned long long heads = __ballot(head);
  const float p0 = wave_incl_scan(s < 0 ? 0.f : wx0, lane);
  const float p1 = wave_incl_scan(s < 0 ? 0.f : wx1, lane);
  // end of my run = lane before the next head (or 63)
  const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
  const int last = above ? (lane + __builtin_ctzll(above)) : 63;
  const float e0 = __shfl(p0, last, 64), e1 = __shfl(p1, last, 64);
  const float b0 = __shfl_up(p0, 1, 64), b1 = __shfl_up(p1, 1, 64);
  if (head && s >= 0) {
    const float r0 = e0 - (lane ? b0 : 0.f), r1 = e1 - (lane ? b1 : 0.f);
    unsigned long long* Wr = W + ((size_t)b * S + s) * (size_t)(G * G);
    wfix_add(Wr + y0 * G + x0, wy0 * r0);
    if (r1 != 0.f) wfix_add(Wr + y0 * G + x1, wy0 * r1);
    if (wy1 != 0.f) {
      wfix_add(Wr + y1 * G + x0, wy1 * r0);
      if (r1 != 0.f) wfix_add(Wr + y1 * G + x1, wy1 * r1);
    }
    atomicAdd(cnt + (size_t)b * S + s, last - lane + 1);
  }
}

// The same table for FEW segments (2 S G 64-bit words fit in LDS: the k-means maps of the STEGO stage, S = 20): one workgroup per
// (band of image rows that share their upper code row, frame).  Every pixel adds its four tap weights -- each quantised on its own, so
// the integer sums do not depend on any grouping -- to an LDS table [2 code rows][S][G], and only the table's non-zero words go to
// memory: a patch belongs to one or two segments, i.e. ~10 K global atomics per frame instead of the ~125 K run heads of the
// per-pixel kernel above (2.1 ms of L2 atomics per 64 frames at 448 x 448).
__global__ __launch_bounds__(512) void segpool_weights_band_kernel(const int* __restrict__ seg, unsigned long long* __restrict__ W,
                                                                   int* __restrict__ cnt, int H, int Wd, int G, int S) {
  extern __shared__ __attribute__((aligned(8))) unsigned long long tab[];   // [2][S][G], then int c[S]
  int* csh = (int*)(tab + 2 * S * G);
  const int band = blockIdx.x, b = blockIdx.y;
  const float scale = (H > 1) ? (float)(G - 1) / (float)(H - 1) : 0.f;
  for (int i = threadIdx.x; i < 2 * S * G; i += blockDim.x) tab[i] = 0ull;
  for (int i = threadIdx.x; i < S; i += blockDim.x) csh[i] = 0;
  // image rows of the band: (int)(scale * y) == band  (the expression of the per-pixel kernel; monotonic in y)
  int ya = scale > 0.f ? max(0, (int)((float)band / scale) - 2) : 0;
  while (ya < H && (int)(scale * (float)ya) < band) ++ya;
  int yb = ya;
  while (yb < H && (int)(scale * (float)yb) == band) ++yb;
  __syncthreads();
  const int npix = H * Wd;
  for (int i = threadIdx.x; i < (yb - ya) * Wd; i += blockDim.x) {
    const int y = ya + i / Wd, x = i - (i / Wd) * Wd;
    int s = seg[(size_t)b * npix + (size_t)y * Wd + x];
    if (s < 0 || s >= S) continue;
    const float sx = scale * (float)x, sy = scale * (float)y;
    const int x0 = (int)sx, y0 = (int)sy;
    const int x1 = x0 + (x0 < G - 1 ? 1 : 0);
    const float wx1 = sx - (float)x0, wx0 = 1.f - wx1;
    const float wy1 = sy - (float)y0, wy0 = 1.f - wy1;
    unsigned long long* t0 = tab + (size_t)s * G;
    unsigned long long* t1 = tab + (size_t)(S + s) * G;
    atomicAdd(t0 + x0, (unsigned long long)__double2ll_rn((double)(wy0 * wx0) * WFIX));
    if (wx1 != 0.f) atomicAdd(t0 + x1, (unsigned long long)__double2ll_rn((double)(wy0 * wx1) * WFIX));
    if (wy1 != 0.f) {
      atomicAdd(t1 + x0, (unsigned long long)__double2ll_rn((double)(wy1 * wx0) * WFIX));
      if (wx1 != 0.f) atomicAdd(t1 + x1, (unsigned long long)__double2ll_rn((double)(wy1 * wx1) * WFIX));
    }
    atomicAdd(csh + s, 1);
  }
  __syncthreads();
  const int y1 = band + (band < G - 1 ? 1 : 0);
  for (int i = threadIdx.x; i < 2 * S * G; i += blockDim.x) {
    const unsigned long long v = tab[i];
    if (v == 0ull) continue;
    const int r = i / (S * G), j = i - r * (S * G), s = j / G, x = j - s * G;
    atomicAdd(W + ((size_t)b * S + s) * (size_t)(G * G) + (size_t)(r ? y1 : band) * G + x, v);
  }
  for (int i = threadIdx.x; i < S; i += blockDim.x)
    if (csh[i]) atomicAdd(cnt + (size_t)b * S + i, csh[i]);
}

// feat[b][s][:] = (sum_p W[b][s][p] * F[b][p][:]) / cnt[b][s]   (0/0 -> NaN like the reference's empty mean)
// One workgroup per (s, b); thread = channel.  W rows are sparse: chunks are staged in LDS and
// zero entries skipped (wave-uniform branch).  Accumulation order is ascending p: deterministic.
__global__ void segpool_reduce_kernel(const unsigned long long* __restrict__ W, const int* __restrict__ cnt,
                                      const float* __restrict__ F, int ldf, float* __restrict__ feat, int P, int S,
                                      int D) {
  __shared__ float wch[256];
  __shared__ unsigned long long nz[4];   // per 64 staged weights: which are non-zero (a segment touches ~1 patch in 15)
  const int s = blockIdx.x, b = blockIdx.y;
  const int d = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
  const unsigned long long* Wr = W + ((size_t)b * S + s) * P;
  const float* Fb = F + (size_t)b * P * ldf;
  float acc = 0.f;
  for (int p0 = 0; p0 < P; p0 += 256) {
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) wch[i] = (p0 + i < P) ? (float)((double)Wr[p0 + i] * (1.0 / WFIX)) : 0.f;
    __syncthreads();
    for (int sc = wave; sc < 4; sc += nwave) {
      const unsigned long long m = __ballot(wch[sc * 64 + lane] != 0.f);
      if (lane == 0) nz[sc] = m;
    }
    __syncthreads();
#pragma unroll
    for (int sc = 0; sc < 4; ++sc) {
      unsigned long long m = nz[sc];                       // (uniform) ascending p: the accumulation order is fixed
      while (m) {
        const int i = sc * 64 + __builtin_ctzll(m);
        m &= m - 1;
        if (d < D) acc = fmaf(wch[i], Fb[(size_t)(p0 + i) * ldf + d], acc);
      }
    }
  }
  if (d < D) feat[((size_t)b * S + s) * D + d] = acc / (float)cnt[(size_t)b * S + s];
}

// ---- label pooling --------------------------------------------------------------------------
// Per-pixel label = nanmean over the mask channels; per-segment sum in 2^-32 fixed point (64-bit integer atomics:
// order-independent, deterministic; labels are traversability scores in [0, 1]).  Batched over nodes through pointer
// tables so that add_supervision_node re-pools every mission node in range with ONE launch pair.
constexpr double LFIX = 4294967296.0;  // 2^32
struct LabelPoolNode {
  const float* mask;        // [C][H][W]
  const int* seg;           // [H][W]
  float* signal;            // [S]
  unsigned char* valid;     // [S]
  int S;
  int pad;
};
__device__ inline void lp_accum(const LabelPoolNode& nd, int C, long long* sum, int* cnt, int npix) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float t = 0.f;
  int c = 0;
  for (int ch = 0; ch < C; ++ch) {
    float v = nd.mask[(size_t)ch * npix + i];
    if (!isnan(v)) { t += v; ++c; }
  }
  int s = nd.seg[i];
  if (c == 0 || s < 0 || s >= nd.S) return;
  atomicAdd((unsigned long long*)(sum + s), (unsigned long long)__double2ll_rn((double)(t / (float)c) * LFIX));
  atomicAdd(cnt + s, 1);
}
__device__ inline void lp_final(const LabelPoolNode& nd, const long long* sum, const int* cnt) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nd.S) return;
  const int n = cnt[s];
  float m = (n > 0) ? (float)((double)sum[s] * (1.0 / LFIX) / (double)n) : 0.f;  // nan_to_num(0/0) = 0
  if (isnan(m)) m = 0.f;
  nd.signal[s] = m;
  nd.valid[s] = m > 0.f;
}
__global__ void label_pool_accum_batched_kernel(const LabelPoolNode* __restrict__ nodes, int C, long long* __restrict__ sum,
                                                int* __restrict__ cnt, int npix, int Smax) {
  lp_accum(nodes[blockIdx.y], C, sum + (size_t)blockIdx.y * Smax, cnt + (size_t)blockIdx.y * Smax, npix);
}
__global__ void label_pool_final_batched_kernel(const LabelPoolNode* __restrict__ nodes, const long long* __restrict__ sum,
                                                const int* __restrict__ cnt, int Smax) {
  lp_final(nodes[blockIdx.y], sum + (size_t)blockIdx.y * Smax, cnt + (size_t)blockIdx.y * Smax);
}
__global__ void label_pool_accum_kernel(LabelPoolNode nd, int C, long long* __restrict__ sum, int* __restrict__ cnt, int npix) {
  lp_accum(nd, C, sum, cnt, npix);
}
__global__ void label_pool_final_kernel(LabelPoolNode nd, const long long* __restrict__ sum, const int* __restrict__ cnt) {
  lp_final(nd, sum, cnt);
}

// ---- centers ----------------------------------------------------------------------------------
__global__ void centers_accum_kernel(const int* __restrict__ seg, unsigned long long* __restrict__ acc, int H, int Wd,
                                     int S) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * Wd) return;
  int s = seg[i];
  if (s < 0 || s >= S) return;
  int y = i / Wd, x = i - y * Wd;
  atomicAdd(acc + 3 * s + 0, (unsigned long long)x);
  atomicAdd(acc + 3 * s + 1, (unsigned long long)y);
  atomicAdd(acc + 3 * s + 2, 1ull);
}
__global__ void centers_final_kernel(const unsigned long long* __restrict__ acc, float* __restrict__ out, int S) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  double n = (double)acc[3 * s + 2];
  out[2 * s + 0] = (float)((double)acc[3 * s + 0] / n);  // (x, y): the reference transposes before nonzero()
  out[2 * s + 1] = (float)((double)acc[3 * s + 1] / n);
}

// ---- adjacency --------------------------------------------------------------------------------
// pair (left=s[y,x], right=s[y,x+1]) and (s[y,x], s[y+1,x]) wherever they differ; key = left + right*S
__global__ void adjacency_mark_kernel(const int* __restrict__ seg, unsigned char* __restrict__ bitmap, int H, int Wd,
                                      int S) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * Wd) return;
  int y = i / Wd, x = i - y * Wd;
  int s = seg[i];
  if (x + 1 < Wd) {
    int r = seg[i + 1];
    if (r != s && r >= 0 && s >= 0 && r < S && s < S) bitmap[(size_t)r * S + s] = 1;
  }
  if (y + 1 < H) {
    int d = seg[i + Wd];
    if (d != s && d >= 0 && s >= 0 && d < S && s < S) bitmap[(size_t)d * S + s] = 1;
  }
}
// single workgroup: ordered compaction of the S*S bitmap (ascending key) -> edges[E][2] (int64), *count = E
__global__ __launch_bounds__(1024) void adjacency_compact_kernel(const unsigned char* __restrict__ bitmap,
                                                                 long long* __restrict__ edges, int* __restrict__ count,
                                                                 int S, int max_edges) {
  __shared__ int part[1024];
  const int n = S * S;
  const int per = (n + 1023) / 1024;
  const int beg = threadIdx.x * per, end = min(n, beg + per);
  int c = 0;
  for (int i = beg; i < end; ++i) c += bitmap[i];
  part[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 1024; ++i) { int t = part[i]; part[i] = run; run += t; }
    *count = run;
  }
  __syncthreads();
  int o = part[threadIdx.x];
  for (int i = beg; i < end; ++i)
    if (bitmap[i]) {
      if (o < max_edges) { edges[2 * o + 0] = i % S; edges[2 * o + 1] = i / S; }
      ++o;
    }
}


// ---- plain per-segment mean of an explicit pixel-resolution map (sparsify_features on a dense tensor) ----
// tokens [B,P,D] (pixel-major), seg [B,P].  Deterministic, no atomics: ONE wave per (pixel chunk, 64-channel slab, frame)
// walks its chunk in ascending pixel order (lane = channel) into an LDS table [S][64], writes the table as a partial to
// scratch; a second kernel adds the chunk partials in ascending chunk order and divides by the pixel count.
constexpr int SM_PIX = 2048;
__global__ __launch_bounds__(64) void segmean_partial_kernel(const int* __restrict__ seg, const float* __restrict__ tok,
                                                             float* __restrict__ part_out, int* __restrict__ cnt_out, int P,
                                                             int S, int D, int nchunk) {
  extern __shared__ float part[];  // [S][64] + int cnt[S]
  int* pc = (int*)(part + (size_t)S * 64);
  const int b = blockIdx.z, c0 = blockIdx.y * 64, chunk = blockIdx.x, p0 = chunk * SM_PIX;
  const int ch = threadIdx.x;
  for (int i = ch; i < S * 64; i += 64) part[i] = 0.f;
  for (int i = ch; i < S; i += 64) pc[i] = 0;
  const int pend = min(P, p0 + SM_PIX);
  const bool live = c0 + ch < D;
  for (int p = p0; p < pend; ++p) {
    const int s = seg[(size_t)b * P + p];   // wave-uniform
    if (s < 0 || s >= S) continue;
    if (live) part[s * 64 + ch] += tok[((size_t)b * P + p) * D + c0 + ch];
    if (ch == 0) pc[s] += 1;
  }
  // (single wave: its own LDS writes are visible to its later reads in program order)
  float* dst = part_out + (((size_t)b * nchunk + chunk) * S) * D + c0;
  for (int s = 0; s < S; ++s)
    if (live) dst[(size_t)s * D + ch] = part[s * 64 + ch];
  if (blockIdx.y == 0)
    for (int i = ch; i < S; i += 64) cnt_out[((size_t)b * nchunk + chunk) * S + i] = pc[i];
}
__global__ void segmean_final_kernel(const float* __restrict__ part, const int* __restrict__ pcnt, float* __restrict__ out,
                                     int* __restrict__ cnt, int B, int S, int D, int nchunk) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * S * D) return;
  const int d = (int)(i % D);
  const int s = (int)((i / D) % S);
  const int b = (int)(i / ((long long)D * S));
  float t = 0.f;
  int n = 0;
  for (int c = 0; c < nchunk; ++c) {   // ascending chunk order
    t += part[(((size_t)b * nchunk + c) * S + s) * D + d];
    n += pcnt[((size_t)b * nchunk + c) * S + s];
  }
  out[i] = t / (float)n;   // 0/0 = NaN for an id without pixels, like the reference's empty mean
  if (d == 0) cnt[(size_t)b * S + s] = n;
}

// ---- fused up-sample + segment mean for PATCH-ALIGNED label maps (k-means clusters, grid cells) ---------
// When the segment map is a G x G label grid nearest-upsampled by exactly the patch size (H == G*P), all
// P*P pixels of patch q share one label, and the mean over a segment's pixels of the bilinearly up-sampled
// features equals the mean over the segment's PATCHES of a fixed, separable 3x3 stencil of the patch map:
//     Fs[q] = sum_{a,b in {-1,0,1}} wy[gy][a] wx[gx][b] F[gy+a][gx+b],   pooled[s] = mean_{q in s} Fs[q]
// (wy/wx = per-patch-row sums of the align_corners tap weights / P, built by the host once per geometry).
// No atomics, deterministic: lane = channel, each wave walks its quarter of the patches in ascending
// order into a private LDS table sums[label][channel]; the tables are combined in wave order.
__global__ __launch_bounds__(256) void segpool_patch_kernel(const int* __restrict__ labels, const float* __restrict__ tok,
                                                            int ldf, const float* __restrict__ wy,
                                                            const float* __restrict__ wx, float* __restrict__ feat,
                                                            int G, int S, int D) {
  // One workgroup per (64-channel slab, segment id, frame).  Each of the 4 waves owns a contiguous quarter of
  // the frame's patches and walks it in ascending order: 64 labels per coalesced load, a ballot picks the
  // patches of this segment, and every lane (= channel) adds their 3x3 stencil values to its running sum.
  // The four partial sums are combined in wave order, so the result does not depend on scheduling.
  __shared__ float part[4][64];
  __shared__ int cnts[4];
  __shared__ unsigned short hits[4][1024];  // per wave: its quarter's patches of this segment, ascending (P / 4 <= 1024)
  __shared__ float wys[192], wxs[192];       // stencil weights (G <= 64)
  for (int i = threadIdx.x; i < G * 3; i += 256) { wys[i] = wy[i]; wxs[i] = wx[i]; }
  __syncthreads();
  const int b = blockIdx.z, s = blockIdx.y, c0 = blockIdx.x * 64;
  const int ch = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int P = G * G;
  const int per = (P + 3) / 4;
  const int q0 = w * per, q1 = min(P, q0 + per);
  const bool live = (c0 + ch) < D;
  const float* F = tok + (size_t)b * P * ldf + c0 + ch;
  const int* lab = labels + (size_t)b * P;
  // pass 1: compact the matching patch indices (ballot + prefix popcount keeps them in ascending order)
  int n = 0;
  for (int base = q0; base < q1; base += 64) {
    const int q = base + ch;
    const bool hit = q < q1 && lab[q] == s;
    const unsigned long long m = __ballot(hit);
    if (hit) hits[w][n + __popcll(m & ((1ull << ch) - 1ull))] = (unsigned short)(q - q0);
    n += __popcll(m);
  }
  // (wave-private list: the wave's own LDS writes are visible to its later reads in program order)
  // pass 2: stencil values of four patches are fetched together (36 independent loads in flight), then added to
  // the running sum one patch at a time, in ascending patch order -- the summation order is unchanged
  // Branch-free stencil: taps that fall outside the map (or carry weight 0) are read at a clamped index and multiplied
  // by their zero weight -- fmaf(0, x, r) == r for finite x, so the value equals the skip-the-tap form bit for bit while
  // all nine loads of a patch (and of its three batch mates) are independent and issue back to back.
  auto stencil = [&](int qq) -> float {
    const int gy = qq / G, gx = qq - gy * G;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = gy + a - 1;
      const bool oky = yy >= 0 && yy < G;
      const float wa = oky ? wys[gy * 3 + a] : 0.f;
      const int yc = min(max(yy, 0), G - 1);
      float row = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int xx = gx + c - 1;
        const bool okx = xx >= 0 && xx < G;
        const float wc = okx ? wxs[gx * 3 + c] : 0.f;
        const int xc = min(max(xx, 0), G - 1);
        row = fmaf(wc, F[(size_t)(yc * G + xc) * ldf], row);
      }
      acc = fmaf(wa, row, acc);
    }
    return acc;
  };
  float sum = 0.f;
  if (live) {
    int i = 0;
    for (; i + 4 <= n; i += 4) {
      const float a0 = stencil(q0 + hits[w][i]), a1 = stencil(q0 + hits[w][i + 1]);
      const float a2 = stencil(q0 + hits[w][i + 2]), a3 = stencil(q0 + hits[w][i + 3]);
      sum += a0; sum += a1; sum += a2; sum += a3;
    }
    for (; i < n; ++i) sum += stencil(q0 + hits[w][i]);
  }
  part[w][ch] = sum;
  if (ch == 0) cnts[w] = n;
  __syncthreads();
  if (w == 0 && live) {
    float t = part[0][ch];
    int nn = cnts[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) { t += part[k][ch]; nn += cnts[k]; }
    feat[((size_t)b * S + s) * D + c0 + ch] = t / (float)nn;  // 0/0 = NaN for an id without patches
  }
}

}  // namespace

int wvn_segpool_patch_launch(const int* labels, const float* tok, int ldf, const float* wy, const float* wx,
                             float* feat, int B, int G, int S, int D, hipStream_t st) {
  if (!labels || !tok || !wy || !wx || !feat || S <= 0 || D <= 0 || S > 65535 || B > 65535 || G > 64) return WVN_ERR_ARG;
  hipLaunchKernelGGL(segpool_patch_kernel, dim3(ceil_div(D, 64), S, B), dim3(256), 0, st, labels, tok, ldf, wy, wx, feat,
                     G, S, D);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_segpool_launch(const int* seg, const float* tok, int ldf, float* feat, void* Wv, int* cnt, int B, int H,
                       int Wd, int G, int S, int D, hipStream_t st) {
  unsigned long long* W = (unsigned long long*)Wv;
  if (!seg || !tok || !feat || !W || !cnt || S <= 0 || D <= 0 || D > 1024 || ((uintptr_t)W & 7)) return WVN_ERR_ARG;
  const int P = G * G;
  hipError_t e = hipMemsetAsync(W, 0, (size_t)B * S * P * sizeof(unsigned long long), st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(cnt, 0, (size_t)B * S * sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  const size_t band_lds = (size_t)2 * S * G * sizeof(unsigned long long) + (size_t)S * sizeof(int);
  if (band_lds <= 48 * 1024 && H == Wd)   // few segments (k-means maps): per-band LDS tables, ~12x fewer global atomics
    hipLaunchKernelGGL(segpool_weights_band_kernel, dim3(G, B), dim3(512), band_lds, st, seg, W, cnt, H, Wd, G, S);
  else
    hipLaunchKernelGGL(segpool_weights_kernel, dim3(ceil_div(H * Wd, 256), B), dim3(256), 0, st, seg, W, cnt, H, Wd, G, S);
  WVN_LAUNCH_CHECK();
  int threads = ((D + 63) / 64) * 64;
  hipLaunchKernelGGL(segpool_reduce_kernel, dim3(S, B), dim3(threads), 0, st, W, cnt, tok, ldf, feat, P, S, D);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// nodes: DEVICE array of n LabelPoolNode-shaped records (include/wvn_hip.h: wvn_label_pool_node); sum / cnt: n * Smax words
int wvn_label_pool_batched_launch(const void* nodes, int n, int C, int H, int Wd, int Smax, long long* sum, int* cnt,
                                  hipStream_t st) {
  if (!nodes || !sum || !cnt || n <= 0 || Smax <= 0 || C <= 0) return WVN_ERR_ARG;
  hipError_t e = hipMemsetAsync(sum, 0, (size_t)n * Smax * sizeof(long long), st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(cnt, 0, (size_t)n * Smax * sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(label_pool_accum_batched_kernel, dim3(ceil_div(H * Wd, 256), n), dim3(256), 0, st,
                     (const LabelPoolNode*)nodes, C, sum, cnt, H * Wd, Smax);
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(label_pool_final_batched_kernel, dim3(ceil_div(Smax, 256), n), dim3(256), 0, st,
                     (const LabelPoolNode*)nodes, sum, cnt, Smax);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

// single node; sum: S 8-byte words, cnt: S ints
int wvn_label_pool_launch(const float* mask, int C, const int* seg, float* signal, unsigned char* valid, void* sum,
                          int* cnt, int H, int Wd, int S, hipStream_t st) {
  if (!mask || !seg || !signal || !valid || !sum || !cnt || ((uintptr_t)sum & 7)) return WVN_ERR_ARG;
  hipError_t e = hipMemsetAsync(sum, 0, (size_t)S * sizeof(long long), st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(cnt, 0, S * sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  LabelPoolNode nd{mask, seg, signal, valid, S, 0};
  hipLaunchKernelGGL(label_pool_accum_kernel, dim3(ceil_div(H * Wd, 256)), dim3(256), 0, st, nd, C, (long long*)sum, cnt, H * Wd);
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(label_pool_final_kernel, dim3(ceil_div(S, 256)), dim3(256), 0, st, nd, (const long long*)sum, (const int*)cnt);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_centers_launch(const int* seg, float* centers, unsigned long long* scratch, int H, int Wd, int S,
                       hipStream_t st) {
  if (!seg || !centers || !scratch) return WVN_ERR_ARG;
  hipError_t e = hipMemsetAsync(scratch, 0, (size_t)3 * S * sizeof(unsigned long long), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(centers_accum_kernel, dim3(ceil_div(H * Wd, 256)), dim3(256), 0, st, seg, scratch, H, Wd, S);
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(centers_final_kernel, dim3(ceil_div(S, 256)), dim3(256), 0, st, scratch, centers, S);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_adjacency_launch(const int* seg, long long* edges, int* count, unsigned char* bitmap, int H, int Wd, int S,
                         int max_edges, hipStream_t st) {
  if (!seg || !edges || !count || !bitmap) return WVN_ERR_ARG;
  hipError_t e = hipMemsetAsync(bitmap, 0, (size_t)S * S, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(adjacency_mark_kernel, dim3(ceil_div(H * Wd, 256)), dim3(256), 0, st, seg, bitmap, H, Wd, S);
  WVN_LAUNCH_CHECK();
  hipLaunchKernelGGL(adjacency_compact_kernel, dim3(1), dim3(1024), 0, st, bitmap, edges, count, S, max_edges);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

size_t wvn_segmean_scratch_bytes_impl(int B, int P, int S, int D) {
  const size_t nchunk = (size_t)ceil_div(P, SM_PIX);
  return (size_t)B * nchunk * S * D * sizeof(float) + (size_t)B * nchunk * S * sizeof(int);
}

int wvn_segmean_tokens_launch(const int* seg, const float* tok, float* out, int* cnt, void* scratch, size_t scratch_bytes, int B,
                              int P, int S, int D, hipStream_t st) {
  if (!seg || !tok || !out || !cnt || !scratch || S <= 0 || (size_t)S * 65 * 4 > 60 * 1024) return WVN_ERR_ARG;
  if (scratch_bytes < wvn_segmean_scratch_bytes_impl(B, P, S, D)) return WVN_ERR_WORKSPACE;
  const int nchunk = ceil_div(P, SM_PIX);
  float* part = (float*)scratch;
  int* pcnt = (int*)(part + (size_t)B * nchunk * S * D);
  size_t shm = (size_t)S * 65 * 4;
  hipLaunchKernelGGL(segmean_partial_kernel, dim3(nchunk, ceil_div(D, 64), B), dim3(64), shm, st, seg, tok, part, pcnt, P, S,
                     D, nchunk);
  WVN_LAUNCH_CHECK();
  long long n = (long long)B * S * D;
  hipLaunchKernelGGL(segmean_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, pcnt, out, cnt, B, S, D,
                     nchunk);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
