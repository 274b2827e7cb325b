// SLIC superpixels on the GPU for FeatureExtractor(segmentation_type="slic") -- the reference's class default
// (feature_extractor.py:23,84-90,221-225), where it is a GPU -> CPU -> GPU round trip through the external C++ package
// fast_slic (Slic(num_components=100, compactness=10).iterate on a uint8 HWC frame).  fast_slic is absent from this image and
// from /root/reference: PARITY WITH fast_slic IS UNPINNED.  This is the published SLIC algorithm (Achanta et al. 2012:
// k-means in (L, a, b, x, y) with distance dc^2 + (m / S)^2 ds^2, grid-initialised centres, each pixel searching the clusters
// of its own and the eight neighbouring grid cells) made DETERMINISTIC AND BIT-REPRODUCIBLE by doing everything in integers:
//   * sRGB -> linear through a 256-entry table, linear RGB -> XYZ with 2^-14 fixed-point coefficients, XYZ -> f(t) through a
//     4096-entry table, Lab kept at 1/64 resolution (tables are supplied by the caller, built in double precision on the host);
//   * distances in 64-bit integers, ties to the lowest cluster id;
//   * centre updates from 64-bit integer sums (atomics; integer addition is order-independent), rounded integer division.
// oracle/slic.py restates the same integer arithmetic in numpy; labels must match bit for bit.  Like fast_slic's output the map
// holds ids in [0, number of clusters); connectivity enforcement (fast_slic's post-pass that re-assigns stray islands) is not
// performed -- ids without pixels give NaN feature rows exactly as an empty id does in the reference (feature_extractor.py:394).
#include "common.h"
#include "wvn_internal.h"

namespace {

struct SlicGeom { int H, W, gs, nx, ny, K; long long S2, m2q; };

__host__ __device__ inline long long floor_div(long long a, long long b) {  // b > 0
  long long q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

// img: CHW planar, uint8 or float in [0,1] (then truncated like the reference's np.uint8(img * 255))
template <typename T>
__global__ void slic_lab_kernel(const T* __restrict__ img, const int* __restrict__ lut_lin, const int* __restrict__ lut_f,
                                int* __restrict__ lab, int npix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  int u[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if constexpr (sizeof(T) == 1) u[c] = (int)img[(size_t)c * npix + i];
    else {
      const float v = (float)img[(size_t)c * npix + i] * 255.0f;
      u[c] = (int)(unsigned char)(int)v;   // np.uint8(x): truncation toward zero, wrap modulo 256
    }
  }
  const int R = lut_lin[u[0]], G = lut_lin[u[1]], B = lut_lin[u[2]];
  // coefficients round(c * 2^14) of the sRGB (D65) matrix, X and Z pre-divided by the white point (0.95047, 1.08883)
  int X = (7110 * R + 6164 * G + 3110 * B + 8192) >> 14;
  int Y = (3484 * R + 11717 * G + 1183 * B + 8192) >> 14;
  int Z = (291 * R + 1794 * G + 14300 * B + 8192) >> 14;
  X = min(max(X, 0), 4095); Y = min(max(Y, 0), 4095); Z = min(max(Z, 0), 4095);
  const int fx = lut_f[X], fy = lut_f[Y], fz = lut_f[Z];   // f(t) * 4096
  lab[i] = (int)floor_div(116ll * fy - 16 * 4096 + 32, 64);            // L * 64
  lab[npix + i] = (int)floor_div(500ll * (fx - fy) + 32, 64);          // a * 64
  lab[2 * npix + i] = (int)floor_div(200ll * (fy - fz) + 32, 64);      // b * 64
}

__global__ void slic_init_kernel(const int* __restrict__ lab, int* __restrict__ cent, SlicGeom g) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.K) return;
  const int i = k / g.nx, j = k - i * g.nx;
  const int cx = ((2 * j + 1) * g.W) / (2 * g.nx), cy = ((2 * i + 1) * g.H) / (2 * g.ny);
  const int p = cy * g.W + cx, npix = g.H * g.W;
  cent[5 * k + 0] = lab[p]; cent[5 * k + 1] = lab[npix + p]; cent[5 * k + 2] = lab[2 * npix + p];
  cent[5 * k + 3] = cx; cent[5 * k + 4] = cy;
}

__global__ __launch_bounds__(256) void slic_assign_kernel(const int* __restrict__ lab, const int* __restrict__ cent,
                                                          int* __restrict__ labels, long long* __restrict__ sums, SlicGeom g,
                                                          int accumulate) {
  extern __shared__ int cs[];  // [K][5]
  for (int i = threadIdx.x; i < 5 * g.K; i += blockDim.x) cs[i] = cent[i];
  __syncthreads();
  const int npix = g.H * g.W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inb = p < npix;
  int best = -1;
  int L = 0, A = 0, Bv = 0, x = 0, y = 0;
  if (inb) {
    y = p / g.W; x = p - y * g.W;
    L = lab[p]; A = lab[npix + p]; Bv = lab[2 * npix + p];
    const int cj = min(x * g.nx / g.W, g.nx - 1), ci = min(y * g.ny / g.H, g.ny - 1);
    long long bestd = 0x7fffffffffffffffll;
    for (int di = -1; di <= 1; ++di) {
      const int i = ci + di;
      if (i < 0 || i >= g.ny) continue;
      for (int dj = -1; dj <= 1; ++dj) {
        const int j = cj + dj;
        if (j < 0 || j >= g.nx) continue;
        const int k = i * g.nx + j;
        const long long dl = L - cs[5 * k], da = A - cs[5 * k + 1], db = Bv - cs[5 * k + 2];
        const long long dx = x - cs[5 * k + 3], dy = y - cs[5 * k + 4];
        const long long d = (dl * dl + da * da + db * db) * g.S2 + g.m2q * (dx * dx + dy * dy);
        if (d < bestd) { bestd = d; best = k; }   // ascending k: ties keep the lowest id
      }
    }
    labels[p] = best;
  }
  if (!accumulate) return;
  // centre sums: when the whole wave agrees on the cluster (the common case inside a superpixel) reduce in the wave first
  const int first = __builtin_amdgcn_readfirstlane(best);
  const bool uniform = __all(best == first) && first >= 0;
  if (uniform) {
    long long v[6] = {L, A, Bv, x, y, 1};
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      long long t = v[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if ((threadIdx.x & 63) == c) atomicAdd((unsigned long long*)(sums + 6 * first + c), (unsigned long long)t);
    }
  } else if (best >= 0) {
    atomicAdd((unsigned long long*)(sums + 6 * best + 0), (unsigned long long)(long long)L);
    atomicAdd((unsigned long long*)(sums + 6 * best + 1), (unsigned long long)(long long)A);
    atomicAdd((unsigned long long*)(sums + 6 * best + 2), (unsigned long long)(long long)Bv);
    atomicAdd((unsigned long long*)(sums + 6 * best + 3), (unsigned long long)(long long)x);
    atomicAdd((unsigned long long*)(sums + 6 * best + 4), (unsigned long long)(long long)y);
    atomicAdd((unsigned long long*)(sums + 6 * best + 5), 1ull);
  }
}

__global__ void slic_update_kernel(int* __restrict__ cent, long long* __restrict__ sums, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const long long n = sums[6 * k + 5];
  if (n > 0) {
#pragma unroll
    for (int c = 0; c < 5; ++c) cent[5 * k + c] = (int)floor_div(2 * sums[6 * k + c] + n, 2 * n);  // round half up
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) sums[6 * k + c] = 0;
}

SlicGeom slic_geom(int H, int W, int num_components, float compactness) {
  SlicGeom g;
  g.H = H; g.W = W;
  g.S2 = (long long)H * W / (num_components > 0 ? num_components : 1);
  if (g.S2 < 1) g.S2 = 1;
  int gs = 1;
  while ((long long)(gs + 1) * (gs + 1) <= g.S2) ++gs;   // isqrt
  g.gs = gs;
  g.nx = (W + gs / 2) / gs; if (g.nx < 1) g.nx = 1;
  g.ny = (H + gs / 2) / gs; if (g.ny < 1) g.ny = 1;
  g.K = g.nx * g.ny;
  g.m2q = (long long)lrintf(compactness * compactness * 4096.f);  // Lab is kept at 1/64: 64^2 = 4096
  return g;
}

}  // namespace

int wvn_slic_num_clusters_impl(int H, int W, int num_components) { return slic_geom(H, W, num_components, 10.f).K; }

size_t wvn_slic_scratch_bytes_impl(int H, int W, int num_components) {
  const SlicGeom g = slic_geom(H, W, num_components, 10.f);
  return align_up((size_t)3 * H * W * sizeof(int), 256) + align_up((size_t)5 * g.K * sizeof(int), 256) +
         align_up((size_t)6 * g.K * sizeof(long long), 256);
}

int wvn_slic_launch(const void* img, int img_u8, int H, int W, int num_components, float compactness, int iters,
                    const int* lut_lin, const int* lut_f, int* labels, void* scratch, size_t scratch_bytes, hipStream_t st) {
  if (!img || !lut_lin || !lut_f || !labels || !scratch || H <= 0 || W <= 0 || num_components <= 0 || iters < 1) return WVN_ERR_ARG;
  if (scratch_bytes < wvn_slic_scratch_bytes_impl(H, W, num_components)) return WVN_ERR_WORKSPACE;
  const SlicGeom g = slic_geom(H, W, num_components, compactness);
  if ((size_t)g.K * 5 * sizeof(int) > 60 * 1024) return WVN_ERR_ARG;
  int* lab = (int*)scratch;
  int* cent = (int*)((char*)scratch + align_up((size_t)3 * H * W * sizeof(int), 256));
  long long* sums = (long long*)((char*)cent + align_up((size_t)5 * g.K * sizeof(int), 256));
  const int npix = H * W;
  if (img_u8) hipLaunchKernelGGL(slic_lab_kernel<unsigned char>, dim3(ceil_div(npix, 256)), dim3(256), 0, st, (const unsigned char*)img, lut_lin, lut_f, lab, npix);
  else hipLaunchKernelGGL(slic_lab_kernel<float>, dim3(ceil_div(npix, 256)), dim3(256), 0, st, (const float*)img, lut_lin, lut_f, lab, npix);
  WVN_LAUNCH_CHECK();
  hipError_t e = hipMemsetAsync(sums, 0, (size_t)6 * g.K * sizeof(long long), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(slic_init_kernel, dim3(ceil_div(g.K, 256)), dim3(256), 0, st, lab, cent, g);
  WVN_LAUNCH_CHECK();
  for (int it = 0; it < iters; ++it) {
    const int last = it == iters - 1;
    hipLaunchKernelGGL(slic_assign_kernel, dim3(ceil_div(npix, 256)), dim3(256), (size_t)5 * g.K * sizeof(int), st, lab, cent, labels,
                       sums, g, last ? 0 : 1);
    WVN_LAUNCH_CHECK();
    if (!last) {
      hipLaunchKernelGGL(slic_update_kernel, dim3(ceil_div(g.K, 256)), dim3(256), 0, st, cent, sums, g.K);
      WVN_LAUNCH_CHECK();
    }
  }
  return WVN_OK;
}
