// Supervision-mask path of TraversabilityEstimator.add_supervision_node (traversability_estimator.py:261-289): for every
// mission node in range, project the footprint polygon of the newest supervision node into the node's camera
// (ImageProjector.project, image_projector.py:126-150), fill the convex polygon (ImageProjector.project_and_render :152-197,
// kornia.utils.draw_convex_polygon) and merge it into the node's supervision mask with torch.fmin (:281-286) -- ONE launch
// for all nodes, masks updated in place; the label re-pooling (:287-289) is the batched label-pool launch of segments.hip.
//
// Arithmetic follows the published kornia routines step by step in fp32, with un-fused multiply/add (this file is compiled
// with -ffp-contract=off) in one fixed order, so that the CPU oracle (oracle/supervision.py) reproduces every pixel:
//   T_cw            = rigid inverse of pose_cam_in_world: R^T, -(R^T t)   [the reference calls torch.inverse; same to 1e-7]
//   p_c             = R_cw p_w + t_cw                                      (kornia transform_points; w stays 1)
//   (x', y', z')    = K[:3,:3] p_c + K[:3,3]                               (PinholeCamera.project: P = K @ I)
//   u = x' * s, v = y' * s, s = |z'| > 1e-8 ? 1 / (z' + 1e-8) : 1          (convert_points_from_homogeneous)
//   points behind the camera (p_c.z < 0) become NaN                        (image_projector.py:180)
//   scan lines (kornia _get_convex_edges): polygon closed if its last vertex differs from the first; per edge
//       dx = clamp((x1 - x0) / (y1 - y0 + 1e-12), -W, W);  xs(y) = (y - y0) * dx + x0, counted where y lies between y0 and y1
//       x_left(y) = min over active edges (W if none), x_right(y) = max (-1 if none);  pixel x is inside iff x_left <= x <= x_right
//   mask            = inside ? colour * traversability : NaN               (:186-195 with colour = 1, :276)
//   node mask       = fmin(node mask, mask)  ==  inside ? fmin(old, value) : old
// kornia is absent from this image (and from /root/reference): PARITY UNPINNED for the scan-line rule; the restatement is
// from the published kornia >= 0.6.7 source, and the oracle is cross-checked against an independent even-odd polygon test
// on interior pixels (tests/test_oracle_supervision.py).
#include "common.h"
#include "wvn_internal.h"

namespace {

// vertex arrays live in dynamic LDS: the footprint of a traversable node has <= 40 points, the "untraversable plane" of
// SupervisionNode.make_footprint_with_node (nodes.py:575-580: make_dense_plane, a 10 x 10 x 10 grid) has 1000
constexpr size_t MAX_LDS_BYTES = 64 * 1024;

struct RenderNode {
  const float* K;      // [4][4] scaled camera matrix of this node
  const float* pose;   // [4][4] pose_cam_in_world
  float* mask;         // [C][H][W] supervision mask, updated in place
  float* projected;    // [N][2] out (may be null): raw coordinates, finite also behind the camera
  float* depth;        // [N] out (may be null): camera-frame z
};

__global__ __launch_bounds__(256) void project_render_fmin_kernel(const RenderNode* __restrict__ nodes,
                                                                  const float* __restrict__ pts, int pts_batched, int N,
                                                                  int C, int H, int W, const float* __restrict__ value_dev,
                                                                  float value_host) {
  __shared__ int closed_n;
  extern __shared__ float rows[];  // [H] x_left, [H] x_right, [N + 1] px, [N + 1] py
  float* xl = rows;
  float* xr = rows + H;
  float* px = rows + 2 * H;
  float* py = px + (N + 1);
  const RenderNode nd = nodes[blockIdx.x];
  const float* P = pts + (pts_batched ? (size_t)blockIdx.x * N * 3 : 0);
  const float value = value_dev ? value_dev[0] : value_host;
  const int tid = threadIdx.x;
  // ---- projection ----
  for (int i = tid; i < N; i += blockDim.x) {
    const float* T = nd.pose;
    const float X = P[3 * i], Y = P[3 * i + 1], Z = P[3 * i + 2];
    float pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // row r of R^T is column r of R; t_cw[r] = -(R[0][r] t0 + R[1][r] t1 + R[2][r] t2)
      const float r0 = T[0 * 4 + r], r1 = T[1 * 4 + r], r2 = T[2 * 4 + r];
      const float tc = -((r0 * T[3] + r1 * T[7]) + r2 * T[11]);
      pc[r] = ((r0 * X + r1 * Y) + r2 * Z) + tc;
    }
    const float* K = nd.K;
    const float xp = ((K[0] * pc[0] + K[1] * pc[1]) + K[2] * pc[2]) + K[3];
    const float yp = ((K[4] * pc[0] + K[5] * pc[1]) + K[6] * pc[2]) + K[7];
    const float zp = ((K[8] * pc[0] + K[9] * pc[1]) + K[10] * pc[2]) + K[11];
    const float s = fabsf(zp) > 1e-8f ? 1.0f / (zp + 1e-8f) : 1.0f;
    float u = xp * s, v = yp * s;
    if (nd.projected) { nd.projected[2 * i] = u; nd.projected[2 * i + 1] = v; }
    if (nd.depth) nd.depth[i] = pc[2];
    if (!(pc[2] >= 0.f)) { u = __builtin_nanf(""); v = __builtin_nanf(""); }
    px[i] = u; py[i] = v;
  }
  __syncthreads();
  if (tid == 0) {
    // torch.allclose(last, first) (rtol 1e-5, atol 1e-8; NaN never close) -> otherwise the first vertex is appended
    const float ax = px[N - 1], bx = px[0], ay = py[N - 1], by = py[0];
    const bool close = fabsf(ax - bx) <= 1e-8f + 1e-5f * fabsf(bx) && fabsf(ay - by) <= 1e-8f + 1e-5f * fabsf(by);
    if (!close) { px[N] = bx; py[N] = by; closed_n = N + 1; } else closed_n = N;
  }
  __syncthreads();
  const int ne = closed_n - 1;
  // ---- scan lines ----
  for (int y = tid; y < H; y += blockDim.x) {
    const float fy = (float)y;
    float l = (float)W, r = -1.f;
    bool poisoned = false;
    for (int e = 0; e < ne; ++e) {
      const float x0 = px[e], y0 = py[e], x1 = px[e + 1], y1 = py[e + 1];
      const bool act = (y0 <= fy && fy <= y1) || (y0 >= fy && fy >= y1);
      if (!act) continue;
      float dx = (x1 - x0) / ((y1 - y0) + 1e-12f);
      if (dx == dx) dx = fminf(fmaxf(dx, -(float)W), (float)W);   // torch.clamp keeps a NaN (inf - inf over an infinite vertex)
      const float xs = (fy - y0) * dx + x0;
      // torch's min / max over the edges propagate a NaN (an infinite vertex passes the comparisons above; inf * 0 and
      // inf - inf are NaN): such a scan line is left unfilled
      poisoned |= xs != xs;
      l = fminf(l, xs);
      r = fmaxf(r, xs);
    }
    if (poisoned) l = r = __builtin_nanf("");
    xl[y] = l; xr[y] = r;
  }
  __syncthreads();
  // ---- fill + fmin merge: one wave per scan line, lanes along x (coalesced) ----
  const int wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
  for (int y = wave; y < H; y += nw) {
    const float l = xl[y], r = xr[y];
    if (!(r >= l)) continue;
    int xa = (int)ceilf(fmaxf(l, 0.f));
    int xb = (int)floorf(fminf(r, (float)(W - 1)));
    for (int x = xa + lane; x <= xb; x += 64) {
      if (!((float)x >= l && (float)x <= r)) continue;
      for (int c = 0; c < C; ++c) {
        float* m = nd.mask + ((size_t)c * H + y) * W + x;
        *m = fminf(*m, value);   // fmin semantics: a NaN operand yields the other one
      }
    }
  }
}

}  // namespace

int wvn_project_render_fmin_launch(const void* nodes, int n, const float* points, int points_batched, int npts, int C, int H,
                                   int W, const float* value_dev, float value, hipStream_t st) {
  if (!nodes || !points || n <= 0 || npts < 2 || C <= 0 || H <= 0 || W <= 0) return WVN_ERR_ARG;
  const size_t lds = ((size_t)H * 2 + ((size_t)npts + 1) * 2) * sizeof(float);
  if (lds > MAX_LDS_BYTES) return WVN_ERR_ARG;   // H = 448: up to 7700 footprint points
  hipLaunchKernelGGL(project_render_fmin_kernel, dim3(n), dim3(256), lds, st,
                     (const RenderNode*)nodes, points, points_batched, npts, C, H, W, value_dev, value);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
