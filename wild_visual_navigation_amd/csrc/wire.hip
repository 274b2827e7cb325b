// A -> B wire format of the two WVN processes.  The feature-extractor node publishes, per frame, the message
// wild_visual_navigation_msgs/ImageFeatures = { Header, sensor_msgs/Image feature_segments (int32, "passthrough"),
// std_msgs/Float32MultiArray features (dims n x feat) } and builds it with seg.cpu().numpy().astype(np.int32) +
// feat.cpu().numpy().flatten().tolist() (wvn_feature_extractor_node.py:373-393): two device->host copies, a host cast and
// a Python list of S*D floats.  The learning node undoes it with np.array(ma.data, dtype=float).reshape(dims).astype(float32)
// (wvn_learning_node.py:651-656).  Here ONE kernel lays the frame out on the device exactly as the message carries it --
//     [ 64-byte header | int32 segments, H*W, row-major | float32 features, S*D, row-major ]
// -- so the publisher needs one device->host copy of one contiguous buffer whose two payload sections ARE the byte arrays of
// the ROS message fields (Image.data with step = 4*W; Float32MultiArray.data), and the subscriber one host->device copy plus
// the inverse kernel (int32 -> int64 segments, as MissionNode stores them).  Bit-exact: float32 features are copied, segment
// ids are converted int64 <-> int32 (ids are < 2^31; -1 = "no segment" survives).
#include "common.h"
#include "wvn_internal.h"

namespace {

constexpr unsigned WIRE_MAGIC = 0x464e5657u;  // "WVNF"
constexpr int WIRE_HEADER = 64;

struct WireHeader { unsigned magic, version; int H, W, S, D; unsigned seg_offset, feat_offset; unsigned pad[8]; };
static_assert(sizeof(WireHeader) == WIRE_HEADER, "header is 64 bytes");

__global__ void wire_pack_kernel(const void* __restrict__ seg, int seg_is_i64, const float* __restrict__ feat, int ldf,
                                 unsigned char* __restrict__ out, int H, int W, int S, int D) {
  const long long npix = (long long)H * W, nfeat = (long long)S * D;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    WireHeader h{};
    h.magic = WIRE_MAGIC; h.version = 1; h.H = H; h.W = W; h.S = S; h.D = D;
    h.seg_offset = WIRE_HEADER; h.feat_offset = WIRE_HEADER + (unsigned)(npix * 4);
    *(WireHeader*)out = h;
  }
  if (i < npix) {
    const int v = seg_is_i64 ? (int)((const long long*)seg)[i] : ((const int*)seg)[i];
    ((int*)(out + WIRE_HEADER))[i] = v;
  }
  if (i < nfeat) {
    const int r = (int)(i / D), c = (int)(i - (long long)r * D);
    ((float*)(out + WIRE_HEADER + npix * 4))[i] = feat[(size_t)r * ldf + c];
  }
}

__global__ void wire_unpack_kernel(const unsigned char* __restrict__ in, long long* __restrict__ seg_i64, int* __restrict__ seg_i32,
                                   float* __restrict__ feat, int H, int W, int S, int D) {
  const long long npix = (long long)H * W, nfeat = (long long)S * D;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npix) {
    const int v = ((const int*)(in + WIRE_HEADER))[i];
    if (seg_i64) seg_i64[i] = v;
    if (seg_i32) seg_i32[i] = v;
  }
  if (i < nfeat) feat[i] = ((const float*)(in + WIRE_HEADER + npix * 4))[i];
}

}  // namespace

size_t wvn_wire_bytes_impl(int H, int W, int S, int D) { return (size_t)WIRE_HEADER + (size_t)H * W * 4 + (size_t)S * D * 4; }

int wvn_wire_pack_launch(const void* seg, int seg_is_i64, const float* feat, int ldf, void* out, int H, int W, int S, int D,
                         hipStream_t st) {
  if (!seg || !feat || !out || H <= 0 || W <= 0 || S <= 0 || D <= 0 || ldf < D || ((uintptr_t)out & 15)) return WVN_ERR_ARG;
  const long long n = (long long)H * W > (long long)S * D ? (long long)H * W : (long long)S * D;
  hipLaunchKernelGGL(wire_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seg, seg_is_i64, feat, ldf,
                     (unsigned char*)out, H, W, S, D);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}

int wvn_wire_unpack_launch(const void* in, long long* seg_i64, int* seg_i32, float* feat, int H, int W, int S, int D,
                           hipStream_t st) {
  if (!in || !feat || (!seg_i64 && !seg_i32) || H <= 0 || W <= 0 || S <= 0 || D <= 0 || ((uintptr_t)in & 15)) return WVN_ERR_ARG;
  const long long n = (long long)H * W > (long long)S * D ? (long long)H * W : (long long)S * D;
  hipLaunchKernelGGL(wire_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const unsigned char*)in, seg_i64,
                     seg_i32, feat, H, W, S, D);
  WVN_LAUNCH_CHECK();
  return WVN_OK;
}
