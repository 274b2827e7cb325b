// C-ABI of libwvn_hip.so (see include/wvn_hip.h) and the host-side launch sequences:
// the ViT forward chain and the three phases of the traversability-MLP optimisation step.
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/wvn_hip.h"
#include <cstdlib>

#include "common.h"
#include "wvn_internal.h"

#define RET_IF(x)            \
  do {                       \
    int rc__ = (x);          \
    if (rc__ != WVN_OK) return rc__; \
  } while (0)

// ---------------------------------------------------------------------------------------------
// profiling: HIP events around each launch category of wvn_vit_forward, on the launch stream
// ---------------------------------------------------------------------------------------------
namespace {
struct ProfSpan { hipEvent_t a, b; int cat; };
bool g_prof_on = false;
std::vector<ProfSpan> g_spans;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t get_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct Span {
  hipStream_t st; hipEvent_t a{}, b{}; int cat; bool on;
  Span(int cat_, hipStream_t st_) : st(st_), cat(cat_), on(g_prof_on) {
    if (on) { a = get_event(); b = get_event(); (void)hipEventRecord(a, st); }
  }
  ~Span() {
    if (on) { (void)hipEventRecord(b, st); g_spans.push_back({a, b, cat}); }
  }
};

// the launchers of the 16-bit-operand speed path, per operand format (operand.h)
struct OperandKernels {
  int fmt;  // 1 bf16, 2 fp16 (the y_fmt / out_mode codes of the elementwise launchers: patchify out_mode = fmt == 2 ? 3 : 1)
  decltype(&wvn_gemm_bf16_launch) gemm;
  decltype(&wvn_qkv_fused_launch) qkv_fused;
  decltype(&wvn_attention_bf16_launch) attention;
  decltype(&wvn_proj_mlp_fused_launch) proj_mlp_fused;
  decltype(&wvn_mlp_fused_launch) mlp_fused;
};
const OperandKernels OPK_BF16 = {1, wvn_gemm_bf16_launch, wvn_qkv_fused_launch, wvn_attention_bf16_launch, wvn_proj_mlp_fused_launch,
                                 wvn_mlp_fused_launch};
const OperandKernels OPK_F16 = {2, wvn_gemm_bf16_launch_f16, wvn_qkv_fused_launch_f16, wvn_attention_bf16_launch_f16,
                                wvn_proj_mlp_fused_launch_f16, wvn_mlp_fused_launch_f16};

struct VitDims {
  int B, S, P, G, D, H, F, KP, KPs, ntok, ntok_s, npad, npatch;
  bool fp8, planes;   // planes: WVN_PREC_X3 / WVN_PREC_MIX (hi + lo bf16 planes)
  size_t esz;
  long long M, Mp;
};
VitDims vit_dims(const wvn_vit_model* m, int batch) {
  VitDims d;
  d.B = batch; d.S = m->img_size; d.P = m->patch; d.G = d.S / d.P; d.D = m->dim; d.H = m->heads; d.F = m->mlp_dim;
  d.KP = 3 * d.P * d.P; d.npatch = d.G * d.G; d.ntok = d.npatch + 1;
  // patch rows as the MFMA GEMMs read them: K padded to a multiple of 64 (588 -> 640 for patch 14; 192 and 768 unchanged);
  // the fp32 FMA path reads the unpadded rows
  d.KPs = m->precision == WVN_PREC_F32 ? d.KP : (d.KP + 63) / 64 * 64;
  d.fp8 = m->precision == WVN_PREC_FP8;
  d.planes = m->precision == WVN_PREC_X3 || m->precision == WVN_PREC_MIX;
  d.ntok_s = (d.ntok + 15) / 16 * 16;  // rows per frame: 8-token (16 B) chunks and the 16-token V^T permutation groups never straddle frames
  d.npad = (d.ntok + 127) / 128 * 128;
  d.esz = (m->precision == WVN_PREC_BF16 || m->precision == WVN_PREC_FP8 || m->precision == WVN_PREC_F16) ? 2 : 4;  // exact mode (X3): two bf16 planes = 4 bytes per element
  d.M = (long long)batch * d.ntok_s; d.Mp = (long long)batch * d.npatch;
  return d;
}
struct VitWs { float* x; void* xn; void* q; void* k; void* v; void* hid; void* patches; unsigned char* xq; unsigned char* hq; float* sa; float* ln_stats; size_t total; };
VitWs vit_carve(const VitDims& d, void* base) {
  VitWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return (void*)((char*)base + o); };
  w.x = (float*)take((size_t)d.M * d.D * 4);
  w.xn = take((size_t)(d.planes ? (d.M + 31) / 32 * 32 : d.M) * d.D * d.esz);
  size_t qkv = (size_t)d.B * d.H * d.npad * 64 * d.esz;
  w.q = take(qkv); w.k = take(qkv); w.v = take(qkv);
  w.hid = take((size_t)(d.planes ? (d.M + 31) / 32 * 32 : d.M) * d.F * d.esz);   // (planes: whole 32-row groups for the fragment-major MLP)
  w.patches = take((size_t)d.Mp * d.KPs * d.esz);
  w.xq = nullptr; w.hq = nullptr; w.sa = nullptr;
  if (d.fp8) {  // e4m3 images of the GEMM inputs + their per-row scales
    w.xq = (unsigned char*)take((size_t)d.M * d.D);
    w.hq = (unsigned char*)take((size_t)d.M * d.F);
    w.sa = (float*)take((size_t)d.M * 4);
  }
  w.ln_stats = d.planes ? (float*)take((size_t)d.M * 8) : nullptr;   // {mean, rstd} per row: LayerNorm across kernel boundaries (split-operand block kernels)
  w.total = off;
  return w;
}
}  // namespace

extern "C" {

int wvn_version(void) { return 100; }

int wvn_prof_enable(int on) { g_prof_on = on != 0; return WVN_OK; }

int wvn_prof_collect(double* ms, long long* launches) {
  for (int i = 0; i < WVN_PROF_NCAT; ++i) { ms[i] = 0.0; launches[i] = 0; }
  for (auto& s : g_spans) {
    hipError_t e = hipEventSynchronize(s.b);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, s.a, s.b);
    if (e != hipSuccess) return (int)e;
    if (s.cat >= 0 && s.cat < WVN_PROF_NCAT) { ms[s.cat] += t; launches[s.cat] += 1; }
    g_event_pool.push_back(s.a);
    g_event_pool.push_back(s.b);
  }
  g_spans.clear();
  return WVN_OK;
}

size_t wvn_vit_workspace_bytes(const wvn_vit_model* m, int batch) {
  if (!m || batch <= 0) return 0;
  VitDims d = vit_dims(m, batch);
  return vit_carve(d, nullptr).total;
}

static int vit_forward_impl(const wvn_vit_model* m, const void* img, int img_u8, const WvnIngest* ing, int batch, float* tokens_f32,
                            void* tokens_lowp, int ld_lowp, void* workspace, size_t workspace_bytes, void* stream,
                            const int* cols_mirror = nullptr);

int wvn_vit_forward(const wvn_vit_model* m, const float* img, int batch, float* tokens_f32, void* tokens_lowp,
                    int ld_lowp, void* workspace, size_t workspace_bytes, void* stream) {
  return vit_forward_impl(m, img, 0, nullptr, batch, tokens_f32, tokens_lowp, ld_lowp, workspace, workspace_bytes, stream);
}

int wvn_vit_forward_u8(const wvn_vit_model* m, const unsigned char* img, int batch, float* tokens_f32, void* tokens_lowp,
                       int ld_lowp, void* workspace, size_t workspace_bytes, void* stream) {
  if (m && ((m->precision != WVN_PREC_BF16 && m->precision != WVN_PREC_FP8 && m->precision != WVN_PREC_F16) || m->patch != 8)) return WVN_ERR_ARG;
  return vit_forward_impl(m, img, 1, nullptr, batch, tokens_f32, tokens_lowp, ld_lowp, workspace, workspace_bytes, stream);
}

int wvn_vit_forward_frames(const wvn_vit_model* m, const void* frames, int frames_u8, int src_h, int src_w, const int* rows,
                           const int* cols, int batch, float* tokens_f32, void* tokens_lowp, int ld_lowp, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (!rows || !cols || src_h <= 0 || src_w <= 0) return WVN_ERR_ARG;
  const WvnIngest ing{rows, cols, src_h, src_w};
  return vit_forward_impl(m, frames, frames_u8 != 0, &ing, batch, tokens_f32, tokens_lowp, ld_lowp, workspace, workspace_bytes, stream);
}

int wvn_vit_forward_frames_pair(const wvn_vit_model* m, const void* frames, int frames_u8, int src_h, int src_w, const int* rows,
                                const int* cols, const int* cols_mirror, int batch, float* tokens_f32, void* tokens_lowp, int ld_lowp,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!rows || !cols || !cols_mirror || src_h <= 0 || src_w <= 0) return WVN_ERR_ARG;
  const WvnIngest ing{rows, cols, src_h, src_w};
  return vit_forward_impl(m, frames, frames_u8 != 0, &ing, batch, tokens_f32, tokens_lowp, ld_lowp, workspace, workspace_bytes, stream,
                          cols_mirror);
}

int wvn_resize_nearest_crop(const void* in, void* out, long long planes, int src_h, int src_w, const int* rows, const int* cols,
                            int out_h, int out_w, int elem_bytes, void* stream) {
  const WvnIngest ing{rows, cols, src_h, src_w};
  return wvn_gather_image_launch(in, out, planes, out_h, out_w, elem_bytes, &ing, (hipStream_t)stream);
}

}  // extern "C"

// cols_mirror (with ing): the `batch` frames go through the network TWICE in one launch sequence of 2 * batch frames -- frame
// batch + i is frame i gathered through the second column table (its mirror image: the flip pass of the upstream Stego.get_code).
// Twice the rows per launch: the persistent block kernels end on a thinner partial round (6.2 -> 12.3 rounds of row blocks).
static int vit_forward_impl(const wvn_vit_model* m, const void* img, int img_u8, const WvnIngest* ing, int batch, float* tokens_f32,
                            void* tokens_lowp, int ld_lowp, void* workspace, size_t workspace_bytes, void* stream,
                            const int* cols_mirror) {
  if (!m || !img || !workspace || batch <= 0 || (cols_mirror && !ing)) return WVN_ERR_ARG;
  const int frames_in = batch;
  if (cols_mirror) batch *= 2;
  if (m->dim != m->heads * 64 || m->depth <= 0 || m->depth > WVN_MAX_DEPTH || m->img_size % m->patch) return WVN_ERR_ARG;
  if (m->dim % 128 || m->mlp_dim % 128) return WVN_ERR_ARG;
  if (m->precision < WVN_PREC_F32 || m->precision > WVN_PREC_MIX) return WVN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const VitDims d = vit_dims(m, batch);
  const VitWs w = vit_carve(d, workspace);
  if (w.total > workspace_bytes) return WVN_ERR_WORKSPACE;
  const bool fp8 = m->precision == WVN_PREC_FP8;
  if (fp8 && ((d.D % 128) || (d.F % 128))) return WVN_ERR_ARG;
  // fp8: everything that is not one of the four block linears runs exactly as in the bf16 mode
  // "bf": the 16-bit-operand speed path, in either operand format (opk)
  const bool f16 = m->precision == WVN_PREC_F16;
  const OperandKernels& opk = f16 ? OPK_F16 : OPK_BF16;
  // mix (WVN_PREC_MIX): every linear as in the exact mode (x3), the two attention products on the fp16-operand kernel
  const bool mix = m->precision == WVN_PREC_MIX;
  const bool bf = m->precision == WVN_PREC_BF16 || fp8 || f16, x3 = m->precision == WVN_PREC_X3 || mix, f32 = m->precision == WVN_PREC_F32;
  const bool lowp16 = m->precision == WVN_PREC_BF16 || f16;   // the precisions the single-kernel block stages exist for
  // The single-kernel block stages are persistent one-workgroup-per-CU kernels (128 / 256 rows per workgroup): measured against the
  // separate kernels (scripts/small_batch_latency.py, 448^2 and 224^2 frames) they win from about half a chip of row blocks on and
  // lose below -- a single live frame is 25 / 13 row blocks on 256 CUs, 2.5 ms against 1.7.
  const bool mlp_ok = (m->flags & WVN_VIT_MLP_FUSED) != 0, qkv_ok = (m->flags & WVN_VIT_QKV_FUSED) != 0;
  if (qkv_ok && (!lowp16 || d.D != 384 || d.H != 6 || (d.ntok_s % 16) != 0)) return WVN_ERR_ARG;
  const bool any_size = (m->flags & WVN_VIT_FUSE_ANY_SIZE) != 0, proj_in_mlp = (m->flags & WVN_VIT_NO_PROJ_IN_MLP) == 0;
  const bool mlp_fused = mlp_ok && (any_size || d.M >= 128 * 128), qkv_fused = qkv_ok && (any_size || d.M >= 144 * 256);
  if (mlp_ok && (!lowp16 || d.D != 384 || (d.F % 64) != 0)) return WVN_ERR_ARG;
  if (x3 && tokens_lowp) return WVN_ERR_ARG;  // exact mode hands out fp32 tokens only (callers split with wvn_split_planes)
  const float scale = 1.0f / sqrtf(64.f);
  const int M = (int)d.M, Mp = (int)d.Mp;
  // exact mode: an activation / weight "matrix" is two stacked bf16 planes, hi then lo
  const size_t pl_xn = (size_t)(d.planes ? (d.M + 31) / 32 * 32 : d.M) * d.D, pl_hid = (size_t)d.M * d.F, pl_qkv = (size_t)d.B * d.H * d.npad * 64,
               pl_pat = (size_t)d.Mp * d.KPs;
  auto lo = [](const void* base, size_t plane_elems) { return (bf16_t*)base + plane_elems; };
  bool stats_written = false;   // set by linear(): the row-panel kernel ran and left the statistics it was asked for

  // One linear of the chain in the model's precision.  A: activation matrix (bf16 | hi+lo planes | fp32), W: weight in the
  // same representation ([N][K], planes stacked), epilogue codes of GemmEpilogue.
  auto linear = [&](const void* A, size_t a_plane, int lda, const void* W, const float* bias, void* C, size_t c_plane, int ldc,
                    int rows, int N, int K, int epi, const float* ls, GemmBf16Params* extra) -> int {
    if (f32) {
      GemmF32Params p{};
      p.A = (const float*)A; p.lda = lda; p.B = (const float*)W; p.ldb = K; p.transB = 1; p.bias = bias;
      p.C = (float*)C; p.ldc = ldc; p.M = rows; p.N = N; p.K = K; p.batch = 1; p.splitk = 1; p.ls = ls;
      int fe = F32_EPI_NONE;
      switch (epi) {
        case EPI_GELU_BF16: fe = F32_EPI_GELU; break;
        case EPI_RESID_F32: fe = F32_EPI_RESID; break;
        case EPI_PATCH: fe = F32_EPI_PATCH; p.pos = m->pos; p.npatch = d.npatch; p.ntok = d.ntok; p.ntok_s = d.ntok_s; break;
        case EPI_QKV:
          fe = F32_EPI_QKV; p.C = (float*)w.q; p.q = (float*)w.q; p.k = (float*)w.k; p.v = (float*)w.v; p.heads = d.H;
          p.npad = d.npad; p.ntok = d.ntok; p.ntok_s = d.ntok_s; break;
        default: return WVN_ERR_ARG;
      }
      return wvn_gemm_f32_launch(p, fe, st);
    }
    GemmBf16Params p{};
    if (extra) p = *extra;
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = K; p.bias = bias; p.C = C; p.ldc = ldc;
    p.M = rows; p.N = N; p.K = K; p.ls = ls;
    if (x3) {
      p.A_lo = lo(A, a_plane); p.W_lo = lo(W, (size_t)N * K); p.C_lo = C ? lo(C, c_plane) : nullptr;
      if (N == 384 && (epi == EPI_RESID_F32 || epi == EPI_PATCH) && rows >= 64 * 128 && !(m->flags & (WVN_VIT_NO_A384_X3 | WVN_VIT_X3_NO_N384))) {   // row panel: fc2, projection, patch embedding
        const int rc = wvn_gemm_n384_x3_launch(p, epi, st);
        if (rc == WVN_OK && p.ln_stats_out) stats_written = true;   // (the tiled kernel below leaves no LayerNorm statistics)
        if (rc != WVN_ERR_ARG) return rc;
      }
      if (K == 384 && rows >= 64 * 128 && !(m->flags & (WVN_VIT_NO_A384_X3 | WVN_VIT_X3_NO_A384))) {   // the A-stationary form from about a quarter chip of row blocks on
        const int rc = wvn_gemm_a384_x3_launch(p, epi, st);
        if (rc != WVN_ERR_ARG) return rc;
      }
      return wvn_gemm_x3_launch(p, epi, st);
    }
    return opk.gemm(p, epi, st);
  };

  {
    Span s(0, st);
    RET_IF(wvn_patchify_launch(img, img_u8, w.patches, x3 ? lo(w.patches, pl_pat) : nullptr, f32 ? 0 : (x3 ? 2 : (f16 ? 3 : 1)), d.KPs,
                               frames_in, d.S, d.P, st, ing));
    if (cols_mirror) {   // the mirror images' patch rows behind the frames'
      WvnIngest ing2 = *ing;
      ing2.cols = cols_mirror;
      const size_t off = (size_t)frames_in * d.npatch * d.KPs;   // elements
      void* p2 = (char*)w.patches + off * (f32 ? 4 : 2);
      RET_IF(wvn_patchify_launch(img, img_u8, p2, x3 ? lo(w.patches, pl_pat) + off : nullptr, f32 ? 0 : (x3 ? 2 : (f16 ? 3 : 1)), d.KPs,
                                 frames_in, d.S, d.P, st, &ing2));
    }
    if (d.KPs != d.KP) {  // zero the K padding of the patch rows (weights are zero there too, but NaN * 0 must not happen)
      RET_IF(wvn_pad_zero_launch(w.patches, (long long)Mp * (x3 ? 2 : 1), (long long)d.KPs * 2, (long long)d.KP * 2,
                                 (long long)(d.KPs - d.KP) * 2, st));
    }
  }
  RET_IF(wvn_cls_rows_launch(m->cls_pos, w.x, d.B, d.ntok_s, d.D, st));
  {
    // Padding hygiene, every call (the carve depends on the batch, so a reused workspace holds stale bytes):
    // residual-stream rows [ntok, ntok_s) start at zero (they then carry finite values through the blocks), and the
    // never-written key/value slots [ntok_s, npad) are zero.  The attention kernels mask padded keys by score, but
    // their V^T / K bytes still enter MFMAs and must be finite.
    RET_IF(wvn_pad_zero_launch(w.x, d.B, (long long)d.ntok_s * d.D * 4, (long long)d.ntok * d.D * 4,
                               (long long)(d.ntok_s - d.ntok) * d.D * 4, st));
    // with the fused LayerNorm + QKV kernel nothing writes the padding rows of xn (the attention output buffer the projection
    // GEMM reads whole): they used to hold LayerNorm 1's output
    if (qkv_fused)
      RET_IF(wvn_pad_zero_launch(w.xn, d.B, (long long)d.ntok_s * d.D * 2, (long long)d.ntok * d.D * 2, (long long)(d.ntok_s - d.ntok) * d.D * 2, st));
    const long long nbh = (long long)d.B * d.H;
    const long long tokb = f32 ? 256 : 128, npl = (x3 && !mix) ? 2 : 1;  // bytes per token row of q / k; planes per tensor
    RET_IF(wvn_pad_zero_launch(w.q, nbh * (x3 ? 2 : 1), d.npad * tokb, d.ntok_s * tokb, (d.npad - d.ntok_s) * tokb, st));   // (mix: q has two fp16 planes)
    RET_IF(wvn_pad_zero_launch(w.k, nbh * npl, d.npad * tokb, d.ntok_s * tokb, (d.npad - d.ntok_s) * tokb, st));
    if (!f32)  // V^T [B*h*64][npad] (bf16, or hi / lo planes)
      RET_IF(wvn_pad_zero_launch(w.v, nbh * 64 * npl, (long long)d.npad * 2, (long long)d.ntok_s * 2, (long long)(d.npad - d.ntok_s) * 2, st));
    else     // V [B*h][npad][64]
      RET_IF(wvn_pad_zero_launch(w.v, nbh, d.npad * tokb, d.ntok_s * tokb, (d.npad - d.ntok_s) * tokb, st));
  }
  {
    Span s(1, st);
    GemmBf16Params e{};
    e.pos = m->pos; e.npatch = d.npatch; e.ntok = d.ntok; e.ntok_s = d.ntok_s;
    RET_IF(linear(w.patches, pl_pat, d.KPs, m->patch_w, m->patch_b, w.x, 0, d.D, Mp, d.D, d.KPs, EPI_PATCH, nullptr, &e));
  }
  // fp8: quantise-then-GEMM for the four linears of a block (A rows -> e4m3 + per-token scale in ws.xq / ws.hq / ws.sa)
  // (Wp: the packed image of a K = 768 weight for the A-stationary kernel, from 4096 rows on; elsewhere, or where that kernel declines, the tiled kernel)
  auto linear_fp8 = [&](const unsigned char* Aq, int K, const void* W, const void* Wp, const float* sw, const float* bias, void* C, int ldc, int N,
                        int epi, const float* ls, const GemmBf16Params* extra) -> int {
    GemmFp8Params p{};
    p.A = Aq; p.lda = K; p.W = (const unsigned char*)W; p.ldw = K; p.sa = w.sa; p.sw = sw; p.bias = bias; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.ls = ls;
    if (extra) { p.q = extra->q; p.k = extra->k; p.vt = extra->vt; p.heads = extra->heads; p.npad = extra->npad; p.ntok = extra->ntok;
                 p.ntok_s = extra->ntok_s; p.q_scale = extra->q_scale; }
    if (Wp && K == 768 && M >= 4096) {
      const int rc = wvn_gemm_a768_fp8_launch(p, Wp, epi, st);
      if (rc != WVN_ERR_ARG) return rc;
    }
    return wvn_gemm_fp8_launch(p, epi, st);
  };
  // the split-operand block kernels (A-stationary K = 384 with the LayerNorm in its prologue, fragment-major MLP): from 8192 rows on
  const bool x3_fast = x3 && d.D == 384 && M >= 64 * 128 && !(m->flags & WVN_VIT_NO_A384_X3);
  bool pre_qkv = false;   // this block's norm1 has been applied by the previous block's projection + MLP kernel (fragments in w.hid)
  // split-operand block kernels: LayerNorm ACROSS kernel boundaries -- the row-panel kernel that updates the residual rows (projection,
  // fc2) leaves their {mean, rstd}, the A-stationary kernel that consumes them (fc1, next block's QKV) normalises as it loads: no LayerNorm
  // kernel, no xn planes, between them.  ln1_stats: w.ln_stats holds the statistics of w.x for THIS block's norm1.
  const bool ln_fuse = x3_fast && w.ln_stats && !(m->flags & WVN_VIT_X3_NO_LN_STATS);
  bool ln1_stats = false;
  for (int l = 0; l < m->depth; ++l) {
    const wvn_vit_layer& L = m->layers[l];
    if (l == 0) pre_qkv = false;
    if (fp8) {
      if (!L.qkv_s || !L.proj_s || !L.fc1_s || !L.fc2_s) return WVN_ERR_ARG;
      { Span s(2, st); RET_IF(wvn_layernorm_fp8_launch(w.x, L.ln1_g, L.ln1_b, w.xq, d.D, w.sa, M, d.D, 1e-6f, st)); }
      {
        Span s(3, st);
        GemmBf16Params e{};
        e.q = (bf16_t*)w.q; e.k = (bf16_t*)w.k; e.vt = (bf16_t*)w.v; e.heads = d.H; e.npad = d.npad; e.ntok = d.ntok; e.ntok_s = d.ntok_s;
        e.q_scale = scale * 1.44269504088896340736f;
        RET_IF(linear_fp8(w.xq, d.D, L.qkv_w, L.qkv_w_mx, L.qkv_s, L.qkv_b, nullptr, 0, 3 * d.D, EPI_QKV, nullptr, &e));
      }
      { Span s(4, st); RET_IF(wvn_attention_bf16_launch((const bf16_t*)w.q, (const bf16_t*)w.k, (const bf16_t*)w.v, (bf16_t*)w.xn, d.B, d.H, d.ntok, d.ntok_s, d.npad, 0.f, st, nullptr)); }
      {
        Span s(5, st);
        RET_IF(wvn_quantize_rows_fp8_launch(w.xn, 1, d.D, w.xq, d.D, w.sa, M, d.D, st));
        RET_IF(linear_fp8(w.xq, d.D, L.proj_w, L.proj_w_mx, L.proj_s, L.proj_b, w.x, d.D, d.D, EPI_RESID_F32, L.ls1, nullptr));
      }
      { Span s(2, st); RET_IF(wvn_layernorm_fp8_launch(w.x, L.ln2_g, L.ln2_b, w.xq, d.D, w.sa, M, d.D, 1e-6f, st)); }
      // fc1 -> fc2 in the MX operand form where the A-stationary kernel runs fc1 (round 6): its GELU epilogue writes the hidden activation as e4m3 with ONE E8M0 block
      // scale per (row, 32 columns) -- a 32-column tile of fc1 is one scale block of fc2's K -- and fc2 multiplies with those block scales on its A operand: no row
      // quantiser (and no bf16 copy of the hidden activation) between the two.  The scales sit in w.hid, which this form does not use otherwise.
      bool hid_mx = false;
      {
        Span s(6, st);
        if (L.fc1_w_mx && d.D == 768 && M >= 4096 && (d.F % 128) == 0 && !getenv("WVN_NO_FP8_MX")) {
          GemmFp8Params p{};
          p.A = w.xq; p.lda = d.D; p.sa = w.sa; p.sw = L.fc1_s; p.bias = L.fc1_b; p.C = w.hq; p.ldc = d.F; p.c_scales = (unsigned char*)w.hid;
          p.M = M; p.N = d.F; p.K = d.D;
          const int rc = wvn_gemm_a768_fp8_launch(p, L.fc1_w_mx, EPI_GELU_MX8, st);
          if (rc != WVN_OK && rc != WVN_ERR_ARG) return rc;
          hid_mx = rc == WVN_OK;
        }
        if (!hid_mx) RET_IF(linear_fp8(w.xq, d.D, L.fc1_w, L.fc1_w_mx, L.fc1_s, L.fc1_b, w.hid, d.F, d.F, EPI_GELU_BF16, nullptr, nullptr));
      }
      {
        Span s(7, st);
        if (hid_mx) {
          GemmFp8Params p{};
          p.A = w.hq; p.lda = d.F; p.a_scales = (const unsigned char*)w.hid; p.W = (const unsigned char*)L.fc2_w; p.ldw = d.F; p.sw = L.fc2_s; p.bias = L.fc2_b;
          p.C = w.x; p.ldc = d.D; p.M = M; p.N = d.D; p.K = d.F; p.ls = L.ls2;
          RET_IF(wvn_gemm_fp8_launch(p, EPI_RESID_F32, st));
        } else {
          RET_IF(wvn_quantize_rows_fp8_launch(w.hid, 1, d.F, w.hq, d.F, w.sa, M, d.F, st));
          RET_IF(linear_fp8(w.hq, d.F, L.fc2_w, nullptr, L.fc2_s, L.fc2_b, w.x, d.D, d.D, EPI_RESID_F32, L.ls2, nullptr));
        }
      }
      continue;
    }
    // WVN_PREC_MIX with a packed projection weight: the attention kernel writes fragments, the projection reads them (both planes of w.xn
    // then hold ceil(M / 32) * 32 rows; pl_xn below is the plane distance of BOTH layouts)
    const bool attn_frag = mix && x3_fast && L.proj_w_frag != nullptr;
    // MX form of the block linears (round 6): fp16 hi * hi + two scaled e5m2 correction MFMAs per 64 k; the activations travel as three planes
    // (fp16 fragments | l8 | h8) through w.xn (attention -> projection) and w.hid (fc1 -> fc2).  Needs the LayerNorm statistics hand-over.
    const bool mx = mix && x3_fast && ln_fuse && d.H * 64 == d.D && (d.F % 128) == 0 && L.qkv_w_mx && L.proj_w_mx && L.fc1_w_mx && L.fc2_w_mx &&
                    !(m->flags & WVN_VIT_NO_MX);
    const size_t mpad32 = (size_t)((d.M + 31) / 32 * 32);
    unsigned char* xn_l8 = (unsigned char*)w.xn + mpad32 * d.D * 2;     // MX planes of the attention output: fp16 fragments | l8 (h8 = e5m2(h) is derived by the consumer)
    unsigned char* hid_l8 = (unsigned char*)w.hid + mpad32 * d.F * 2;   // ... and of the hidden activation
    // the two-plane q in the leading blocks only (include/wvn_hip.h: WVN_VIT_QSPLIT_BLOCKS)
    const int qs_field = (m->flags >> 16) & 63;
    const bool qsplit = mix && l < (qs_field ? qs_field - 1 : WVN_VIT_QSPLIT_DEFAULT);
    bool qkv_done = false;
    if (qkv_fused) {  // LayerNorm 1 + QKV projection: one launch, no xn round trip
      Span s(3, st);
      // (pre_qkv: the previous block's projection + MLP kernel has already applied this block's norm1 to the rows as they left it,
      //  and parked the result as operand fragments in w.hid)
      const int rc = pre_qkv ? opk.qkv_fused(nullptr, 0, nullptr, nullptr, 0.f, (const bf16_t*)L.qkv_w_fused, L.qkv_b, (bf16_t*)w.q, (bf16_t*)w.k,
                                             (bf16_t*)w.v, d.H, d.npad, d.ntok_s, scale * 1.44269504088896340736f, M, st, (const bf16_t*)w.hid)
                             : opk.qkv_fused(w.x, d.D, L.ln1_g, L.ln1_b, 1e-6f, (const bf16_t*)L.qkv_w, L.qkv_b, (bf16_t*)w.q, (bf16_t*)w.k,
                                             (bf16_t*)w.v, d.H, d.npad, d.ntok_s, scale * 1.44269504088896340736f, M, st, nullptr);
      if (rc != WVN_OK && rc != WVN_ERR_ARG) return rc;   // WVN_ERR_ARG: not eligible (e.g. q / k / v beyond the 2 GB a buffer
      qkv_done = rc == WVN_OK;                            // descriptor spans) -- the separate kernels below, as proj_mlp does
    }
    if (!qkv_done) {
    if (l == 0) ln1_stats = false;
    bool qkv_lna = false;
    {
      GemmBf16Params e{};
      e.q = (bf16_t*)w.q; e.k = (bf16_t*)w.k; e.vt = (bf16_t*)w.v; e.heads = d.H; e.npad = d.npad; e.ntok = d.ntok; e.ntok_s = d.ntok_s;
      // bf16: the softmax scale is folded into q by the QKV epilogue (q leaves it as an exp2 argument) and attention takes
      // the running max as an MFMA operand (attention_bf16.hip, PRE)
      if (bf) e.q_scale = scale * 1.44269504088896340736f;
      // one fp16 plane each for k and v^T, q pre-scaled and in TWO fp16 planes (its rounding residue behind it): the fp16 attention kernel's operands
      if (mix) { e.qkv_f16 = 1; e.q_scale = scale * 1.44269504088896340736f; e.q_lo = qsplit ? lo(w.q, pl_qkv) : nullptr; }
      else if (x3) { e.q_lo = lo(w.q, pl_qkv); e.k_lo = lo(w.k, pl_qkv); e.vt_lo = lo(w.v, pl_qkv); }
      if (ln_fuse && ln1_stats) {   // norm1 on load, from the statistics the previous block's fc2 kernel left
        GemmBf16Params q = e;
        q.W = (const bf16_t*)L.qkv_w; q.W_lo = lo(L.qkv_w, (size_t)3 * d.D * d.D); q.ldw = d.D; q.bias = L.qkv_b; q.M = M; q.N = 3 * d.D; q.K = d.D;
        q.ln_x = w.x; q.ln_ldx = d.D; q.ln_stats = w.ln_stats; q.ln_g = L.ln1_g; q.ln_b = L.ln1_b;
        Span s(3, st);
        int rc = WVN_ERR_ARG;
        if (mx) {
          GemmBf16Params qm = q;
          qm.W = (const bf16_t*)L.qkv_w_mx; qm.W_lo = (const bf16_t*)L.qkv_w_mx + (size_t)3 * d.D * d.D;
          rc = wvn_gemm_a384_mx_launch(qm, EPI_QKV, st);
          if (rc != WVN_OK && rc != WVN_ERR_ARG) return rc;
        }
        if (rc != WVN_OK) rc = wvn_gemm_a384_x3_launch(q, EPI_QKV, st);
        if (rc != WVN_OK && rc != WVN_ERR_ARG) return rc;
        qkv_lna = rc == WVN_OK;
      }
      if (!qkv_lna) {
        { Span s(2, st); RET_IF(wvn_layernorm_launch(w.x, L.ln1_g, L.ln1_b, w.xn, f32 ? 0 : opk.fmt, d.D, nullptr, 0, M, d.D, 1e-6f, 0, d.ntok, d.ntok_s, st, x3 ? lo(w.xn, pl_xn) : nullptr)); }
        Span s(3, st);
        RET_IF(linear(w.xn, pl_xn, d.D, L.qkv_w, L.qkv_b, nullptr, 0, 0, M, 3 * d.D, d.D, EPI_QKV, nullptr, &e));
      }
    }
    ln1_stats = false;
    }
    {
      Span s(4, st);
      if (bf) RET_IF(opk.attention((const bf16_t*)w.q, (const bf16_t*)w.k, (const bf16_t*)w.v, (bf16_t*)w.xn, d.B, d.H, d.ntok, d.ntok_s, d.npad, 0.f, st, nullptr, nullptr, 0));
      else if (mix && mx) RET_IF(wvn_attention_bf16_launch_f16((const bf16_t*)w.q, (const bf16_t*)w.k, (const bf16_t*)w.v, (bf16_t*)w.xn, d.B, d.H, d.ntok, d.ntok_s, d.npad, 0.f, st, (bf16_t*)xn_l8, qsplit ? lo(w.q, pl_qkv) : nullptr, 2));
      else if (mix) RET_IF(wvn_attention_bf16_launch_f16((const bf16_t*)w.q, (const bf16_t*)w.k, (const bf16_t*)w.v, (bf16_t*)w.xn, d.B, d.H, d.ntok, d.ntok_s, d.npad, 0.f, st, lo(w.xn, pl_xn), qsplit ? lo(w.q, pl_qkv) : nullptr, attn_frag ? 1 : 0));
      else if (x3) RET_IF(wvn_attention_x3_launch((const bf16_t*)w.q, lo(w.q, pl_qkv), (const bf16_t*)w.k, lo(w.k, pl_qkv), (const bf16_t*)w.v, lo(w.v, pl_qkv), (bf16_t*)w.xn, lo(w.xn, pl_xn), d.B, d.H, d.ntok, d.ntok_s, d.npad, scale, st));
      else RET_IF(wvn_attention_f32_launch((const float*)w.q, (const float*)w.k, (const float*)w.v, (float*)w.xn, d.B, d.H, d.ntok, d.ntok_s, d.npad, scale, st));
    }
    if (mlp_fused && proj_in_mlp) {  // attention projection + LayerNorm 2 + fc1 + GELU + fc2 + both residual updates: one launch
      Span s(6, st);
      if (!L.fc2_w_fused) return WVN_ERR_ARG;
      // the resident form may also apply the NEXT block's norm1 and hand the rows to its QKV kernel as operand fragments
      const bool last = l + 1 == m->depth;
      bool hand_over = qkv_fused && !last && L.fc1_w_fused && m->layers[last ? l : l + 1].qkv_w_fused && !(m->flags & WVN_VIT_NO_LN_HANDOVER);
      const wvn_vit_layer& Ln = m->layers[last ? l : l + 1];
      int rc = WVN_ERR_ARG;
      if (hand_over)
        rc = opk.proj_mlp_fused((const bf16_t*)w.xn, d.D, (const bf16_t*)L.proj_w, L.proj_b, L.ls1, L.ln2_g, L.ln2_b, 1e-6f,
                                (const bf16_t*)L.fc1_w, L.fc1_b, (const bf16_t*)L.fc2_w_fused, L.fc2_b, L.ls2, w.x, d.D, M, d.F, st,
                                (const bf16_t*)L.fc1_w_fused, Ln.ln1_g, Ln.ln1_b, 1e-6f, (bf16_t*)w.hid);
      if (rc == WVN_ERR_ARG) {
        hand_over = false;
        rc = opk.proj_mlp_fused((const bf16_t*)w.xn, d.D, (const bf16_t*)L.proj_w, L.proj_b, L.ls1, L.ln2_g, L.ln2_b, 1e-6f,
                                (const bf16_t*)L.fc1_w, L.fc1_b, (const bf16_t*)L.fc2_w_fused, L.fc2_b, L.ls2, w.x, d.D, M, d.F, st,
                                (const bf16_t*)L.fc1_w_fused, nullptr, nullptr, 0.f, nullptr);
      }
      pre_qkv = rc == WVN_OK && hand_over;
      if (rc == WVN_OK) continue;
      if (rc != WVN_ERR_ARG) return rc;   // (WVN_ERR_ARG: not eligible -- separate kernels)
    }
    bool ln2_stats = false;   // w.ln_stats holds the statistics of w.x for this block's norm2
    if (mx) {   // the attention output arrived as MX operand planes: the projection on the MX row-panel kernel
      GemmBf16Params pp{};
      pp.A = (const bf16_t*)w.xn; pp.A_lo = (const bf16_t*)xn_l8; pp.lda = d.D; pp.W = (const bf16_t*)L.proj_w_mx; pp.ldw = d.D; pp.bias = L.proj_b; pp.ls = L.ls1;
      pp.C = w.x; pp.ldc = d.D; pp.M = M; pp.N = d.D; pp.K = d.D;
      pp.ln_stats_out = w.ln_stats; pp.ln_eps = 1e-6f;
      Span s(5, st);
      RET_IF(wvn_gemm_n384_mx_launch(pp, EPI_RESID_F32, st));
      ln2_stats = true;
    } else if (attn_frag) {   // the attention output arrived as operand fragments: the projection on the fragment form of the row-panel kernel
      GemmBf16Params pp{};
      pp.A = (const bf16_t*)w.xn; pp.A_lo = lo(w.xn, pl_xn); pp.lda = d.D; pp.W = (const bf16_t*)L.proj_w_frag; pp.ldw = d.D; pp.bias = L.proj_b; pp.ls = L.ls1;
      pp.C = w.x; pp.ldc = d.D; pp.M = M; pp.N = d.D; pp.K = d.D;
      if (ln_fuse) { pp.ln_stats_out = w.ln_stats; pp.ln_eps = 1e-6f; }
      Span s(5, st);
      RET_IF(wvn_gemm_n384_x3_frag_launch(pp, EPI_RESID_F32, st));
      ln2_stats = ln_fuse;
    } else {
      GemmBf16Params e{};
      if (ln_fuse) { e.ln_stats_out = w.ln_stats; e.ln_eps = 1e-6f; }
      stats_written = false;
      Span s(5, st);
      RET_IF(linear(w.xn, pl_xn, d.D, L.proj_w, L.proj_b, w.x, 0, d.D, M, d.D, d.D, EPI_RESID_F32, L.ls1, ln_fuse ? &e : nullptr));
      ln2_stats = ln_fuse && stats_written;
    }
    bool ln2_done = false;
    if (mlp_fused) {  // LayerNorm 2 + fc1 + GELU + fc2 + residual: one launch, no xn / hid round trip
      Span s(6, st);
      if (!L.fc2_w_fused) return WVN_ERR_ARG;
      RET_IF(opk.mlp_fused(nullptr, 0, L.ln2_g, L.ln2_b, 1e-6f, (const bf16_t*)L.fc1_w, L.fc1_b, (const bf16_t*)L.fc2_w_fused, L.fc2_b,
                                  L.ls2, w.x, d.D, M, d.F, st));
      continue;
    }
    if (mx && ln2_stats) {   // LayerNorm-on-load fc1 + GELU -> MX operand planes -> MX row-panel fc2 (+ the next block's LayerNorm statistics)
      GemmBf16Params p1{};
      p1.W = (const bf16_t*)L.fc1_w_mx; p1.W_lo = (const bf16_t*)L.fc1_w_mx + (size_t)d.F * d.D; p1.ldw = d.D; p1.bias = L.fc1_b;
      p1.C = w.hid; p1.C_lo = hid_l8; p1.ldc = d.F; p1.M = M; p1.N = d.F; p1.K = d.D;
      p1.ln_x = w.x; p1.ln_ldx = d.D; p1.ln_stats = w.ln_stats; p1.ln_g = L.ln2_g; p1.ln_b = L.ln2_b;
      { Span s(6, st); RET_IF(wvn_gemm_a384_mx_launch(p1, EPI_GELU_FRAG, st)); }
      GemmBf16Params p2{};
      p2.A = (const bf16_t*)w.hid; p2.A_lo = (const bf16_t*)hid_l8; p2.lda = d.F; p2.W = (const bf16_t*)L.fc2_w_mx; p2.ldw = d.F; p2.bias = L.fc2_b;
      p2.ls = L.ls2; p2.C = w.x; p2.ldc = d.D; p2.M = M; p2.N = d.D; p2.K = d.F;
      const bool want = l + 1 < m->depth;   // the next block's norm1
      if (want) { p2.ln_stats_out = w.ln_stats; p2.ln_eps = 1e-6f; }
      Span s(7, st);
      RET_IF(wvn_gemm_n384_mx_launch(p2, EPI_RESID_F32, st));
      ln1_stats = want;
      continue;
    }
    if (x3_fast && L.fc2_w_fused && !(m->flags & WVN_VIT_X3_NO_FRAG_MLP)) {
      // the split-operand MLP with the hidden activation handed over FRAGMENT-MAJOR: fc1 (gemm_a384_x3, EPI_GELU_FRAG) writes the
      // MFMA operand fragments of fc2 straight from its accumulators, fc2 (gemm_n384_x3, AFRAG) fetches them with one coalesced load
      // per lane and plane -- no LDS transpose on either side, every access a contiguous kilobyte
      const size_t pl_frag = (size_t)((d.M + 31) / 32 * 32) * d.F;
      GemmBf16Params p1{};
      p1.W = (const bf16_t*)L.fc1_w; p1.W_lo = lo(L.fc1_w, (size_t)d.F * d.D);
      p1.ldw = d.D; p1.bias = L.fc1_b; p1.C = w.hid; p1.C_lo = lo(w.hid, pl_frag); p1.ldc = d.F; p1.M = M; p1.N = d.F; p1.K = d.D;
      int rc = WVN_ERR_ARG;
      if (ln2_stats) {   // norm2 on load, from the statistics the projection kernel left
        GemmBf16Params pf = p1;
        pf.ln_x = w.x; pf.ln_ldx = d.D; pf.ln_stats = w.ln_stats; pf.ln_g = L.ln2_g; pf.ln_b = L.ln2_b;
        Span s(6, st);
        rc = wvn_gemm_a384_x3_launch(pf, EPI_GELU_FRAG, st);
        if (rc != WVN_OK && rc != WVN_ERR_ARG) return rc;
      }
      if (rc != WVN_OK) {
        { Span s(2, st); RET_IF(wvn_layernorm_launch(w.x, L.ln2_g, L.ln2_b, w.xn, opk.fmt, d.D, nullptr, 0, M, d.D, 1e-6f, 0, d.ntok, d.ntok_s, st, lo(w.xn, pl_xn))); }
        ln2_done = true;
        p1.A = (const bf16_t*)w.xn; p1.A_lo = lo(w.xn, pl_xn); p1.lda = d.D;
        Span s(6, st);
        rc = wvn_gemm_a384_x3_launch(p1, EPI_GELU_FRAG, st);
      }
      if (rc == WVN_OK) {
        GemmBf16Params p2{};
        p2.A = (const bf16_t*)w.hid; p2.A_lo = lo(w.hid, pl_frag); p2.lda = d.F; p2.W = (const bf16_t*)L.fc2_w_fused; p2.ldw = d.F; p2.bias = L.fc2_b;
        p2.ls = L.ls2; p2.C = w.x; p2.ldc = d.D; p2.M = M; p2.N = d.D; p2.K = d.F;
        const bool want = ln_fuse && l + 1 < m->depth;   // the next block's norm1
        if (want) { p2.ln_stats_out = w.ln_stats; p2.ln_eps = 1e-6f; }
        Span s(7, st);
        RET_IF(wvn_gemm_n384_x3_frag_launch(p2, EPI_RESID_F32, st));
        ln1_stats = want;
        continue;
      }
      if (rc != WVN_ERR_ARG) return rc;
    }
    if (!mlp_fused && !ln2_done) { Span s(2, st); RET_IF(wvn_layernorm_launch(w.x, L.ln2_g, L.ln2_b, w.xn, f32 ? 0 : opk.fmt, d.D, nullptr, 0, M, d.D, 1e-6f, 0, d.ntok, d.ntok_s, st, x3 ? lo(w.xn, pl_xn) : nullptr)); }
    { Span s(6, st); RET_IF(linear(w.xn, pl_xn, d.D, L.fc1_w, L.fc1_b, w.hid, pl_hid, d.F, M, d.F, d.D, EPI_GELU_BF16, nullptr, nullptr)); }
    { Span s(7, st); RET_IF(linear(w.hid, pl_hid, d.F, L.fc2_w, L.fc2_b, w.x, 0, d.D, M, d.D, d.F, EPI_RESID_F32, L.ls2, nullptr)); }
  }
  {
    Span s(2, st);
    RET_IF(wvn_layernorm_launch(w.x, m->norm_g, m->norm_b, tokens_lowp, bf ? opk.fmt : 0, ld_lowp, tokens_f32, d.D, Mp, d.D, 1e-6f, 1, d.ntok, d.ntok_s, st));
  }
  return WVN_OK;
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------
int wvn_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                  int K, int epi, void* stream) {
  if (epi < 0 || epi > EPI_ACCUM_F32 || !C) return WVN_ERR_ARG;
  GemmBf16Params p{};
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  return wvn_gemm_bf16_launch(p, epi, (hipStream_t)stream);
}

int wvn_gemm_f16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                 int K, int epi, void* stream) {
  if (epi < 0 || epi > EPI_ACCUM_F32 || !C) return WVN_ERR_ARG;
  GemmBf16Params p{};
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  return wvn_gemm_bf16_launch_f16(p, epi, (hipStream_t)stream);
}
int wvn_qkv_fused_f16(const float* x, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const void* W, const float* bias,
                      void* q, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, int M, void* stream) {
  return wvn_qkv_fused_launch_f16(x, ldx, ln_g, ln_b, ln_eps, (const bf16_t*)W, bias, (bf16_t*)q, (bf16_t*)k, (bf16_t*)vt, heads, npad,
                                  ntok_s, q_scale, M, (hipStream_t)stream, nullptr);
}
int wvn_qkv_prenorm_f16(const void* xn_frag, const void* Wperm, const float* bias, void* q, void* k, void* vt, int heads, int npad,
                        int ntok_s, float q_scale, int M, void* stream) {
  if (!xn_frag) return WVN_ERR_ARG;
  return wvn_qkv_fused_launch_f16(nullptr, 0, nullptr, nullptr, 0.f, (const bf16_t*)Wperm, bias, (bf16_t*)q, (bf16_t*)k, (bf16_t*)vt, heads,
                                  npad, ntok_s, q_scale, M, (hipStream_t)stream, (const bf16_t*)xn_frag);
}
int wvn_proj_mlp_fused_f16(const void* attn, int lda, const void* Wp, const float* bp, const float* ls1, const float* ln_g,
                           const float* ln_b, float ln_eps, const void* W1, const float* b1, const void* W2p, const float* b2,
                           const float* ls2, float* x, int ldx, int M, int F, void* stream) {
  return wvn_proj_mlp_fused_launch_f16((const bf16_t*)attn, lda, (const bf16_t*)Wp, bp, ls1, ln_g, ln_b, ln_eps, (const bf16_t*)W1, b1,
                                       (const bf16_t*)W2p, b2, ls2, x, ldx, M, F, (hipStream_t)stream, nullptr, nullptr, nullptr, 0.f, nullptr);
}
int wvn_proj_mlp_resident_f16(const void* attn, int lda, const void* Wp, const float* bp, const float* ln_g, const float* ln_b,
                              float ln_eps, const void* W1p, const float* b1, const void* W2p, const float* b2, float* x, int ldx,
                              int M, int F, const float* next_ln_g, const float* next_ln_b, float next_ln_eps, void* xn_next,
                              void* stream) {
  if (!W1p) return WVN_ERR_ARG;
  return wvn_proj_mlp_fused_launch_f16((const bf16_t*)attn, lda, (const bf16_t*)Wp, bp, nullptr, ln_g, ln_b, ln_eps, nullptr, b1,
                                       (const bf16_t*)W2p, b2, nullptr, x, ldx, M, F, (hipStream_t)stream, (const bf16_t*)W1p, next_ln_g, next_ln_b,
                                   next_ln_eps, (bf16_t*)xn_next);
}
int wvn_mlp_fused_f16(const void* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const void* W1, const float* b1,
                      const void* W2p, const float* b2, const float* ls, float* x, int ldx, int M, int F, void* stream) {
  return wvn_mlp_fused_launch_f16((const bf16_t*)xn, lda, ln_g, ln_b, ln_eps, (const bf16_t*)W1, b1, (const bf16_t*)W2p, b2, ls, x,
                                  ldx, M, F, (hipStream_t)stream);
}
int wvn_attention_f16(const void* q, const void* k, const void* vt, void* out, int B, int heads, int ntok, int npad, float scale,
                      void* stream) {
  return wvn_attention_bf16_launch_f16((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, B, heads, ntok, ntok,
                                       npad, scale, (hipStream_t)stream);
}

int wvn_qkv_fused(const float* x, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const void* W, const float* bias,
                  void* q, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, int M, void* stream) {
  return wvn_qkv_fused_launch(x, ldx, ln_g, ln_b, ln_eps, (const bf16_t*)W, bias, (bf16_t*)q, (bf16_t*)k, (bf16_t*)vt, heads, npad, ntok_s,
                              q_scale, M, (hipStream_t)stream, nullptr);
}
int wvn_qkv_prenorm(const void* xn_frag, const void* Wperm, const float* bias, void* q, void* k, void* vt, int heads, int npad, int ntok_s,
                    float q_scale, int M, void* stream) {
  if (!xn_frag) return WVN_ERR_ARG;
  return wvn_qkv_fused_launch(nullptr, 0, nullptr, nullptr, 0.f, (const bf16_t*)Wperm, bias, (bf16_t*)q, (bf16_t*)k, (bf16_t*)vt, heads, npad,
                              ntok_s, q_scale, M, (hipStream_t)stream, (const bf16_t*)xn_frag);
}

int wvn_proj_mlp_fused(const void* attn, int lda, const void* Wp, const float* bp, const float* ls1, const float* ln_g,
                       const float* ln_b, float ln_eps, const void* W1, const float* b1, const void* W2p, const float* b2,
                       const float* ls2, float* x, int ldx, int M, int F, void* stream) {
  return wvn_proj_mlp_fused_launch((const bf16_t*)attn, lda, (const bf16_t*)Wp, bp, ls1, ln_g, ln_b, ln_eps, (const bf16_t*)W1, b1,
                                   (const bf16_t*)W2p, b2, ls2, x, ldx, M, F, (hipStream_t)stream, nullptr, nullptr, nullptr, 0.f, nullptr);
}
int wvn_proj_mlp_resident(const void* attn, int lda, const void* Wp, const float* bp, const float* ln_g, const float* ln_b,
                          float ln_eps, const void* W1p, const float* b1, const void* W2p, const float* b2, float* x, int ldx, int M,
                          int F, const float* next_ln_g, const float* next_ln_b, float next_ln_eps, void* xn_next, void* stream) {
  if (!W1p) return WVN_ERR_ARG;
  return wvn_proj_mlp_fused_launch((const bf16_t*)attn, lda, (const bf16_t*)Wp, bp, nullptr, ln_g, ln_b, ln_eps, nullptr, b1,
                                   (const bf16_t*)W2p, b2, nullptr, x, ldx, M, F, (hipStream_t)stream, (const bf16_t*)W1p, next_ln_g, next_ln_b,
                                   next_ln_eps, (bf16_t*)xn_next);
}

int wvn_mlp_fused(const void* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const void* W1, const float* b1,
                  const void* W2p, const float* b2, const float* ls, float* x, int ldx, int M, int F, void* stream) {
  return wvn_mlp_fused_launch((const bf16_t*)xn, lda, ln_g, ln_b, ln_eps, (const bf16_t*)W1, b1, (const bf16_t*)W2p, b2, ls, x, ldx,
                              M, F, (hipStream_t)stream);
}

int wvn_gemm_x3(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo, int ldw, const float* bias,
                void* C, void* C_lo, int ldc, int M, int N, int K, int epi, void* stream) {
  if (epi < 0 || epi > EPI_ACCUM_F32 || !C) return WVN_ERR_ARG;
  GemmBf16Params p{};
  p.A = (const bf16_t*)A_hi; p.A_lo = (const bf16_t*)A_lo; p.lda = lda; p.W = (const bf16_t*)W_hi; p.W_lo = (const bf16_t*)W_lo;
  p.ldw = ldw; p.bias = bias; p.C = C; p.C_lo = C_lo; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  return wvn_gemm_x3_launch(p, epi, (hipStream_t)stream);
}
int wvn_quantize_rows_fp8(const void* src, int src_is_bf16, int lds, void* q, int ldq, float* scale, int rows, int cols,
                          void* stream) {
  return wvn_quantize_rows_fp8_launch(src, src_is_bf16, lds, (unsigned char*)q, ldq, scale, rows, cols, (hipStream_t)stream);
}
int wvn_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* q, int ldq, float* scale, int rows, int D,
                      float eps, void* stream) {
  return wvn_layernorm_fp8_launch(x, gamma, beta, (unsigned char*)q, ldq, scale, rows, D, eps, (hipStream_t)stream);
}
int wvn_gemm_fp8(const void* A_q, int lda, const void* W_q, int ldw, const float* sa, const float* sw, const float* bias,
                 void* C, int ldc, int M, int N, int K, int epi, void* stream) {
  if (epi != EPI_BF16 && epi != EPI_GELU_BF16 && epi != EPI_F32 && epi != EPI_RESID_F32) return WVN_ERR_ARG;
  GemmFp8Params p{};
  p.A = (const unsigned char*)A_q; p.lda = lda; p.W = (const unsigned char*)W_q; p.ldw = ldw; p.sa = sa; p.sw = sw; p.bias = bias;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  return wvn_gemm_fp8_launch(p, epi, (hipStream_t)stream);
}
int wvn_gemm_a768_fp8(const void* A_q, int lda, const void* W_packed, const float* sa, const float* sw, const float* bias, const float* ls, void* C, int ldc,
                      int M, int N, int epi, void* q, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, void* stream) {
  GemmFp8Params p{};
  p.A = (const unsigned char*)A_q; p.lda = lda; p.sa = sa; p.sw = sw; p.bias = bias; p.ls = ls; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = 768;
  p.q = (bf16_t*)q; p.k = (bf16_t*)k; p.vt = (bf16_t*)vt; p.heads = heads; p.npad = npad; p.ntok_s = ntok_s; p.q_scale = q_scale;
  if (epi == EPI_GELU_MX8) p.c_scales = (unsigned char*)q;   // (epi 9: C = e4m3 [M][ldc], q = its E8M0 block scales [M][N / 32])
  return wvn_gemm_a768_fp8_launch(p, W_packed, epi, (hipStream_t)stream);
}
int wvn_gemm_fp8_mx(const void* A_q, int lda, const void* a_scales, const void* W_q, int ldw, const float* sw, const float* bias, const float* ls,
                    void* C, int ldc, int M, int N, int K, int epi, void* stream) {
  if (!a_scales || (epi != EPI_F32 && epi != EPI_RESID_F32)) return WVN_ERR_ARG;
  GemmFp8Params p{};
  p.A = (const unsigned char*)A_q; p.lda = lda; p.a_scales = (const unsigned char*)a_scales; p.W = (const unsigned char*)W_q; p.ldw = ldw; p.sw = sw; p.bias = bias;
  p.ls = ls; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  return wvn_gemm_fp8_launch(p, epi, (hipStream_t)stream);
}
int wvn_split_planes(const float* src, int lds, void* hi, void* lo, int ldd, int rows, int cols, void* stream) {
  return wvn_split_planes_launch(src, lds, (bf16_t*)hi, (bf16_t*)lo, ldd, rows, cols, (hipStream_t)stream);
}
int wvn_attention_x3(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                     const void* vt_lo, void* out_hi, void* out_lo, int B, int heads, int ntok, int npad, float scale,
                     void* stream) {
  return wvn_attention_x3_launch((const bf16_t*)q_hi, (const bf16_t*)q_lo, (const bf16_t*)k_hi, (const bf16_t*)k_lo,
                                 (const bf16_t*)vt_hi, (const bf16_t*)vt_lo, (bf16_t*)out_hi, (bf16_t*)out_lo, B, heads, ntok,
                                 ntok, npad, scale, (hipStream_t)stream);
}

int wvn_debug_f16_saturate(const float* in, void* out_f16, int n, void* stream) {
  return (in && out_f16 && n > 0) ? wvn_f16_saturate_probe_launch(in, (uint16_t*)out_f16, n, (hipStream_t)stream) : WVN_ERR_ARG;
}
int wvn_debug_attention_timing(long long* dbg) { wvn_attention_bf16_set_debug(dbg); return WVN_OK; }
int wvn_debug_mlp_fused_timing(long long* dbg) { g_mlp_fused_dbg = dbg; return WVN_OK; }
int wvn_debug_qkv_fused_timing(long long* dbg) { g_qkv_fused_dbg = dbg; return WVN_OK; }
int wvn_stream_create_cu_mask(void** stream, const unsigned int* mask, int words) {
  if (!stream || !mask || words <= 0) return WVN_ERR_ARG;
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) return (int)e;
  *stream = (void*)st;
  return WVN_OK;
}
int wvn_stream_destroy(void* stream) {
  if (!stream) return WVN_ERR_ARG;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? WVN_OK : (int)e;
}

int wvn_debug_gemm_a384_x3(const void* A, const void* A_lo, int lda, const void* W, const void* W_lo, const float* bias, void* C,
                           void* C_lo, int ldc, int M, int N, int epi, long long* dbg, void* stream) {
  if (epi != EPI_GELU_BF16 && epi != EPI_RESID_F32) return WVN_ERR_ARG;
  GemmBf16Params p{};
  p.A = (const bf16_t*)A; p.A_lo = (const bf16_t*)A_lo; p.lda = lda; p.W = (const bf16_t*)W; p.W_lo = (const bf16_t*)W_lo; p.ldw = 384;
  p.bias = bias; p.C = C; p.C_lo = C_lo; p.ldc = ldc; p.M = M; p.N = N; p.K = 384; p.dbg = dbg;
  return wvn_gemm_a384_x3_launch(p, epi, (hipStream_t)stream);
}
int wvn_debug_gemm_n384_x3(const void* A, const void* A_lo, int lda, const void* W, const void* W_lo, const float* bias, const float* ls,
                           float* C, int ldc, int M, int K, long long* dbg, void* stream) {
  GemmBf16Params p{};
  p.A = (const bf16_t*)A; p.A_lo = (const bf16_t*)A_lo; p.lda = lda; p.W = (const bf16_t*)W; p.W_lo = (const bf16_t*)W_lo; p.ldw = K;
  p.bias = bias; p.ls = ls; p.C = C; p.ldc = ldc; p.M = M; p.N = 384; p.K = K; p.dbg = dbg;
  return wvn_gemm_n384_x3_launch(p, EPI_RESID_F32, (hipStream_t)stream);
}
int wvn_debug_mlp_x3_frag(const void* xn, const void* xn_lo, const void* W1, const void* W1_lo, const float* b1, void* hid, void* hid_lo,
                          const void* W2p, const float* b2, float* x, int M, int F, long long* dbg1, long long* dbg2, void* stream) {
  GemmBf16Params p1{};
  p1.A = (const bf16_t*)xn; p1.A_lo = (const bf16_t*)xn_lo; p1.lda = 384; p1.W = (const bf16_t*)W1; p1.W_lo = (const bf16_t*)W1_lo; p1.ldw = 384;
  p1.bias = b1; p1.C = hid; p1.C_lo = hid_lo; p1.ldc = F; p1.M = M; p1.N = F; p1.K = 384; p1.dbg = dbg1;
  RET_IF(wvn_gemm_a384_x3_launch(p1, EPI_GELU_FRAG, (hipStream_t)stream));
  GemmBf16Params p2{};
  p2.A = (const bf16_t*)hid; p2.A_lo = (const bf16_t*)hid_lo; p2.lda = F; p2.W = (const bf16_t*)W2p; p2.ldw = F; p2.bias = b2; p2.C = x; p2.ldc = 384;
  p2.M = M; p2.N = 384; p2.K = F; p2.dbg = dbg2;
  return wvn_gemm_n384_x3_frag_launch(p2, EPI_RESID_F32, (hipStream_t)stream);
}
int wvn_debug_gemm_n384_mx(const void* A_h, const void* A_l8, const void* A_h8, const void* Wp, const float* bias, const float* ls, float* C,
                           int ldc, int M, int K, long long* dbg, void* stream) {
  GemmBf16Params p{};
  p.A = (const bf16_t*)A_h; p.A_lo = (const bf16_t*)A_l8; p.A_h8 = A_h8; p.lda = K; p.W = (const bf16_t*)Wp; p.ldw = K; p.bias = bias; p.ls = ls;
  p.C = C; p.ldc = ldc; p.M = M; p.N = 384; p.K = K; p.dbg = dbg;
  return wvn_gemm_n384_mx_launch(p, EPI_RESID_F32, (hipStream_t)stream);
}
int wvn_debug_mlp_mx(const float* x, int ldx, const float* ln_stats, const float* ln_g, const float* ln_b, const void* W1p, const float* b1,
                     void* hid_h, void* hid_l8, void* hid_h8, const void* W2p, const float* b2, float* xout, int M, int F, long long* dbg1,
                     long long* dbg2, void* stream) {
  GemmBf16Params p1{};
  p1.W = (const bf16_t*)W1p; p1.W_lo = (const bf16_t*)W1p + (size_t)F * 384; p1.ldw = 384; p1.bias = b1;
  p1.C = hid_h; p1.C_lo = hid_l8; p1.C_h8 = hid_h8; p1.ldc = F; p1.M = M; p1.N = F; p1.K = 384; p1.dbg = dbg1;
  p1.ln_x = x; p1.ln_ldx = ldx; p1.ln_stats = ln_stats; p1.ln_g = ln_g; p1.ln_b = ln_b;
  RET_IF(wvn_gemm_a384_mx_launch(p1, EPI_GELU_FRAG, (hipStream_t)stream));
  if (!W2p) return WVN_OK;
  GemmBf16Params p2{};
  p2.A = (const bf16_t*)hid_h; p2.A_lo = (const bf16_t*)hid_l8; p2.A_h8 = hid_h8; p2.lda = F; p2.W = (const bf16_t*)W2p; p2.ldw = F; p2.bias = b2;
  p2.C = xout; p2.ldc = 384; p2.M = M; p2.N = 384; p2.K = F; p2.dbg = dbg2;
  return wvn_gemm_n384_mx_launch(p2, EPI_RESID_F32, (hipStream_t)stream);
}
int wvn_debug_qkv_mx(const float* x, int ldx, const float* ln_stats, const float* ln_g, const float* ln_b, const void* Wp, const float* bias, void* q,
                     void* q_lo, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, int M, long long* dbg, void* stream) {
  GemmBf16Params p{};
  p.W = (const bf16_t*)Wp; p.W_lo = (const bf16_t*)Wp + (size_t)3 * heads * 64 * 384; p.ldw = 384; p.bias = bias; p.M = M; p.N = 3 * heads * 64; p.K = 384;
  p.q = (bf16_t*)q; p.q_lo = (bf16_t*)q_lo; p.k = (bf16_t*)k; p.vt = (bf16_t*)vt; p.heads = heads; p.npad = npad; p.ntok_s = ntok_s; p.q_scale = q_scale;
  p.qkv_f16 = 1; p.dbg = dbg;
  p.ln_x = x; p.ln_ldx = ldx; p.ln_stats = ln_stats; p.ln_g = ln_g; p.ln_b = ln_b;
  return wvn_gemm_a384_mx_launch(p, EPI_QKV, (hipStream_t)stream);
}
int wvn_debug_n384_pair(int on) {
  if (on >= 32) { wvn_gemm_a384_mx_set_form(on - 32); return WVN_OK; }   // 33 / 34: form 1 / 2 of the A-stationary MX kernel
  wvn_gemm_n384_x3_set_pair(on);
  return WVN_OK;
}
int wvn_debug_kmeans_screen_stats(unsigned long long* out, int reset) { return out ? wvn_kmeans_pixels_screen_stats(out, reset) : WVN_ERR_ARG; }
int wvn_debug_kmeans_assign_form(int form) { wvn_kmeans_pixels_set_assign_form(form); return WVN_OK; }
int wvn_debug_attention_variant(int v) { wvn_attention_bf16_set_variant(v); wvn_attention_bf16_set_variant_f16(v); return WVN_OK; }

int wvn_debug_gemm_bf16_timed(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                              int N, int K, int epi, long long* dbg, void* stream) {
  if (epi < 0 || epi > EPI_ACCUM_F32 || !C) return WVN_ERR_ARG;
  GemmBf16Params p{};
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.dbg = dbg;
  return wvn_gemm_bf16_launch(p, epi, (hipStream_t)stream);
}

int wvn_debug_gemm_n384(const void* A, int lda, const void* W, int ldw, const float* bias, float* C, int ldc, int M, int K,
                        void* stream) {
  GemmBf16Params p{};
  p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = 384; p.K = K;
  return wvn_gemm_n384_launch(p, EPI_RESID_F32, (hipStream_t)stream, nullptr, 1);
}

int wvn_gemm_f32(const float* A, int lda, int transA, const float* B, int ldb, int transB, const float* bias, float* C,
                 int ldc, int M, int N, int K, int epi, const float* mask, int ldmask, void* stream) {
  if (epi < 0 || epi > F32_EPI_RELUMASK) return WVN_ERR_ARG;
  GemmF32Params p{};
  p.A = A; p.lda = lda; p.transA = transA; p.B = B; p.ldb = ldb; p.transB = transB; p.bias = bias; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.batch = 1; p.splitk = 1; p.mask = mask; p.ldmask = ldmask;
  return wvn_gemm_f32_launch(p, epi, (hipStream_t)stream);
}

int wvn_layernorm(const float* x, const float* gamma, const float* beta, void* y, int y_is_bf16, int rows, int D,
                  float eps, void* stream) {
  if (!y) return WVN_ERR_ARG;
  return wvn_layernorm_launch(x, gamma, beta, y, y_is_bf16 ? 1 : 0, D, nullptr, 0, rows, D, eps, 0, 0, 0, (hipStream_t)stream);
}

int wvn_attention_bf16(const void* q, const void* k, const void* vt, void* out, int B, int heads, int ntok, int npad,
                       float scale, void* stream) {
  return wvn_attention_bf16_launch((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, B, heads, ntok,
                                   ntok, npad, scale, (hipStream_t)stream);
}
int wvn_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int heads, int ntok, int npad,
                      float scale, void* stream) {
  return wvn_attention_f32_launch(q, k, v, out, B, heads, ntok, ntok, npad, scale, (hipStream_t)stream);
}
int wvn_patchify(const float* img, void* patches, int out_is_bf16, int B, int S, int P, void* stream) {
  return wvn_patchify_launch(img, 0, patches, nullptr, out_is_bf16 ? 1 : 0, 0, B, S, P, (hipStream_t)stream);
}

int wvn_patchify_u8(const unsigned char* img, void* patches_bf16, int B, int S, int P, void* stream) {
  return wvn_patchify_launch(img, 1, patches_bf16, nullptr, 1, 0, B, S, P, (hipStream_t)stream);
}
int wvn_cast_f32_to_bf16(const float* src, void* dst, long long n, void* stream) {
  if (!src || !dst || n <= 0 || n > 0x7fffffffll) return WVN_ERR_ARG;
  return wvn_cast_f32_bf16_launch(src, (int)n, (bf16_t*)dst, (int)n, 1, (int)n, (hipStream_t)stream);
}
int wvn_cast_rows(const float* src, int lds, void* dst, int ldd, int rows, int cols, int to_f16, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return WVN_ERR_ARG;
  return wvn_cast_f32_bf16_launch(src, lds, (bf16_t*)dst, ldd, rows, cols, (hipStream_t)stream, to_f16 != 0);
}
int wvn_upsample_bilinear(const float* tokens, float* dense, int B, int G, int D, int H, void* stream) {
  return wvn_upsample_bilinear_launch(tokens, dense, B, G, D, H, (hipStream_t)stream);
}
int wvn_upsample_nearest_i32(const int* labels, int* out, int B, int G, int H, void* stream) {
  if (!labels || !out) return WVN_ERR_ARG;
  return wvn_upsample_nearest_i32_launch(labels, out, B, G, H, (hipStream_t)stream);
}
int wvn_segpool_bilinear_mean(const int* seg, const float* tokens, int ld, float* feat, void* scratch_w,
                              int* scratch_cnt, int B, int H, int W, int G, int S, int D, void* stream) {
  return wvn_segpool_launch(seg, tokens, ld, feat, scratch_w, scratch_cnt, B, H, W, G, S, D, (hipStream_t)stream);
}
int wvn_segpool_patch_labels(const int* labels, const float* tokens, int ld, const float* wy, const float* wx,
                             float* feat, int B, int G, int S, int D, void* stream) {
  return wvn_segpool_patch_launch(labels, tokens, ld, wy, wx, feat, B, G, S, D, (hipStream_t)stream);
}
size_t wvn_segmean_scratch_bytes(int B, int P, int S, int D) {
  return (B > 0 && P > 0 && S > 0 && D > 0) ? wvn_segmean_scratch_bytes_impl(B, P, S, D) : 0;
}
int wvn_segmean_tokens(const int* seg, const float* tokens, float* feat, int* scratch_cnt, void* scratch, size_t scratch_bytes,
                       int B, int P, int S, int D, void* stream) {
  return wvn_segmean_tokens_launch(seg, tokens, feat, scratch_cnt, scratch, scratch_bytes, B, P, S, D, (hipStream_t)stream);
}
int wvn_label_pool(const float* mask, int C, const int* seg, float* signal, unsigned char* valid, void* scratch_sum,
                   int* scratch_cnt, int H, int W, int S, void* stream) {
  return wvn_label_pool_launch(mask, C, seg, signal, valid, scratch_sum, scratch_cnt, H, W, S, (hipStream_t)stream);
}
int wvn_label_pool_batched(const wvn_label_pool_node* nodes_dev, int n, int C, int H, int W, int Smax, void* scratch_sum,
                           int* scratch_cnt, void* stream) {
  return wvn_label_pool_batched_launch(nodes_dev, n, C, H, W, Smax, (long long*)scratch_sum, scratch_cnt, (hipStream_t)stream);
}
int wvn_project_render_fmin(const wvn_render_node* nodes_dev, int n, const float* points, int points_batched, int npts, int C,
                            int H, int W, const float* value_dev, float value, void* stream) {
  return wvn_project_render_fmin_launch(nodes_dev, n, points, points_batched, npts, C, H, W, value_dev, value, (hipStream_t)stream);
}
size_t wvn_wire_bytes(int H, int W, int S, int D) { return (H > 0 && W > 0 && S > 0 && D > 0) ? wvn_wire_bytes_impl(H, W, S, D) : 0; }
int wvn_wire_pack(const void* seg, int seg_is_i64, const float* feat, int ldf, void* out, int H, int W, int S, int D, void* stream) {
  return wvn_wire_pack_launch(seg, seg_is_i64, feat, ldf, out, H, W, S, D, (hipStream_t)stream);
}
int wvn_wire_unpack(const void* in, long long* seg_i64, int* seg_i32, float* feat, int H, int W, int S, int D, void* stream) {
  return wvn_wire_unpack_launch(in, seg_i64, seg_i32, feat, H, W, S, D, (hipStream_t)stream);
}
int wvn_slic_num_clusters(int H, int W, int num_components) {
  return (H > 0 && W > 0 && num_components > 0) ? wvn_slic_num_clusters_impl(H, W, num_components) : 0;
}
size_t wvn_slic_scratch_bytes(int H, int W, int num_components) {
  return (H > 0 && W > 0 && num_components > 0) ? wvn_slic_scratch_bytes_impl(H, W, num_components) : 0;
}
int wvn_slic(const void* img, int img_is_u8, int H, int W, int num_components, float compactness, int iters, const int* lut_lin,
             const int* lut_f, int* labels, void* scratch, size_t scratch_bytes, void* stream) {
  return wvn_slic_launch(img, img_is_u8, H, W, num_components, compactness, iters, lut_lin, lut_f, labels, scratch, scratch_bytes,
                         (hipStream_t)stream);
}
int wvn_seg_centers(const int* seg, float* centers, void* scratch, int H, int W, int S, void* stream) {
  return wvn_centers_launch(seg, centers, (unsigned long long*)scratch, H, W, S, (hipStream_t)stream);
}
int wvn_seg_adjacency(const int* seg, long long* edges, int* count, unsigned char* scratch_bitmap, int H, int W, int S,
                      int max_edges, void* stream) {
  return wvn_adjacency_launch(seg, edges, count, scratch_bitmap, H, W, S, max_edges, (hipStream_t)stream);
}
int wvn_normalize_rows(const float* code, int ldc, float* xn, int rows, int C, void* stream) {
  return wvn_normalize_rows_launch(code, ldc, xn, rows, C, (hipStream_t)stream);
}
int wvn_argmax_rows(const float* x, int ld, int rows, int cols, int* out, void* stream) {
  return wvn_argmax_rows_launch(x, ld, rows, cols, out, (hipStream_t)stream);
}
size_t wvn_kmeans_scratch_bytes(int B, int P, int C, int K) {
  return (B > 0 && P > 0 && C > 0 && K > 0) ? wvn_kmeans_scratch_floats(B, P, C, K) * sizeof(float) : 0;
}
size_t wvn_kmeans_pixels_scratch_bytes(int B, int G, int H, int C, int K) {
  return (B > 0 && G > 0 && H > 0 && C > 0 && K > 0) ? wvn_kmeans_pixels_scratch_floats(B, G, H, C, K) * sizeof(float) : 0;
}
int wvn_kmeans_cosine_pixels(const float* code, int* labels, int* nseg, void* scratch, int B, int G, int H, int C, int K, int iters,
                             int relabel, void* stream) {
  return wvn_kmeans_pixels_launch(code, labels, nseg, (float*)scratch, B, G, H, C, K, iters, relabel, (hipStream_t)stream);
}
int wvn_kmeans_pixels_linear_supported_shape(int G, int H, int C, int K) { return wvn_kmeans_pixels_linear_supported(G, H, C, K); }
size_t wvn_kmeans_pixels_linear_scratch_bytes(int B, int G, int H, int C, int K) {
  return (B > 0 && wvn_kmeans_pixels_linear_supported(G, H, C, K)) ? wvn_kmeans_pixels_linear_scratch_floats(B, G, H, C, K) * sizeof(float) : 0;
}
int wvn_kmeans_cosine_pixels_linear(const float* code, int* labels, int* nseg, void* scratch, int B, int G, int H, int C, int K,
                                    int iters, int relabel, void* stream) {
  return wvn_kmeans_pixels_linear_launch(code, labels, nseg, (float*)scratch, B, G, H, C, K, iters, relabel, (hipStream_t)stream);
}
int wvn_kmeans_cosine_pixels_linear_ac(const float* code, int* labels, int* nseg, void* scratch, int B, int G, int H, int C, int K,
                                       int iters, int relabel, int align_corners, void* stream) {
  return wvn_kmeans_pixels_linear_launch(code, labels, nseg, (float*)scratch, B, G, H, C, K, iters, relabel, (hipStream_t)stream, align_corners ? 1 : 0);
}
int wvn_table_bilerp_argmax_ac(const float* table, int* labels, int B, int G, int H, int K, int align_corners, void* stream) {
  return wvn_table_bilerp_argmax_launch(table, labels, B, G, H, K, (hipStream_t)stream, align_corners ? 1 : 0);
}
int wvn_table_argmax_slots(int K) { return wvn_table_slots(K); }
int wvn_table_bilerp_argmax(const float* table, int* labels, int B, int G, int H, int K, void* stream) {
  return wvn_table_bilerp_argmax_launch(table, labels, B, G, H, K, (hipStream_t)stream);
}
int wvn_debug_kmeans_linear_rows(int rc) { wvn_kmeans_pixels_linear_set_rows(rc); return WVN_OK; }
int wvn_flip_average(const float* a, const float* mirrored, float* out, int B, int G, int C, void* stream) {
  return wvn_flip_average_launch(a, mirrored, out, B, G, C, (hipStream_t)stream);
}
int wvn_kmeans_cosine(const float* xn, int* labels, int* nseg, void* scratch, int B, int P, int C, int K, int iters,
                      int relabel, void* stream) {
  return wvn_kmeans_launch(xn, labels, nseg, (float*)scratch, B, P, C, K, iters, relabel, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// traversability MLP
// ---------------------------------------------------------------------------------------------
namespace {
struct MlpOff { size_t W1, b1, W2, b2, W3, b3, total; int O; };
MlpOff mlp_off(const wvn_mlp_desc* d) {
  MlpOff o;
  o.O = 1 + d->D;
  o.W1 = 0; o.b1 = o.W1 + (size_t)d->H1 * d->D; o.W2 = o.b1 + d->H1; o.b2 = o.W2 + (size_t)d->H2 * d->H1;
  o.W3 = o.b2 + d->H2; o.b3 = o.W3 + (size_t)o.O * d->H2; o.total = o.b3 + o.O;
  return o;
}
int mlp_splitk(int R) { int s = (R + 511) / 512; return s < 1 ? 1 : (s > 32 ? 32 : s); }
struct MlpWs { float *h1, *h2, *out, *lr, *g_out, *g_h2, *g_h1, *trav_w, *trav_raw, *part; void* fused; size_t total; };
MlpWs mlp_carve(const wvn_mlp_desc* d, int R, void* base) {
  MlpWs w;
  size_t off = 0;
  const int O = 1 + d->D;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n * sizeof(float), 256); return (float*)((char*)base + o); };
  w.h1 = take((size_t)R * d->H1); w.h2 = take((size_t)R * d->H2); w.out = take((size_t)R * O); w.lr = take(R);
  w.g_out = take((size_t)R * O); w.g_h2 = take((size_t)R * d->H2); w.g_h1 = take((size_t)R * d->H1);
  w.trav_w = take(R); w.trav_raw = take(R);
  size_t mx = (size_t)d->H1 * d->D;
  if ((size_t)d->H2 * d->H1 > mx) mx = (size_t)d->H2 * d->H1;
  if ((size_t)O * d->H2 > mx) mx = (size_t)O * d->H2;
  w.part = take(mx * mlp_splitk(R));
  w.fused = take(wvn_mlp_train_fused_scratch_bytes(R) / sizeof(float) + 1);   // per-tile partials of the four-launch step
  w.total = off;
  return w;
}
int mlp_fwd(const wvn_mlp_desc* d, const float* P, const float* x, int ldx, int R, float* out, float* h1, float* h2,
            hipStream_t st) {
  const MlpOff o = mlp_off(d);
  GemmF32Params p{};
  p.batch = 1; p.splitk = 1; p.transB = 1;
  p.A = x; p.lda = ldx; p.B = P + o.W1; p.ldb = d->D; p.bias = P + o.b1; p.C = h1; p.ldc = d->H1; p.M = R; p.N = d->H1; p.K = d->D;
  RET_IF(wvn_gemm_f32_launch(p, F32_EPI_RELU, st));
  p.A = h1; p.lda = d->H1; p.B = P + o.W2; p.ldb = d->H1; p.bias = P + o.b2; p.C = h2; p.ldc = d->H2; p.N = d->H2; p.K = d->H1;
  RET_IF(wvn_gemm_f32_launch(p, F32_EPI_RELU, st));
  p.A = h2; p.lda = d->H2; p.B = P + o.W3; p.ldb = d->H2; p.bias = P + o.b3; p.C = out; p.ldc = o.O; p.N = o.O; p.K = d->H2;
  RET_IF(wvn_gemm_f32_launch(p, F32_EPI_SIGMOID0, st));
  return WVN_OK;
}
// dW[M,N] = G^T[M,R] * Hm[R,N] via deterministic split-K
int mlp_wgrad(const float* G, int ldg, const float* Hm, int ldh, int M, int N, int R, float* part, float* dst,
              hipStream_t st) {
  const int sk = mlp_splitk(R);
  GemmF32Params p{};
  p.batch = 1; p.splitk = sk; p.A = G; p.lda = ldg; p.transA = 1; p.B = Hm; p.ldb = ldh; p.transB = 0;
  p.C = sk > 1 ? part : dst; p.ldc = N; p.M = M; p.N = N; p.K = R;
  RET_IF(wvn_gemm_f32_launch(p, F32_EPI_NONE, st));
  if (sk > 1) RET_IF(wvn_splitk_reduce_launch(part, sk, (size_t)M * N, nullptr, N, dst, st));
  return WVN_OK;
}
}  // namespace

size_t wvn_mlp_param_count(const wvn_mlp_desc* d) { return d ? mlp_off(d).total : 0; }
size_t wvn_mlp_workspace_bytes(const wvn_mlp_desc* d, int rows) {
  if (!d || rows <= 0) return 0;
  return mlp_carve(d, rows, nullptr).total;
}

int wvn_mlp_forward(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, int R, float* out, float* h1,
                    float* h2, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !params || !x || !out || R <= 0) return WVN_ERR_ARG;
  if (!h1 || !h2) {
    if (!workspace) return WVN_ERR_ARG;
    MlpWs w = mlp_carve(d, R, workspace);
    if (w.total > workspace_bytes) return WVN_ERR_WORKSPACE;
    if (!h1) h1 = w.h1;
    if (!h2) h2 = w.h2;
  }
  return mlp_fwd(d, params, x, ldx, R, out, h1, h2, (hipStream_t)stream);
}

int wvn_compact_segment_rows(const float* feat, int D, const float* side, int Dside, const int* nseg, int B, int S, float* x_out,
                             float* side_out, int* rows_dev, void* stream) {
  return wvn_compact_segment_rows_launch(feat, D, side, Dside, nseg, B, S, x_out, side_out, rows_dev, (hipStream_t)stream);
}

int wvn_mlp_train_phase_a_rows(const wvn_mlp_desc* d, const float* params, const float* x, int ldx,
                               const unsigned char* y_valid, int R, const int* rows_dev, double* stats, void* workspace,
                               size_t workspace_bytes, unsigned int* sync_word, void* stream) {
  if (!d || !params || !x || !y_valid || !stats || !workspace || R <= 0) return WVN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  MlpWs w = mlp_carve(d, R, workspace);
  if (w.total > workspace_bytes) return WVN_ERR_WORKSPACE;
  if (sync_word && wvn_mlp_train_fused_ok(d->D, d->H1, d->H2, R)) {   // the four-launch step (mlp_train.hip): forward + statistic
    const MlpOff o = mlp_off(d);
    const size_t off[6] = {o.W1, o.b1, o.W2, o.b2, o.W3, o.b3};
    return wvn_mlp_train_fwd_launch(params, off, o.total, x, ldx, y_valid, R, d->D, rows_dev, w.h1, w.h2, w.out, w.lr, stats, w.fused,
                                    sync_word, st);
  }
  RET_IF(mlp_fwd(d, params, x, ldx, R, w.out, w.h1, w.h2, st));
  return wvn_mlp_rowloss_stats_launch(w.out, 1 + d->D, x, ldx, y_valid, w.lr, stats, R, d->D, st, rows_dev);
}
int wvn_mlp_train_phase_a(const wvn_mlp_desc* d, const float* params, const float* x, int ldx,
                          const unsigned char* y_valid, int R, double* stats, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return wvn_mlp_train_phase_a_rows(d, params, x, ldx, y_valid, R, nullptr, stats, workspace, workspace_bytes, nullptr, stream);
}

int wvn_mlp_train_phase_b(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, const float* y,
                          const unsigned char* y_valid, int R, const double* stats, float std_factor, float w_trav,
                          float w_reco, float* grads, float* confidence_out, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return wvn_mlp_train_phase_b_rows(d, params, x, ldx, y, y_valid, R, nullptr, stats, std_factor, w_trav, w_reco, grads,
                                    confidence_out, workspace, workspace_bytes, 0, stream);
}

int wvn_mlp_train_phase_b_rows(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, const float* y,
                               const unsigned char* y_valid, int R, const int* rows_dev, const double* stats, float std_factor,
                               float w_trav, float w_reco, float* grads, float* confidence_out, void* workspace,
                               size_t workspace_bytes, int fused, void* stream) {
  if (!d || !params || !x || !y || !y_valid || !stats || !grads || !workspace || R <= 0) return WVN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  MlpWs w = mlp_carve(d, R, workspace);
  if (w.total > workspace_bytes) return WVN_ERR_WORKSPACE;
  const MlpOff o = mlp_off(d);
  const int O = o.O;
  if (fused && wvn_mlp_train_fused_ok(d->D, d->H1, d->H2, R)) {   // (phase A of this step ran the fused forward on this workspace)
    const size_t off[6] = {o.W1, o.b1, o.W2, o.b2, o.W3, o.b3};
    return wvn_mlp_train_bwd_launch(params, off, o.total, x, ldx, y, y_valid, R, d->D, rows_dev, w.h1, w.h2, w.out, w.lr, w.g_out, w.g_h2,
                                    w.g_h1, stats, std_factor, w_trav, w_reco, confidence_out, grads, w.fused, st);
  }
  RET_IF(wvn_mlp_gradout_launch(w.out, O, x, ldx, y, y_valid, w.lr, stats, std_factor, w_trav, w_reco, w.g_out, O,
                                w.trav_w, w.trav_raw, confidence_out, grads + o.total, R, d->D, st, rows_dev));
  // layer 3
  RET_IF(mlp_wgrad(w.g_out, O, w.h2, d->H2, O, d->H2, R, w.part, grads + o.W3, st));
  RET_IF(wvn_colsum_launch(w.g_out, O, R, O, grads + o.b3, st));
  {
    GemmF32Params p{};
    p.batch = 1; p.splitk = 1; p.A = w.g_out; p.lda = O; p.B = params + o.W3; p.ldb = d->H2; p.transB = 0;
    p.C = w.g_h2; p.ldc = d->H2; p.M = R; p.N = d->H2; p.K = O; p.mask = w.h2; p.ldmask = d->H2;
    RET_IF(wvn_gemm_f32_launch(p, F32_EPI_RELUMASK, st));
  }
  // layer 2
  RET_IF(mlp_wgrad(w.g_h2, d->H2, w.h1, d->H1, d->H2, d->H1, R, w.part, grads + o.W2, st));
  RET_IF(wvn_colsum_launch(w.g_h2, d->H2, R, d->H2, grads + o.b2, st));
  {
    GemmF32Params p{};
    p.batch = 1; p.splitk = 1; p.A = w.g_h2; p.lda = d->H2; p.B = params + o.W2; p.ldb = d->H1; p.transB = 0;
    p.C = w.g_h1; p.ldc = d->H1; p.M = R; p.N = d->H1; p.K = d->H2; p.mask = w.h1; p.ldmask = d->H1;
    RET_IF(wvn_gemm_f32_launch(p, F32_EPI_RELUMASK, st));
  }
  // layer 1
  RET_IF(mlp_wgrad(w.g_h1, d->H1, x, ldx, d->H1, d->D, R, w.part, grads + o.W1, st));
  RET_IF(wvn_colsum_launch(w.g_h1, d->H1, R, d->H1, grads + o.b1, st));
  return WVN_OK;
}

int wvn_mlp_train_phase_c(const wvn_mlp_desc* d, float* params, const float* grads, float* adam_m, float* adam_v,
                          int step, float lr, const double* stats, float w_trav, float w_reco, float* losses,
                          void* stream) {
  if (!d || !params || !grads || !adam_m || !adam_v || !stats || step <= 0) return WVN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const MlpOff o = mlp_off(d);
  // Adam and the step's losses in ONE launch
  return wvn_adam_launch(params, grads, adam_m, adam_v, (int)o.total, step, lr, 0.9f, 0.999f, 1e-8f, st, stats, grads + o.total, w_trav,
                         w_reco, losses);
}

// fused per-pixel inference (pixel_mlp.hip)
size_t wvn_pixel_mlp_pack_bytes(const wvn_mlp_desc* d) {
  if (!d || d->H1 != 256 || d->H2 != 32) return 0;
  return wvn_pixel_mlp_pack_bytes_impl(d->D);
}
int wvn_pixel_mlp_zx_cols(const wvn_mlp_desc* d) {
  if (!d || d->H1 != 256 || d->H2 != 32) return 0;
  return wvn_pixel_mlp_zx_cols_impl(d->D);
}
int wvn_pixel_mlp_pack(const wvn_mlp_desc* d, const float* params, void* packed, void* stream) {
  if (!d) return WVN_ERR_ARG;
  return wvn_pixel_mlp_pack_launch(d->D, d->H1, d->H2, params, packed, (hipStream_t)stream);
}
int wvn_pixel_mlp_infer(const wvn_mlp_desc* d, const void* packed, void* zx, int ldzx, int batch, int grid, int out_h,
                        int out_w, float mean, float std, float std_factor, const float* conf_state, float* trav,
                        float* conf, float* loss_reco, void* stream) {
  if (!d) return WVN_ERR_ARG;
  return wvn_pixel_mlp_infer_launch(d->D, d->H1, d->H2, packed, zx, ldzx, batch, grid, out_h, out_w, mean, std, std_factor,
                                    conf_state, trav, conf, loss_reco, (hipStream_t)stream);
}

size_t wvn_pixel_mlp_exact_pack_bytes(const wvn_mlp_desc* d) {
  if (!d || d->H1 != 256 || d->H2 != 32) return 0;
  return wvn_pixel_mlp_exact_pack_bytes_impl(d->D);
}
size_t wvn_pixel_mlp_exact_workspace_bytes(const wvn_mlp_desc* d, int batch, int grid) {
  if (!d || d->H1 != 256 || d->H2 != 32 || batch <= 0 || grid <= 0) return 0;
  return wvn_pixel_mlp_exact_workspace_bytes_impl(d->D, batch, grid);
}
int wvn_pixel_mlp_exact_pack(const wvn_mlp_desc* d, const float* params, void* packed, void* stream) {
  if (!d) return WVN_ERR_ARG;
  return wvn_pixel_mlp_exact_pack_launch(d->D, d->H1, d->H2, params, packed, (hipStream_t)stream);
}
int wvn_pixel_mlp_infer_exact(const wvn_mlp_desc* d, const float* params, const void* packed, const float* tokens,
                              int ld_tokens, int batch, int grid, int out_h, int out_w, float mean, float std,
                              float std_factor, const float* conf_state, float* trav, float* conf, float* loss_reco,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!d) return WVN_ERR_ARG;
  return wvn_pixel_mlp_infer_exact_launch(d->D, d->H1, d->H2, params, packed, tokens, ld_tokens, batch, grid, out_h, out_w,
                                          mean, std, std_factor, conf_state, trav, conf, loss_reco, workspace,
                                          workspace_bytes, (hipStream_t)stream);
}

int wvn_mlp_confidence(const float* out, int ldo, const float* x, int ldx, float mean, float std, float std_factor,
                       float* trav, float* conf, int R, int D, void* stream) {
  if (!out || !x) return WVN_ERR_ARG;
  return wvn_mlp_confidence_launch(out, ldo, x, ldx, mean, std, std_factor, trav, conf, R, D, (hipStream_t)stream);
}

}  // extern "C"
