// Internal (C++) launcher interfaces shared between the .hip translation units and api.hip.
// The public, C-ABI surface is include/wvn_hip.h.
#pragma once
#include "common.h"

// ---- bf16 MFMA GEMM (gemm_bf16.hip) -------------------------------------------------------------
enum GemmEpilogue {
  EPI_BF16 = 0,       // C(bf16)  = acc + bias
  EPI_GELU_BF16 = 1,  // C(bf16)  = gelu(acc + bias)            (exact erf GELU)
  EPI_RELU_BF16 = 2,  // C(bf16)  = relu(acc + bias)
  EPI_F32 = 3,        // C(f32)   = acc + bias
  EPI_RESID_F32 = 4,  // C(f32)  += acc + bias                  (residual stream update, in place)
  EPI_ACCUM_F32 = 5,  // C(f32)  += acc + bias                  (same arithmetic; second GEMM of a sum)
  EPI_PATCH = 6,      // patch-embed: row remap (b,p) -> b*ntok+1+p, + pos[1+p]
  EPI_QKV = 7,        // scatter to q/k [b,h,npad,64] and v^T [b,h,64,npad]
  EPI_GELU_MX8 = 9,   // gemm_a768_fp8 only: gelu(acc + bias) as e4m3 with one E8M0 block scale per (row, 32 columns): the MX operand of the next product (gemm_fp8.hip AMX)
  EPI_GELU_FRAG = 8,  // gemm_a384_x3 only: gelu(acc + bias) as fragment-major hi / lo planes (the A operand of gemm_n384_x3's AFRAG form)
};

struct GemmBf16Params {
  const bf16_t* A; int lda;   // [M,K]
  const bf16_t* W; int ldw;   // [N,K]
  const float* bias;          // [N] or nullptr
  void* C; int ldc;
  int M, N, K;
  // EPI_PATCH
  const float* pos; int npatch; int ntok;
  int ntok_s;  // rows per frame in the token matrices (ntok rounded up to 8); row of (b,t) = b*ntok_s + t
  // EPI_QKV
  bf16_t* q; bf16_t* k; bf16_t* vt; int heads; int npad;
  float q_scale;  // EPI_QKV: the q third is multiplied by this before it is rounded to bf16 (0 = leave as is)
  int qkv_f16;    // x3 kernels, EPI_QKV: q / k / vt leave as ONE fp16 plane each (the operands of the fp16 attention kernel, WVN_PREC_MIX);
                  // with q_lo set, q leaves as TWO fp16 planes (q_lo = the rounding residue of q: attention_bf16.hip QSPLIT)
  const float* ls;  // EPI_RESID_F32: optional LayerScale vector [N] (DINOv2): C += ls * (acc + bias); nullptr = plain residual
  // exact mode (gemm_x3.hip): the lo planes of the operands and of plane-typed outputs (hi planes are A / W / C / q / k / vt)
  const bf16_t* A_lo; const bf16_t* W_lo; void* C_lo; bf16_t* q_lo; bf16_t* k_lo; bf16_t* vt_lo;
  // MX correction terms (round 6; gemm_n384_x3.hip / gemm_a384_x3.hip): operands as an fp16 plane + two e5m2 planes (l8 = the rounding residue * 2^12,
  // h8 = the value itself); A_lo / C_lo name the l8 plane, A_h8 / C_h8 the h8 plane
  const void* A_h8; void* C_h8;
  long long* dbg;  // optional: per-wave phase timings of the A-stationary kernel (scripts/ab_kernels.py --timing)
  // LayerNorm across kernel boundaries (round 4, split-operand kernels): the row-panel kernels (gemm_n384_x3.hip) can leave the statistics
  // of the rows they have just updated, ln_stats_out[m] = {mean, 1 / sqrt(var + ln_eps)} over the N = 384 columns; the A-stationary
  // kernel (gemm_a384_x3.hip) can take its A operand as LayerNorm(ln_x) on the fly -- ln_x fp32 [M][ln_ldx], the statistics of its rows,
  // gamma / beta [384] -- normalising and splitting each row into its two planes as it loads it (A / A_lo are then ignored)
  float* ln_stats_out; float ln_eps;
  const float* ln_x; int ln_ldx; const float* ln_stats; const float* ln_g; const float* ln_b;
};
// The launchers of the 16-bit-operand speed path exist twice, once per operand format (operand.h): the plain names take bf16
// operands, the *_f16 names fp16 operands (same kernels, compiled with -DWVN_OPERAND_F16=1).  "bf16_t" in these signatures
// is the raw 16-bit storage type of either format.
#define WVN_DECLARE_OPERAND_LAUNCHERS(SFX)                                                                                          \
  int wvn_gemm_bf16_launch##SFX(const GemmBf16Params& p, int epi, hipStream_t st);                                                  \
  /* A-stationary kernel for K == 384 (gemm_a384.hip); WVN_ERR_ARG when the shape is not eligible */                                \
  int wvn_gemm_a384_launch##SFX(const GemmBf16Params& p, int epi, hipStream_t st);                                                  \
  /* mlp_fused.hip: x += gelu(xn W1^T + b1) W2p^T + b2 with the hidden activation kept in registers (D = 384 only);                 \
     xn == nullptr: xn = LayerNorm(x; ln_g, ln_b, ln_eps) computed in the kernel, once per row block */                             \
  int wvn_mlp_fused_launch##SFX(const bf16_t* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const bf16_t* W1,    \
                                const float* b1, const bf16_t* W2p, const float* b2, const float* ls, float* x, int ldx, int M,     \
                                int F, hipStream_t st);                                                                             \
  /* gemm_proj.hip: x[M,384] += (A[M,384] W[384,384]^T + bias) (* ls): W resident in LDS, one 32-row group per wave at a time */    \
  int wvn_proj_resid_launch##SFX(const bf16_t* A, int lda, const bf16_t* W, const float* bias, const float* ls, float* x, int ldx,  \
                                 int M, hipStream_t st);                                                                            \
  /* mlp_fused.hip, with the attention output projection of the block in its prologue:                                              \
     x += (attn Wp^T + bp) (* ls1); x += MLP(LN(x)) */                                                                              \
  int wvn_proj_mlp_fused_launch##SFX(const bf16_t* attn, int lda_attn, const bf16_t* Wp, const float* bp, const float* ls1,         \
                                     const float* ln_g, const float* ln_b, float ln_eps, const bf16_t* W1, const float* b1,         \
                                     const bf16_t* W2p, const float* b2, const float* ls2, float* x, int ldx, int M, int F,         \
                                     hipStream_t st, const bf16_t* W1p, const float* nx_g, const float* nx_b, float nx_eps,         \
                                     bf16_t* xn_next);                                                                              \
  /* qkv_fused.hip: LayerNorm(x) -> q | k | v^T in the layouts of attention_bf16.hip (D = 384, heads = 6), one launch */            \
  int wvn_qkv_fused_launch##SFX(const float* x, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const bf16_t* W,       \
                                const float* bias, bf16_t* q, bf16_t* k, bf16_t* vt, int heads, int npad, int ntok_s,               \
                                float q_scale, int M, hipStream_t st, const bf16_t* xn_frag);                                       \
  /* row-panel kernel for N == 384 residual updates with long K (gemm_n384.hip); WVN_ERR_ARG when not eligible */                   \
  int wvn_gemm_n384_launch##SFX(const GemmBf16Params& p, int epi, hipStream_t st, int* rows_done, int force = 0);                   \
  /* attention_bf16.hip */                                                                                                          \
  int wvn_attention_bf16_launch##SFX(const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* out, int B, int heads, int ntok,   \
                                     int ntok_s, int npad, float scale, hipStream_t st, bf16_t* out_lo = nullptr,                   \
                                     const bf16_t* q_lo = nullptr, int out_frag = 0);                                               \
  void wvn_attention_bf16_set_debug##SFX(long long* dbg); /* per-wave phase timings (TIMING build), nullptr = off */                \
  void wvn_attention_bf16_set_variant##SFX(int v);                                                                                  \
  extern long long* g_mlp_fused_dbg##SFX;                                                                                           \
  extern long long* g_qkv_fused_dbg##SFX;
WVN_DECLARE_OPERAND_LAUNCHERS()
WVN_DECLARE_OPERAND_LAUNCHERS(_f16)
// exact mode: hi/lo bf16 planes, three MFMAs per product (gemm_x3.hip); same epilogue codes, plane-typed outputs for the
// "bf16" ones
int wvn_gemm_x3_launch(const GemmBf16Params& p, int epi, hipStream_t st);
// the A-stationary form for K == 384 (gemm_a384_x3.hip); WVN_ERR_ARG when the shape / epilogue is not eligible
int wvn_gemm_a384_x3_launch(const GemmBf16Params& p, int epi, hipStream_t st);
// the row-panel form for N == 384 residual updates (gemm_n384_x3.hip: fc2, attention projection); WVN_ERR_ARG when not eligible
int wvn_gemm_n384_x3_launch(const GemmBf16Params& p, int epi, hipStream_t st);
// the same with A as the fragment-major planes EPI_GELU_FRAG writes and W packed by wvn_pack_n384_x3_weight's layout (k-step-major,
// both planes, the column permutation of the fragments, swizzled 32-byte rows: every DMA piece one contiguous kilobyte)
int wvn_gemm_n384_x3_frag_launch(const GemmBf16Params& p, int epi, hipStream_t st);
// the MX form of the fragment kernel (fp16 hi plane + two e5m2 planes per operand; W packed by backbone.pack_n384_mx)
int wvn_gemm_n384_mx_launch(const GemmBf16Params& p, int epi, hipStream_t st);
// the MX form of the A-stationary kernel (LayerNorm on load; EPI_GELU_FRAG -> MX fragment planes, EPI_QKV -> fp16 q | k | v^T); W packed by backbone.pack_a384_mx
int wvn_gemm_a384_mx_launch(const GemmBf16Params& p, int epi, hipStream_t st);

// ---- fp8 (e4m3) MFMA GEMM with per-row scales of both operands (gemm_fp8.hip) + the row quantisers (fp8.hip) -----------
struct GemmFp8Params {
  const unsigned char* A; int lda;   // [M,K] e4m3, lda in elements (= bytes)
  const unsigned char* W; int ldw;   // [N,K] e4m3
  const float* sa;                   // [M] scale of every A row  (value = q * sa)
  const float* sw;                   // [N] scale of every W row
  const float* bias;                 // [N] or nullptr
  void* C; int ldc;
  int M, N, K;
  const float* pos; int npatch; int ntok; int ntok_s;                         // (EPI_PATCH is not instantiated for fp8)
  bf16_t* q; bf16_t* k; bf16_t* vt; int heads; int npad; float q_scale;      // EPI_QKV (bf16 outputs for attention_bf16.hip)
  const float* ls;                                                            // EPI_RESID_F32: optional LayerScale
  const unsigned char* a_scales;     // optional: MX block scales of A, one E8M0 byte per (row, 32 k): [M][K / 32]; sa is then not read (EPI_F32 / EPI_RESID_F32)
  unsigned char* c_scales;           // gemm_a768_fp8.hip, EPI_GELU_MX8: the block scales of the e4m3 output C [M][ldc], [M][N / 32]
};
int wvn_gemm_fp8_launch(const GemmFp8Params& p, int epi, hipStream_t st);
// the DMA-fed form for long K (gemm_fp8_dma.hip: K % 512 == 0, N % 128 == 0, EPI_F32 / EPI_RESID_F32, per-row or MX block scales on A): wvn_gemm_fp8_launch tries it first
int wvn_gemm_fp8_dma_launch(const GemmFp8Params& p, int epi, hipStream_t st);
// the A-stationary form for K == 768 (gemm_a768_fp8.hip): Wp = backbone.pack_a768_fp8 of the e4m3 weight; EPI_BF16 / EPI_GELU_BF16 / EPI_RESID_F32 / EPI_QKV;
// WVN_ERR_ARG where the shape is not its (the caller falls back to wvn_gemm_fp8_launch)
int wvn_gemm_a768_fp8_launch(const GemmFp8Params& p, const void* Wp, int epi, hipStream_t st);
// rows of src (fp32 or bf16, leading dimension lds_) -> e4m3 rows (leading dimension ldq, bytes) + scale[rows] = amax / 448
int wvn_quantize_rows_fp8_launch(const void* src, int src_bf16, int lds_, unsigned char* q, int ldq, float* scale, int rows,
                                 int cols, hipStream_t st);
// LayerNorm (fp32 rows of x, D = 64 * {6, 12}) fused with the row quantiser
int wvn_layernorm_fp8_launch(const float* x, const float* gamma, const float* beta, unsigned char* q, int ldq, float* scale,
                             int rows, int D, float eps, hipStream_t st);

// ---- fp32 GEMM (gemm_f32.hip): exact-mode linears + the traversability MLP ---------------------
enum GemmF32Epilogue {
  F32_EPI_NONE = 0,       // C = acc + bias
  F32_EPI_RELU = 1,
  F32_EPI_GELU = 2,
  F32_EPI_RESID = 3,      // C += acc + bias
  F32_EPI_SIGMOID0 = 4,   // C = acc + bias, sigmoid applied to column 0 (SimpleMLP output)
  F32_EPI_RELUMASK = 5,   // C = (mask > 0) ? acc : 0    (backward through ReLU; mask = forward activation)
  F32_EPI_PATCH = 6,
  F32_EPI_QKV = 7,        // scatter to q/k [b,h,npad,64] and v [b,h,npad,64] (fp32, not transposed)
};
struct GemmF32Params {
  const float* A; int lda; int transA;  // transA=0: A[M,K] row-major ; 1: A stored [K,M]
  const float* B; int ldb; int transB;  // transB=0: B[K,N] row-major ; 1: B stored [N,K] (torch Linear)
  const float* bias;
  float* C; int ldc;
  int M, N, K;
  int batch; long long strideA, strideB, strideC;  // batched over blockIdx.z (0 => shared)
  int splitk;                                      // >1: C must hold splitk partial [M,N] slabs (stride M*ldc)
  const float* mask; int ldmask;                   // F32_EPI_RELUMASK
  const float* pos; int npatch; int ntok; int ntok_s;  // F32_EPI_PATCH (ntok_s: rows per frame)
  float* q; float* k; float* v; int heads; int npad;  // F32_EPI_QKV
  const float* ls;                                    // F32_EPI_RESID: optional LayerScale vector [N]
};
int wvn_gemm_f32_launch(const GemmF32Params& p, int epi, hipStream_t st);

// ---- elementwise / normalisation (elementwise.hip) --------------------------------------------
// Frame ingest: NEAREST resize + centre crop as two index tables over the source frame.  Network pixel (y, x) is
// frame pixel (rows[y], cols[x]) of a [.., src_h, src_w] frame; rows / cols: device int32 [S].
struct WvnIngest { const int* rows; const int* cols; int src_h, src_w; };
// img: fp32 in [0,1], or raw uint8 pixels when img_u8 != 0.  out_mode: 0 fp32, 1 bf16, 2 hi / lo bf16 planes (patches_lo), 3 fp16;
// ldp: row stride in elements (0 = 3*P*P; pad columns are NOT written); ing: optional gather tables (the frames are then
// [B,3,src_h,src_w])
int wvn_patchify_launch(const void* img, int img_u8, void* patches, void* patches_lo, int out_mode, int ldp, int B, int S, int P,
                        hipStream_t st, const WvnIngest* ing = nullptr);
int wvn_gather_image_launch(const void* in, void* out, long long planes, int out_h, int out_w, int elem_bytes, const WvnIngest* ing,
                            hipStream_t st);
int wvn_split_planes_launch(const float* src, int lds_, bf16_t* hi, bf16_t* lo, int ldd, int rows, int cols, hipStream_t st);
int wvn_cls_rows_launch(const float* cls_pos, float* x, int B, int ntok_s, int D, hipStream_t st);
// zero bytes [col0, col0 + ncol) of each of nrows rows (all multiples of 4)
int wvn_pad_zero_launch(void* base, long long nrows, long long row_stride_bytes, long long col0_bytes,
                        long long ncol_bytes, hipStream_t st);
// LayerNorm over rows of x[rows, D] (fp32) -> y (y_fmt: 0 fp32, 1 bf16, 2 fp16; leading dim ldy); optional second fp32 output.
// row_map: 0 = identity; 1 = drop the class token (input row b*ntok+1+p -> output row b*(ntok-1)+p)
int wvn_layernorm_launch(const float* x, const float* gamma, const float* beta, void* y, int y_fmt, int ldy,
                         float* y2, int ldy2, int rows_out, int D, float eps, int drop_cls, int ntok,
                         int ntok_s, hipStream_t st, void* y_lo = nullptr);  // y_lo: exact mode, lo plane of the bf16 output
int wvn_cast_f32_bf16_launch(const float* src, int lds_, bf16_t* dst, int ldd, int rows, int cols, hipStream_t st, int f16 = 0);

// ---- attention (attention_bf16.hip / attention_f32.hip) ---------------------------------------
int wvn_attention_f32_launch(const float* q, const float* k, const float* v, float* out, int B, int heads, int ntok,
                             int ntok_s, int npad, float scale, hipStream_t st);
// exact mode on the matrix pipe (attention_x3.hip): hi / lo planes of q, k [B,h,npad,64], v^T [B,h,64,npad] (token-permuted),
// out planes [B*ntok_s, h*64]
int wvn_attention_x3_launch(const bf16_t* q_hi, const bf16_t* q_lo, const bf16_t* k_hi, const bf16_t* k_lo, const bf16_t* vt_hi,
                            const bf16_t* vt_lo, bf16_t* out_hi, bf16_t* out_lo, int B, int heads, int ntok, int ntok_s,
                            int npad, float scale, hipStream_t st);

// ---- misc launchers defined across the translation units ---------------------------------------
int wvn_splitk_reduce_launch(const float* part, int splitk, size_t n, const float* bias, int ncols, float* out,
                             hipStream_t st);
int wvn_upsample_bilinear_launch(const float* tok, float* out, int B, int G, int D, int H, hipStream_t st);
int wvn_upsample_nearest_i32_launch(const int* lab, int* out, int B, int G, int H, hipStream_t st);
int wvn_segpool_launch(const int* seg, const float* tok, int ldf, float* feat, void* W, int* cnt, int B, int H,
                       int Wd, int G, int S, int D, hipStream_t st);
int wvn_label_pool_batched_launch(const void* nodes, int n, int C, int H, int Wd, int Smax, long long* sum, int* cnt,
                                  hipStream_t st);
int wvn_label_pool_launch(const float* mask, int C, const int* seg, float* signal, unsigned char* valid, void* sum,
                          int* cnt, int H, int Wd, int S, hipStream_t st);
int wvn_centers_launch(const int* seg, float* centers, unsigned long long* scratch, int H, int Wd, int S,
                       hipStream_t st);
int wvn_adjacency_launch(const int* seg, long long* edges, int* count, unsigned char* bitmap, int H, int Wd, int S,
                         int max_edges, hipStream_t st);
int wvn_normalize_rows_launch(const float* code, int ldc, float* xn, int rows, int C, hipStream_t st);
int wvn_argmax_rows_launch(const float* x, int ld, int rows, int cols, int* out, hipStream_t st);
size_t wvn_kmeans_scratch_floats(int B, int P, int C, int K);
size_t wvn_kmeans_pixels_scratch_floats(int B, int G, int H, int C, int K);
int wvn_kmeans_pixels_launch(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int C, int K, int iters,
                             int relabel, hipStream_t st);
int wvn_km_pix_prepare_launch(const float* code, float* rinv, float* cent, int B, int G, int H, int C, int K, hipStream_t st, int align_corners = 1);
int wvn_km_relabel_launch(int* labels, int* nseg, int B, long long P, int K, int relabel, hipStream_t st);
int wvn_kmeans_pixels_linear_supported(int G, int H, int C, int K);
size_t wvn_kmeans_pixels_linear_scratch_floats(int B, int G, int H, int C, int K);
void wvn_kmeans_pixels_linear_set_rows(int rc);
int wvn_kmeans_pixels_linear_launch(const float* code, int* labels, int* nseg, float* scratch, int B, int G, int H, int C, int K,
                                    int iters, int relabel, hipStream_t st, int align_corners = 1);
int wvn_f16_saturate_probe_launch(const float* in, uint16_t* out, int n, hipStream_t st);
void wvn_gemm_n384_x3_set_pair(int on);
void wvn_gemm_a384_mx_set_form(int form);   // 1: one wave per SIMD (gemm_a384_x3_kernel<..., MX>), 2: two workgroups per CU (gemm_a384_mx2_kernel), 0: the default (fc1 on 2, QKV on 1)
int wvn_table_slots(int K);
int wvn_table_bilerp_argmax_launch(const float* table, int* labels, int B, int G, int H, int K, hipStream_t st, int align_corners = 1);
void wvn_kmeans_pixels_set_assign_form(int form);
int wvn_kmeans_pixels_screen_stats(unsigned long long* out, int reset);   // -1 default (MFMA where eligible), 0 the VALU form always
int wvn_flip_average_launch(const float* a, const float* mirrored, float* out, int B, int G, int C, hipStream_t st);
int wvn_kmeans_launch(const float* xn, int* labels, int* nseg, float* scratch, int B, int P, int C, int K, int iters,
                      int relabel, hipStream_t st);
int wvn_mlp_rowloss_stats_launch(const float* out, int ldo, const float* x, int ldx, const unsigned char* valid,
                                 float* lr, double* stats, int R, int D, hipStream_t st, const int* rows_dev = nullptr);
int wvn_compact_segment_rows_launch(const float* feat, int D, const float* side, int Ds, const int* nseg, int B, int S, float* x,
                                    float* side_out, int* count, hipStream_t st);
// mlp_train.hip: the four-launch optimisation step (fwd | bwd + wgrad | adam).  off: {W1, b1, W2, b2, W3, b3} offsets into the flat
// parameter / gradient vectors; scratch: wvn_mlp_train_fused_scratch_bytes(R); sync_word: a zero device word, left at zero
bool wvn_mlp_train_fused_ok(int D, int H1, int H2, int R);
size_t wvn_mlp_train_fused_scratch_bytes(int R);
int wvn_mlp_train_fwd_launch(const float* P, const size_t* off, size_t ntotal, const float* x, int ldx, const unsigned char* valid, int R,
                             int D, const int* rows_dev, float* h1, float* h2, float* out, float* lr, double* stats, void* scratch,
                             unsigned* sync_word, hipStream_t st);
int wvn_mlp_train_bwd_launch(const float* P, const size_t* off, size_t ntotal, const float* x, int ldx, const float* y,
                             const unsigned char* valid, int R, int D, const int* rows_dev, float* h1, float* h2, float* out, float* lr,
                             float* g_out, float* g_h2, float* g_h1, const double* stats, float std_factor, float w_trav, float w_reco,
                             float* conf_out, float* grads, void* scratch, hipStream_t st);
int wvn_mlp_gradout_launch(const float* out, int ldo, const float* x, int ldx, const float* y,
                           const unsigned char* valid, const float* lr, const double* stats, float std_factor,
                           float w_trav, float w_reco, float* g, int ldg, float* trav_w, float* trav_raw,
                           float* conf_out, float* extra, int R, int D, hipStream_t st, const int* rows_dev = nullptr);
int wvn_colsum_launch(const float* A, int lda, int R, int N, float* outv, hipStream_t st);
// stats / extra / losses non-null: the same launch also writes the step's losses (block 0)
int wvn_adam_launch(float* p, const float* g, float* m, float* v, int n, int step, float lr, float b1, float b2,
                    float eps, hipStream_t st, const double* stats = nullptr, const float* extra = nullptr, float w_trav = 0.f,
                    float w_reco = 0.f, float* losses = nullptr);
int wvn_mlp_losses_launch(const double* stats, const float* extra, float w_trav, float w_reco, float* losses,
                          hipStream_t st);
int wvn_mlp_confidence_launch(const float* out, int ldo, const float* x, int ldx, float mean, float std,
                              float std_factor, float* trav, float* conf, int R, int D, hipStream_t st);
int wvn_segpool_patch_launch(const int* labels, const float* tok, int ldf, const float* wy, const float* wx,
                             float* feat, int B, int G, int S, int D, hipStream_t st);
size_t wvn_segmean_scratch_bytes_impl(int B, int P, int S, int D);
int wvn_segmean_tokens_launch(const int* seg, const float* tok, float* out, int* cnt, void* scratch, size_t scratch_bytes, int B, int P, int S, int D,
                              hipStream_t st);

// ---- fused per-pixel traversability inference (pixel_mlp.hip) ------------------------------------
size_t wvn_pixel_mlp_pack_bytes_impl(int D);
int wvn_pixel_mlp_zx_cols_impl(int D);
int wvn_pixel_mlp_pack_launch(int D, int h1, int h2, const float* params, void* packed, hipStream_t st);
int wvn_pixel_mlp_infer_launch(int D, int h1, int h2, const void* packed, void* zx, int ldzx, int B, int G, int out_h,
                               int out_w, float mean, float std, float std_factor, const float* conf_state, float* trav,
                               float* conf, float* loss, hipStream_t st);
// exact mode (hi + lo split operands, three MFMAs per product)
size_t wvn_pixel_mlp_exact_pack_bytes_impl(int D);
size_t wvn_pixel_mlp_exact_workspace_bytes_impl(int D, int B, int G);
int wvn_pixel_mlp_exact_pack_launch(int D, int h1, int h2, const float* params, void* packed, hipStream_t st);
int wvn_pixel_mlp_infer_exact_launch(int D, int h1, int h2, const float* params, const void* packed, const float* tokens,
                                     int ldt, int B, int G, int out_h, int out_w, float mean, float std, float std_factor,
                                     const float* conf_state, float* trav, float* conf, float* loss, void* workspace,
                                     size_t workspace_bytes, hipStream_t st);

// ---- supervision path (supervision.hip) and SLIC (slic.hip) -------------------------------------------------------------
int wvn_project_render_fmin_launch(const void* nodes, int n, const float* points, int points_batched, int npts, int C, int H,
                                   int W, const float* value_dev, float value, hipStream_t st);
int wvn_slic_num_clusters_impl(int H, int W, int num_components);
size_t wvn_slic_scratch_bytes_impl(int H, int W, int num_components);
int wvn_slic_launch(const void* img, int img_u8, int H, int W, int num_components, float compactness, int iters,
                    const int* lut_lin, const int* lut_f, int* labels, void* scratch, size_t scratch_bytes, hipStream_t st);

// ---- A -> B wire format (wire.hip) ---------------------------------------------------------------------------------------
size_t wvn_wire_bytes_impl(int H, int W, int S, int D);
int wvn_wire_pack_launch(const void* seg, int seg_is_i64, const float* feat, int ldf, void* out, int H, int W, int S, int D,
                         hipStream_t st);
int wvn_wire_unpack_launch(const void* in, long long* seg_i64, int* seg_i32, float* feat, int H, int W, int S, int D,
                           hipStream_t st);
