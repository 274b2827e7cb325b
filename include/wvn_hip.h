/*
 * wvn_hip.h -- C-ABI of libwvn_hip.so: the MI355X (gfx950) native implementation of Wild Visual
 * Navigation's  feature_extractor -> traversability_estimator  hot path.
 *
 * The reference (leggedrobotics/wild_visual_navigation) has no FFI layer: the path sits behind plain
 * Python classes that issue stock PyTorch ops.  Each entry point below replaces the arithmetic of
 * one reference function (cited as file:line, relative to the reference root) and is what a
 * maintainer binds with ctypes.  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host; no allocation happens inside
 *     the library (callers own memory -- torch.empty in the Python host layer);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work
 *     is stream-ordered and asynchronous;
 *   - return value: 0 = ok, 1001 = bad argument, 1002 = workspace too small, otherwise a hipError_t;
 *   - tensors are dense row-major; "tokens" are [B, G*G, D] (patch-major == NHWC feature map).
 */
#ifndef WVN_HIP_H
#define WVN_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WVN_MAX_DEPTH 32
#define WVN_PREC_F32 0  /* exact mode: fp32 storage + fp32 FMA everywhere (parity gate, <= 1e-3) */
#define WVN_PREC_BF16 1 /* fast mode: bf16 operands on MFMA, fp32 accumulate / statistics / residual */
#define WVN_PREC_X3 2   /* exact mode ON THE MATRIX PIPE: every MFMA operand is two bf16 planes (hi + lo, 16 significant bits),  \
                           every product hi*hi + hi*lo + lo*hi in fp32 accumulators (fp32-class results, <= 1e-3 gate), erf GELU, \
                           fp32 residual / LayerNorm / softmax */
#define WVN_PREC_FP8 3  /* BASELINE configs[4]: the four linears of every block on fp8 (OCP e4m3) MFMA at twice the bf16 rate --    \
                           per-token scales of the activations (computed by the LayerNorm / row-quantise kernels), per-output-channel \
                           scales of the weights, fp32 accumulate; attention, patch embedding, LayerNorm and the residual stream as   \
                           in WVN_PREC_BF16 */
#define WVN_PREC_F16 4  /* the WVN_PREC_BF16 path with fp16 operands: the same kernels compiled for v_mfma_f32_32x32x16_f16 /      \
                           v_cvt_pk_f16_f32 / v_dot2c_f32_f16 (same matrix rate and instruction counts), 11 significand bits        \
                           instead of 8 -- 8x less operand rounding; accumulation, residual stream, LayerNorm and softmax            \
                           statistics are fp32 in both.  Range: csrc/operand.h.  Weights: fp16 bits in the bf16 layouts */

#define WVN_PREC_MIX 5  /* the <= 1e-3 parity mode sized by the error budget (profiles/r04a_error_budget_*.md): every LINEAR of the   \
                           network (patch embedding, QKV, projection, fc1, fc2) exactly as in WVN_PREC_X3 -- hi + lo bf16 planes,    \
                           three MFMAs per product, erf GELU -- because each of them alone spends a third of the 1e-3 budget at one  \
                           16-bit value per operand; the two attention products (Q K^T, P V: 58 % of the multiply-adds, 3 % of the  \
                           error) on the fp16-operand kernel of WVN_PREC_F16, fed fp16 q / k / v^T by the QKV epilogue and writing  \
                           its output as hi + lo planes from its fp32 accumulators.  Weights as for WVN_PREC_X3 */

int wvn_version(void);

/* ---------------------------------------------------------------------------------------------
 * DINO ViT backbone  (replaces stego.backbones.backbone.get_backbone(cfg)(img), called from
 * wild_visual_navigation/feature_extractor/dino_interface.py:45,84, plus the T.Normalize of :52).
 * Weights: matrices in torch.nn.Linear layout [out][in]; bf16 bits (uint16) when precision == WVN_PREC_BF16, float for
 * WVN_PREC_F32, for WVN_PREC_X3 TWO stacked bf16 planes [2][out][in]: hi = bf16(w), lo = bf16(w - hi), and for WVN_PREC_FP8
 * e4m3 bytes with the row scales in qkv_s .. fc2_s (the patch weight stays bf16).  For the MFMA precisions the patch weight
 * rows are zero-padded to a multiple of 64 columns (588 -> 640 for patch 14).
 * Biases, LayerNorm affine, LayerScale and the position table are always fp32.
 * ------------------------------------------------------------------------------------------- */
/* wvn_vit_model.flags: which single-kernel forms of the block stages wvn_vit_forward MAY use (WVN_PREC_BF16, D = 384).  They are
 * persistent one-workgroup-per-CU kernels and pay off once the token matrix fills the chip; below that (a single live frame)
 * wvn_vit_forward runs the separate LayerNorm / GEMM kernels whatever the flags say (thresholds: csrc/api.hip).
 * WVN_VIT_MLP_FUSED (F % 64 == 0): every layer also carries fc2_w_fused = fc2.weight with the hidden (input) index permuted --
 * bits 2 and 3 swapped inside each aligned group of 16, fc2_w_fused[n][k] = fc2.weight[n][swap23(k)] -- and the block MLP,
 * including its LayerNorm (blocks.i.norm2), runs as ONE kernel that keeps the normalised rows and the hidden activation in
 * registers (csrc/mlp_fused.hip). */
#define WVN_VIT_MLP_FUSED 1
/* WVN_VIT_QKV_FUSED (WVN_PREC_BF16, D = 384, heads = 6): blocks.i.norm1 and the QKV projection run as ONE kernel that normalises the
 * residual rows in registers (csrc/qkv_fused.hip); no weight re-layout. */
#define WVN_VIT_QKV_FUSED 2
/* use the allowed single-kernel forms at every size (tests, A/B runs), not only where they pay */
#define WVN_VIT_FUSE_ANY_SIZE 4
/* keep the attention projection a separate kernel where the fused MLP kernel runs (A/B runs; default: it is folded into that kernel) */
#define WVN_VIT_NO_PROJ_IN_MLP 8
/* A/B runs: do not let the projection + MLP kernel of a block apply the NEXT block's norm1 (default where fc1_w_fused and the next
 * layer's qkv_w_fused are given: the rows leave that kernel a second time as LayerNorm'ed operand fragments and the next QKV
 * kernel starts from those -- no second pass over the fp32 residual stream, no statistics, half the bytes) */
#define WVN_VIT_NO_LN_HANDOVER 16
#define WVN_VIT_NO_A384_X3 32     /* A/B and tests: the K = 384 linears of WVN_PREC_X3 / WVN_PREC_MIX on the tiled gemm_x3 kernel instead of the A-stationary one */
/* A/B and tests of the split-operand block kernels, one piece at a time (python: WVN_X3_DEBUG_BITS): 64 no A-stationary kernel, 128 no
 * row-panel kernel, 256 no fragment-major MLP hand-over */
#define WVN_VIT_X3_NO_A384 64
#define WVN_VIT_X3_NO_N384 128
#define WVN_VIT_X3_NO_FRAG_MLP 256
/* ... and: no LayerNorm across kernel boundaries (default for WVN_PREC_MIX / WVN_PREC_X3 from 8192 rows on: the row-panel kernels leave
 * {mean, rstd} of the rows they update, the A-stationary kernels normalise as they load; with this flag every LayerNorm is its own
 * kernel writing hi / lo planes again) */
#define WVN_VIT_X3_NO_LN_STATS 512
/* WVN_PREC_MIX: do NOT use the MX form of the block linears (round 6) even where the layer carries the packed weights *_w_mx: the bf16 x 3
 * kernels of rounds 4 / 5 (A/B runs and tests) */
#define WVN_VIT_NO_MX 1024
/* WVN_PREC_MIX: in how many LEADING blocks the attention kernel takes q as two fp16 planes (8 more MFMAs per tile, three workgroups per CU
 * instead of four: 3.08 against 2.05 ms per 128-frame launch).  The query's rounding is the one attention operand error that does not
 * average out over a row's keys, and it is injected in the EARLY blocks: on the reference's real 448^2 frame the token error is 1.05e-3
 * with no block split, 5.1e-4 / 4.3e-4 / 2.3e-4 with the first 2 / 4 / 6, 7.6e-5 with all twelve -- and still 1.05e-3 with only the LAST six
 * (profiles/r05_qsplit_blocks.md).  Bits 16 - 21 of flags hold n + 1; the field left 0 selects the default (6).  WVN_VIT_QSPLIT_BLOCKS(12)
 * is the round-4 behaviour. */
#define WVN_VIT_QSPLIT_BLOCKS(n) ((((n) + 1) & 63) << 16)
#define WVN_VIT_QSPLIT_DEFAULT 6
typedef struct wvn_vit_layer {
  const void* qkv_w;  /* [3D][D]   blocks.i.attn.qkv.weight  */
  const void* proj_w; /* [D][D]    blocks.i.attn.proj.weight */
  const void* fc1_w;  /* [F][D]    blocks.i.mlp.fc1.weight   */
  const void* fc2_w;  /* [D][F]    blocks.i.mlp.fc2.weight   */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float *ls1, *ls2; /* [D] LayerScale of the attention / MLP branch (DINOv2 blocks.i.ls{1,2}.gamma); NULL = none (DINO) */
  const float *qkv_s, *proj_s, *fc1_s, *fc2_s; /* WVN_PREC_FP8: per-output-channel scale of each e4m3 weight row (w = q * s) */
  const void* fc2_w_fused;                     /* WVN_VIT_MLP_FUSED: fc2.weight with the permuted hidden index (see above); else NULL.
                                                * WVN_PREC_X3 / WVN_PREC_MIX (optional, D = 384): fc2.weight packed for the fragment-major MLP
                                                * (csrc/gemm_n384_x3.hip, AFRAG): bf16 [F / 16 k-steps][2 planes hi, lo][384 rows n][2 chunks][8],
                                                * chunk position cp of row n holds chunk c = cp ^ ((n >> 3) & 1), element j of chunk c =
                                                * fc2.weight[n][16 s + swap23(8 c + j)] (swap23: bits 2 and 3 exchanged); NULL: the row-major MLP */
  const void* fc1_w_fused; /* optional with WVN_VIT_MLP_FUSED: fc1.weight with its COLUMN (input) index permuted the same way,
                            * fc1_w_fused[f][k] = fc1.weight[f][swap23(k)].  With it (and no LayerScale) the block's projection + MLP
                            * kernel keeps the residual rows in its accumulator registers: one read and one write of the residual
                            * stream per block instead of two and two (wvn_proj_mlp_resident).  NULL: the form without it */
  const void* qkv_w_fused; /* optional with WVN_VIT_QKV_FUSED: qkv.weight with its column index permuted the same way: the QKV kernel
                            * of this block can then start from the fragments the previous block's kernel left (wvn_qkv_prenorm) */
  const void* proj_w_frag; /* WVN_PREC_X3 / WVN_PREC_MIX (optional, D = 384): attn.proj.weight in the packed layout of fc2_w_fused's
                            * split-operand form.  With it, WVN_PREC_MIX's attention kernel writes its output as MFMA operand fragments
                            * and the projection runs on the fragment form of csrc/gemm_n384_x3.hip.  NULL: the row-major projection */
  /* WVN_PREC_MIX, D = 384 (optional, all four or none): the weights in the MX operand representation -- per element h = fp16(w), l8 = e5m2((w - h) * 2^12),
   * h8 = e5m2(w) -- packed for the MX kernels (round 6): qkv / fc1 by backbone.pack_a384_mx (TWO images of N * 1536 bytes each, back to back: the planes of the
   * one-wave-per-SIMD kernel, then the chunk images of the two-workgroups-per-CU kernel that runs by default -- csrc/gemm_a384_x3.hip), proj / fc2 by
   * backbone.pack_n384_mx ([K / 64][4 stages][24 KB], csrc/gemm_n384_x3.hip).  With them every block linear from block 1 on (block 0's QKV has no producer
   * of LayerNorm statistics in front of it) computes a w = a_h w_h (fp16 MFMAs) + 2^-12 (a_h8 w_l8 + a_l8 w_h8) (scaled e5m2 MFMAs of K = 64): two thirds of
   * the matrix-pipe time of the bf16 x 3 products; the activations travel between the kernels as the h and l8 planes (h8 is derived in registers).
   * WVN_PREC_FP8, D = 768 (optional, each on its own): qkv_w_mx / proj_w_mx / fc1_w_mx = backbone.pack_a768_fp8 of the e4m3 weight -- with them these three
   * linears run on the A-stationary kernel (csrc/gemm_a768_fp8.hip) from 4096 rows on; fc2_w_mx is not read in this precision */
  const void* qkv_w_mx;
  const void* proj_w_mx;
  const void* fc1_w_mx;
  const void* fc2_w_mx;
} wvn_vit_layer;

typedef struct wvn_vit_model {
  int img_size; /* network input side S (448)            */
  int patch;    /* P (8, 14 or 16)                       */
  int dim;      /* D (384); heads * 64                   */
  int depth;    /* number of blocks (12)                 */
  int heads;    /* h (6); head dim is fixed at 64        */
  int mlp_dim;  /* F (1536)                              */
  int precision;
  int flags;    /* WVN_VIT_* bits */
  const void* patch_w;  /* [D][3*P*P (padded, see above)] conv weight flattened (c, py, px) */
  const float* patch_b; /* [D]                                                      */
  const float* cls_pos; /* [D]  = cls_token + pos_embed[0]                          */
  const float* pos;     /* [1+G*G][D] position table already resampled to the grid  */
  const float *norm_g, *norm_b;
  wvn_vit_layer layers[WVN_MAX_DEPTH];
} wvn_vit_model;

size_t wvn_vit_workspace_bytes(const wvn_vit_model* m, int batch);

/* img [B,3,S,S] fp32 in [0,1] (already resized/cropped, dino_interface.py:54-57) ->
 *   tokens_f32  [B, G*G, D]  final-LayerNorm'ed patch tokens (class token dropped), may be NULL
 *   tokens_lowp [B*G*G rows, ld_lowp] same values in the model precision (bf16 / fp16 / f32), may be NULL (must be NULL for
 *               WVN_PREC_X3: split the fp32 tokens with wvn_split_planes where planes are needed)
 * The workspace needs no initialisation (padding rows are reset by every call). */
int wvn_vit_forward(const wvn_vit_model* m, const float* img, int batch, float* tokens_f32, void* tokens_lowp,
                    int ld_lowp, void* workspace, size_t workspace_bytes, void* stream);

/* Same, for raw 8-bit frames: img [B,3,S,S] uint8 (4-byte aligned).  x/255 is fused into the patch gather, bit-identical
 * to wvn_vit_forward on img.float()/255 (what quick_start.py:160-161 and ros_converter.py:113-126 hand the reference),
 * with a quarter of the upload/HBM bytes.  bf16 / fp16 / fp8 models with patch 8 only (WVN_ERR_ARG otherwise; the general
 * form is wvn_vit_forward_frames). */
int wvn_vit_forward_u8(const wvn_vit_model* m, const unsigned char* img, int batch, float* tokens_f32, void* tokens_lowp,
                       int ld_lowp, void* workspace, size_t workspace_bytes, void* stream);

/* Frame ingest fused into the patch gather: frames [B,3,src_h,src_w] (fp32 in [0,1], or raw uint8 when
 * frames_u8) straight from the camera; network pixel (y, x) of frame b is frame pixel (rows[y], cols[x]).  rows / cols: device
 * int32 [S] -- the composition of T.Resize(input_size, NEAREST) and T.CenterCrop(input_size) (dino_interface.py:52-59,
 * stego_interface.py:51-58) or of ImageProjector.resize_image (image_projector.py:56-59, 199-200) as index tables, built once
 * per camera geometry by the host layer (feature_extractor/transforms.py).  Bit-identical to wvn_vit_forward on the resized /
 * cropped image; the resized image never exists.  Every precision and patch size. */
int wvn_vit_forward_frames(const wvn_vit_model* m, const void* frames, int frames_u8, int src_h, int src_w, const int* rows,
                           const int* cols, int batch, float* tokens_f32, void* tokens_lowp, int ld_lowp, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The same for the frames AND their mirror images in ONE launch sequence of 2 * batch frames (the flip pass of the upstream
 * Stego.get_code, stego_interface.py:91): frame batch + i of the outputs is frame i gathered through cols_mirror (the reversed
 * column table).  tokens_f32 [2 * batch, G*G, D]; the workspace is that of wvn_vit_workspace_bytes(m, 2 * batch).  Bit-identical to two
 * calls of wvn_vit_forward_frames, and faster than them where the persistent block kernels run: twice the rows per launch. */
int wvn_vit_forward_frames_pair(const wvn_vit_model* m, const void* frames, int frames_u8, int src_h, int src_w, const int* rows,
                                const int* cols, const int* cols_mirror, int batch, float* tokens_f32, void* tokens_lowp, int ld_lowp,
                                void* workspace, size_t workspace_bytes, void* stream);
/* The same gather as an image op (ImageProjector.resize_image, image_projector.py:199-200, whose result the callers also
 * display): out [planes, out_h, out_w] = in [planes, src_h, src_w][.., rows[y], cols[x]]; elem_bytes 1 (uint8) or 4 (fp32, int32). */
int wvn_resize_nearest_crop(const void* in, void* out, long long planes, int src_h, int src_w, const int* rows, const int* cols,
                            int out_h, int out_w, int elem_bytes, void* stream);

/* Per-kernel-category HIP-event timing of wvn_vit_forward (bench.py roofline leg).  Categories:
 * 0 patchify 1 patch_gemm 2 layernorm 3 qkv_gemm 4 attention 5 proj_gemm 6 fc1_gemm 7 fc2_gemm */
#define WVN_PROF_NCAT 8
int wvn_prof_enable(int on);
int wvn_prof_collect(double* ms_by_cat_host, long long* launches_by_cat_host); /* syncs recorded events, resets */

/* ---------------------------------------------------------------------------------------------
 * Building blocks, exported so tests/ can check each kernel against the oracle.
 * ------------------------------------------------------------------------------------------- */
/* LayerNorm + QKV projection in one launch: x [M,ldx] fp32 (rows b * ntok_s + t) -> q, k [B*heads][npad][64] and
 * v^T [B*heads][64][npad] (token order of wvn_attention_bf16) in bf16; W [3*heads*64][384] bf16 in qkv.weight row order, q rows
 * multiplied by q_scale before rounding (0 = 1).  heads == 6, ntok_s % 16 == 0, M % 16 == 0, npad % 16 == 0. */
int wvn_qkv_fused(const float* x, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const void* W, const float* bias,
                  void* q, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, int M, void* stream);
/* Attention projection + block MLP in one launch: x += (attn [M,lda] bf16 * Wp[384,384]^T + bp) (* ls1), then the block MLP of
 * wvn_mlp_fused with the LayerNorm inside on the updated rows.  What wvn_vit_forward uses where WVN_VIT_MLP_FUSED applies: the
 * projection's separate pass over the residual stream disappears (the updated rows stay in registers for the LayerNorm). */
int wvn_proj_mlp_fused(const void* attn, int lda, const void* Wp, const float* bp, const float* ls1, const float* ln_g,
                       const float* ln_b, float ln_eps, const void* W1, const float* b1, const void* W2p, const float* b2,
                       const float* ls2, float* x, int ldx, int M, int F, void* stream);
/* The same result (no LayerScale) with the residual rows RESIDENT in the accumulator registers of the kernel for the whole block:
 * x is read once and written once (wvn_proj_mlp_fused: twice and twice), the LayerNorm reads the rows where the projection MFMAs
 * left them, and fc2 accumulates on top of them.  W1p = fc1.weight with bits 2 and 3 of its column index swapped inside every
 * aligned group of 16 (wvn_vit_layer.fc1_w_fused), W2p as above.  Differs from wvn_proj_mlp_fused by fp32 summation order only
 * (the products are accumulated onto x + bias instead of being added to it at the end). */
int wvn_proj_mlp_resident(const void* attn, int lda, const void* Wp, const float* bp, const float* ln_g, const float* ln_b,
                          float ln_eps, const void* W1p, const float* b1, const void* W2p, const float* b2, float* x, int ldx, int M,
                          int F, const float* next_ln_g, const float* next_ln_b, float next_ln_eps, void* xn_next, void* stream);
/* xn_next != NULL: the kernel also applies LayerNorm(next_ln_g, next_ln_b, next_ln_eps) -- blocks.(i+1).norm1 -- to the finished rows
 * and writes the result as MFMA operand fragments: fragment (32-row group R, k-step s) = 1 KB at xn_next + (R * 24 + s) * 1024
 * bytes, lane l's 16 bytes at l * 16 = the 8 values of row 32 R + (l & 31) at columns 16 s + 4 (l >> 5) + {0..3} and
 * 16 s + 8 + 4 (l >> 5) + {0..3}; (M + 31) / 32 * 24 KB.  wvn_qkv_prenorm is the QKV projection that starts from them (Wperm =
 * qkv.weight with bits 2 and 3 of its column index swapped, wvn_vit_layer.qkv_w_fused): = wvn_qkv_fused on the same rows up to
 * fp32 summation order.  b2 then joins the rows as two operand-format terms through the matrix pipe, like bp. */
int wvn_qkv_prenorm(const void* xn_frag, const void* Wperm, const float* bias, void* q, void* k, void* vt, int heads, int npad, int ntok_s,
                    float q_scale, int M, void* stream);
/* Block MLP in one launch: x [M,ldx] fp32 += gelu(xn [M,lda] bf16 * W1[F,384]^T + b1) * W2[384,F]^T + b2  (optionally
 * times LayerScale ls [384]).  W2p = W2 with the hidden index permuted as WVN_VIT_MLP_FUSED describes.  xn == NULL: the kernel
 * computes xn = LayerNorm(x; ln_g, ln_b, ln_eps) itself (what wvn_vit_forward uses: blocks.i.norm2 never touches memory).
 * D is fixed at 384, F % 64 == 0, F <= 1536.  Same arithmetic as wvn_gemm_bf16(epi 1) followed by wvn_gemm_bf16(epi 4) except
 * the summation order inside an MFMA (and inside the LayerNorm statistics). */
int wvn_mlp_fused(const void* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const void* W1, const float* b1,
                  const void* W2p, const float* b2, const float* ls, float* x, int ldx, int M, int F, void* stream);
/* C = epilogue(A[M,K] * W[N,K]^T + bias).  epi: 0 bf16 out, 1 gelu->bf16, 2 relu->bf16, 3 f32 out,
 * 4 f32 out += (residual), 5 same as 4.  A/W bf16, K % 64 == 0. */
int wvn_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                  int K, int epi, void* stream);
/* fp16-operand twins of the five entries above and of wvn_attention_bf16 below (WVN_PREC_F16: same layouts with fp16 bits, same
 * epilogue codes -- "bf16 out" reads "fp16 out") */
int wvn_gemm_f16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, int epi,
                 void* stream);
int wvn_qkv_fused_f16(const float* x, int ldx, const float* ln_g, const float* ln_b, float ln_eps, const void* W, const float* bias,
                      void* q, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, int M, void* stream);
int wvn_proj_mlp_fused_f16(const void* attn, int lda, const void* Wp, const float* bp, const float* ls1, const float* ln_g,
                           const float* ln_b, float ln_eps, const void* W1, const float* b1, const void* W2p, const float* b2,
                           const float* ls2, float* x, int ldx, int M, int F, void* stream);
int wvn_proj_mlp_resident_f16(const void* attn, int lda, const void* Wp, const float* bp, const float* ln_g, const float* ln_b,
                              float ln_eps, const void* W1p, const float* b1, const void* W2p, const float* b2, float* x, int ldx,
                              int M, int F, const float* next_ln_g, const float* next_ln_b, float next_ln_eps, void* xn_next,
                              void* stream);
int wvn_qkv_prenorm_f16(const void* xn_frag, const void* Wperm, const float* bias, void* q, void* k, void* vt, int heads, int npad,
                        int ntok_s, float q_scale, int M, void* stream);
int wvn_mlp_fused_f16(const void* xn, int lda, const float* ln_g, const float* ln_b, float ln_eps, const void* W1, const float* b1,
                      const void* W2p, const float* b2, const float* ls, float* x, int ldx, int M, int F, void* stream);
int wvn_attention_f16(const void* q, const void* k, const void* vt, void* out, int B, int heads, int ntok, int npad, float scale,
                      void* stream);
/* The same GEMM in exact mode (WVN_PREC_X3): A and W as hi / lo bf16 planes (same leading dimensions), three MFMAs per
 * product.  epi 0-2 write C as hi / lo planes (C, C_lo, bf16 [M, ldc] each; gelu = exact erf form), epi 3-5 write fp32 C
 * (C_lo ignored). */
int wvn_gemm_x3(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo, int ldw, const float* bias,
                void* C, void* C_lo, int ldc, int M, int N, int K, int epi, void* stream);
/* The fp8 building blocks (WVN_PREC_FP8): rows of src (fp32, or bf16 when src_is_bf16) -> e4m3 rows q [rows, ldq] + scale[rows]
 * (= amax / 448, 1 for a zero row);  C = epilogue((A_q W_q^T) sa[m] sw[n] + bias) on v_mfma_scale_f32_32x32x64_f8f6f4 with
 * epi 0 (bf16 out), 1 (gelu -> bf16), 3 (fp32 out), 4 (fp32 +=).  K % 128 == 0, lda / ldw % 16 == 0. */
int wvn_quantize_rows_fp8(const void* src, int src_is_bf16, int lds, void* q, int ldq, float* scale, int rows, int cols,
                          void* stream);
/* LayerNorm (fp32 rows of x [rows, D], D % 64 == 0, biased variance, eps inside the sqrt) fused with the row quantiser: q [rows, ldq]
 * e4m3 + scale [rows] of the NORMALISED rows -- what every fp8 block runs in front of its QKV / fc1 GEMM */
int wvn_layernorm_fp8(const float* x, const float* gamma, const float* beta, void* q, int ldq, float* scale, int rows, int D,
                      float eps, void* stream);
int wvn_gemm_fp8(const void* A_q, int lda, const void* W_q, int ldw, const float* sa, const float* sw, const float* bias,
                 void* C, int ldc, int M, int N, int K, int epi, void* stream);
/* The A-stationary form of the same product for K == 768 (csrc/gemm_a768_fp8.hip; what WVN_PREC_FP8 runs for the QKV, projection and fc1 linears of
 * ViT-Base from 4096 rows on -- the layer's qkv_w_mx / proj_w_mx / fc1_w_mx fields carry the packed weights in that precision): a workgroup keeps 128 rows of
 * A in registers and streams 32-column tiles of the weight, packed by backbone.pack_a768_fp8 as [N / 32][12 k-steps][2 halves][64 lanes][16 B].
 * epi (the numbers of wvn_gemm_fp8): 0 | 1 (C bf16 [M][ldc], 1 = gelu), 4 (C fp32 += ls * (...), ls optional), 7 (q | k | v^T bf16 in the
 * layouts of wvn_attention_bf16, inside one 2 GB span; q scaled by q_scale when it is not 0), 9 (gelu -> C = e4m3 [M][ldc] with MX block scales: one E8M0 byte per
 * (row, 32 columns) written to q [M][N / 32] -- the A operand of wvn_gemm_fp8_mx).  N % 32 == 0; WVN_ERR_ARG for any other shape. */
int wvn_gemm_a768_fp8(const void* A_q, int lda, const void* W_packed, const float* sa, const float* sw, const float* bias, const float* ls, void* C, int ldc,
                      int M, int N, int epi, void* q, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, void* stream);
/* wvn_gemm_fp8 with MX block scales on the A operand: A_q e4m3 [M][lda], a_scales [M][K / 32] E8M0 bytes (value = q * 2^(byte - 127) per 32 consecutive k; what epi 9
 * above writes), W_q with one fp32 scale per row as before; epi 3 (C fp32 =) | 4 (C fp32 += ls * (...)).  K % 128 == 0.  WVN_PREC_FP8 runs ViT-Base's fc2 this way
 * behind the A-stationary fc1 (no row quantiser between them; WVN_NO_FP8_MX=1 in the environment: the bf16 hidden activation and the row quantiser again) */
int wvn_gemm_fp8_mx(const void* A_q, int lda, const void* a_scales, const void* W_q, int ldw, const float* sw, const float* bias, const float* ls,
                    void* C, int ldc, int M, int N, int K, int epi, void* stream);
/* fp32 [rows, lds] -> hi = bf16(x), lo = bf16(x - hi), both [rows, ldd] */
int wvn_split_planes(const float* src, int lds, void* hi, void* lo, int ldd, int rows, int cols, void* stream);
/* exact-mode attention: q / k planes [B,h,npad,64], V^T planes [B,h,64,npad] (token permutation of the bf16 path),
 * out planes [B*ntok, h*64]; scale > 0 */
int wvn_attention_x3(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                     const void* vt_lo, void* out_hi, void* out_lo, int B, int heads, int ntok, int npad, float scale,
                     void* stream);
/* fp32 GEMM C = epilogue(op(A) op(B) + bias); transA: A stored [K,M]; transB: B stored [N,K].
 * epi: 0 none, 1 relu, 2 gelu, 3 C +=, 4 sigmoid on column 0, 5 relu-mask (mask > 0 ? acc : 0). */
int wvn_gemm_f32(const float* A, int lda, int transA, const float* B, int ldb, int transB, const float* bias, float* C,
                 int ldc, int M, int N, int K, int epi, const float* mask, int ldmask, void* stream);
int wvn_layernorm(const float* x, const float* gamma, const float* beta, void* y, int y_is_bf16, int rows, int D,
                  float eps, void* stream);
/* q,k [B,h,npad,64]; v: bf16 path takes V^T [B,h,64,npad], f32 path takes V [B,h,npad,64];
 * out [B*ntok, h*64].  npad % 128 == 0, pad rows must be finite.
 * bf16 path: V^T is stored with the tokens of every aligned group of 16 permuted -- position
 * 16G + 8h + 4a + e holds token 16G + 8a + 4h + e (bits 2 and 3 of the token index swapped; order
 * 0-3, 8-11, 4-7, 12-15) -- so that the 8 keys a half-wave multiplies with are one 16-byte LDS read.
 * The QKV projection epilogues of wvn_vit_forward write this layout.
 * scale > 0: q holds the raw projections, scores = scale * q.k.  scale == 0 (bf16 path only): q is already multiplied by
 * softmax_scale * log2(e) (wvn_vit_forward's QKV epilogue does that before rounding q to bf16); the kernel then feeds the
 * running max into the S^T MFMA chain as its C operand and the accumulators come out as exp2 arguments. */
int wvn_attention_bf16(const void* q, const void* k, const void* vt, void* out, int B, int heads, int ntok, int npad,
                       float scale, void* stream);
int wvn_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int heads, int ntok, int npad,
                      float scale, void* stream);
int wvn_patchify(const float* img, void* patches, int out_is_bf16, int B, int S, int P, void* stream);
int wvn_patchify_u8(const unsigned char* img, void* patches_bf16, int B, int S, int P, void* stream); /* P in {8, 14, 16} */
int wvn_cast_f32_to_bf16(const float* src, void* dst, long long n, void* stream);
/* strided rows: dst[r][c] = (bf16 | fp16 when to_f16) src[r][c], fp32 src with leading dimension lds, 16-bit dst with ldd */
int wvn_cast_rows(const float* src, int lds, void* dst, int ldd, int rows, int cols, int to_f16, void* stream);

/* F.interpolate(features, (H,H), mode="bilinear", align_corners=True) of dino_interface.py:87-90 /
 * stego_interface.py:107: tokens [B,G*G,D] -> dense [B,D,H,H] fp32. */
int wvn_upsample_bilinear(const float* tokens, float* dense, int B, int G, int D, int H, void* stream);
/* F.interpolate(pred[None].float(), (H,H), mode="nearest").int()  (stego_interface.py:108-109) */
int wvn_upsample_nearest_i32(const int* labels, int* out, int B, int G, int H, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Segments
 * ------------------------------------------------------------------------------------------- */
/* FeatureExtractor.sparsify_features (feature_extractor.py:390-396) fused with the bilinear
 * up-sampling that precedes it: feat[b][s][:] = mean over {seg[b]==s} of upsample(tokens[b]).
 * seg [B,H,W] int32 (-1 = ignore), tokens [B,G*G,ld] fp32, feat [B,S,D].
 * scratch_w: B*S*G*G 8-byte words (the tap weights accumulate in 64-bit fixed point: integer atomics are order-independent,
 * so the result is deterministic), scratch_cnt: B*S ints (receives the pixel count per segment). */
int wvn_segpool_bilinear_mean(const int* seg, const float* tokens, int ld, float* feat, void* scratch_w,
                              int* scratch_cnt, int B, int H, int W, int G, int S, int D, void* stream);
/* Same result as wvn_segpool_bilinear_mean for PATCH-ALIGNED segment maps (a [B,G,G] label grid that
 * the caller would nearest-upsample by exactly the patch size: k-means clusters, grid cells that are a
 * multiple of the patch): mean over a segment's patches of a separable 3x3 stencil of the patch map.
 * wy, wx: [G][3] fp32 stencil weights for offsets -1,0,+1 (per-patch-row sums of the align_corners
 * tap weights / P).  Deterministic, no atomics.  labels [B,G*G] int32, feat [B,S,D]. */
int wvn_segpool_patch_labels(const int* labels, const float* tokens, int ld, const float* wy, const float* wx,
                             float* feat, int B, int G, int S, int D, void* stream);
/* FeatureExtractor.sparsify_features on an explicit pixel-resolution map (no resampling):
 * tokens [B,P,D] pixel-major, seg [B,P] -> feat [B,S,D] = per-segment mean (0/0 = NaN). S <= 236.  Deterministic (fixed
 * summation order, no atomics); scratch: wvn_segmean_scratch_bytes() bytes; scratch_cnt [B*S] receives the pixel counts. */
size_t wvn_segmean_scratch_bytes(int B, int P, int S, int D);
int wvn_segmean_tokens(const int* seg, const float* tokens, float* feat, int* scratch_cnt, void* scratch, size_t scratch_bytes,
                       int B, int P, int S, int D, void* stream);
/* MissionNode.update_supervision_signal (traversability_estimator/nodes.py:400-440).
 * mask [C,H,W] fp32 (NaN = unlabeled), seg [H,W] int32 -> signal [S] fp32, valid [S] uint8.
 * scratch_sum: S 8-byte words (2^-32 fixed-point sums: deterministic), scratch_cnt: S ints. */
int wvn_label_pool(const float* mask, int C, const int* seg, float* signal, unsigned char* valid, void* scratch_sum,
                   int* scratch_cnt, int H, int W, int S, void* stream);
/* The same for n nodes in ONE launch pair (TraversabilityEstimator.add_supervision_node re-pools every mission node in
 * range, traversability_estimator.py:287-289).  nodes_dev: DEVICE array of n records; scratch_sum: n*Smax 8-byte words,
 * scratch_cnt: n*Smax ints; all masks [C,H,W], all segment maps [H,W]. */
typedef struct wvn_label_pool_node {
  const float* mask;    /* [C][H][W] */
  const int* seg;       /* [H][W]    */
  float* signal;        /* [S] out   */
  unsigned char* valid; /* [S] out   */
  int S;
  int reserved;
} wvn_label_pool_node;
int wvn_label_pool_batched(const wvn_label_pool_node* nodes_dev, int n, int C, int H, int W, int Smax, void* scratch_sum,
                           int* scratch_cnt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Supervision masks: ImageProjector.project_and_render (image_projector.py:126-197: pinhole projection +
 * kornia draw_convex_polygon) fused with the torch.fmin merge of traversability_estimator.py:281-286, for n mission nodes in
 * one launch.  Per node: K [4][4] (the projector's scaled camera matrix), pose [4][4] = pose_cam_in_world, mask [C][H][W]
 * updated IN PLACE (inside the projected polygon: fmin(old, value); elsewhere untouched = fmin(old, NaN)), projected [npts][2]
 * + depth [npts] (optional outs; the polygon itself is drawn with the behind-the-camera vertices set to NaN, :180).  points: [npts][3] world coordinates shared by all nodes, or
 * [n][npts][3] when points_batched.  value = colour * traversability, read from value_dev[0] if non-NULL (no host sync when
 * the traversability lives on the GPU).  npts: any count whose vertex arrays fit the 64 KB of LDS beside the 2*H row limits (H = 448: 7700; the
 * untraversable plane of SupervisionNode.make_footprint_with_node has 1000).  Scan lines touched by a NaN edge stay unfilled (torch min / max).  Arithmetic order: csrc/supervision.hip header.
 * ------------------------------------------------------------------------------------------- */
typedef struct wvn_render_node {
  const float* K;
  const float* pose;
  float* mask;
  float* projected; /* [npts][2] out, optional: the RAW pinhole coordinates of every point (finite also behind the camera, as
                       ImageProjector.project returns them, image_projector.py:126-150) */
  float* depth;     /* [npts] out, optional: camera-frame z of every point (valid_z = depth >= 0, check_validity :112-124) */
} wvn_render_node;
int wvn_project_render_fmin(const wvn_render_node* nodes_dev, int n, const float* points, int points_batched, int npts, int C,
                            int H, int W, const float* value_dev, float value, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SLIC superpixels (FeatureExtractor default segmentation_type, feature_extractor.py:84-90,221-225: fast_slic on the CPU in
 * the reference).  Integer-arithmetic SLIC (csrc/slic.hip): img [3][H][W] uint8 or float in [0,1]; labels [H][W] int32 in
 * [0, wvn_slic_num_clusters()).  lut_lin [256], lut_f [4096]: sRGB->linear and CIELAB f(t) tables (device, int32; built by the
 * caller in double precision: wild_visual_navigation_amd/ops.py slic_tables).  Parity with fast_slic is unpinned.
 * ------------------------------------------------------------------------------------------- */
int wvn_slic_num_clusters(int H, int W, int num_components);
size_t wvn_slic_scratch_bytes(int H, int W, int num_components);
int wvn_slic(const void* img, int img_is_u8, int H, int W, int num_components, float compactness, int iters, const int* lut_lin,
             const int* lut_f, int* labels, void* scratch, size_t scratch_bytes, void* stream);
/* SegmentExtractor.centers (segment_extractor.py:70-92): centers [S,2] fp32 (x,y). scratch: 3*S u64. */
int wvn_seg_centers(const int* seg, float* centers, void* scratch, int H, int W, int S, void* stream);
/* SegmentExtractor.adjacency_list (segment_extractor.py:39-67): edges [max_edges,2] int64 sorted by
 * key = left + right*S, *count = E (device int).  scratch_bitmap: S*S bytes. */
int wvn_seg_adjacency(const int* seg, long long* edges, int* count, unsigned char* scratch_bitmap, int H, int W, int S,
                      int max_edges, void* stream);

/* ---------------------------------------------------------------------------------------------
 * STEGO head + clustering  (stego.stego.Stego.get_code / postprocess, stego_interface.py:91-100)
 * ------------------------------------------------------------------------------------------- */
/* rows of code [rows, ldc] -> xn [rows, C] = code / max(||code||, 1e-12), sequential fp32 */
int wvn_normalize_rows(const float* code, int ldc, float* xn, int rows, int C, void* stream);
/* out[r] = argmax over the cols of row r (lowest index wins ties): label maps of the STEGO cluster probe (cosine similarity
 * against the checkpoint's learned centroids, run_clustering=False) and linear probe (stego_interface.py:94-100) */
int wvn_argmax_rows(const float* x, int ld, int rows, int cols, int* out, void* stream);
/* deterministic cosine k-means per image on xn [B,P,C]; labels [B,P] int32; nseg [B] distinct ids;
 * relabel != 0 compacts ids to 0..K'-1 ascending (feature_extractor.py:245-246). C in {16,64,90}.
 * scratch: wvn_kmeans_scratch_bytes(B,P,C,K) bytes (centroids + per-chunk partial sums); on return its first B*K*C floats hold the
 * final centroids [B][K][C] (both forms). */
/* Summation order of the centroid update (both forms): chunks of 64 consecutive points, members of a cluster in ascending point
 * order; chunk partials in ascending order inside groups of 8 chunks; group partials in ascending order; every chain starts from
 * +0.  Normalisation: x * (1 / max(||x||, 1e-12)) with one correctly rounded reciprocal per row. */
size_t wvn_kmeans_scratch_bytes(int B, int P, int C, int K);
/* The same k-means over the H x H code PIXELS of every frame (stego_interface.py:94-109 read as "postprocess clusters the
 * up-sampled code"): code [B, G*G, C] fp32 patch codes (NOT normalised); the points are the rows of
 * normalise(F.interpolate(code, (H,H), bilinear, align_corners=True)), re-created on the fly in every pass from the L2-resident
 * patch codes -- the [B, H*H, C] array (4.6 GB per 64 frames at 448^2) never exists.  labels [B, H*H] int32.  Bit-identical to
 * wvn_upsample_bilinear + wvn_normalize_rows + wvn_kmeans_cosine on the materialised rows.  C in {16, 90}. */
size_t wvn_kmeans_pixels_scratch_bytes(int B, int G, int H, int C, int K);
int wvn_kmeans_cosine_pixels(const float* code, int* labels, int* nseg, void* scratch, int B, int G, int H, int C, int K, int iters,
                             int relabel, void* stream);
/* The pixel-resolution k-means through its LINEARITY (round 5; the form StegoInterface uses, stego_interface.py:94-109): the same
 * clustering -- same points, same initial centroids, same iteration count, same tie rule -- with the assignment taken from a per-pass
 * similarity table S = code . c^T interpolated per pixel (rinv_p > 0 moves no argmax, <., c_k> is linear in the four taps) and the
 * centroid sums from per-(cluster, patch) summed tap weights times the patch codes (SURVEY.md 8(a7)'s pooling identity): ~1/20 of the
 * arithmetic of wvn_kmeans_cosine_pixels.  Every summation order is fixed (csrc/stego_linear.hip; oracle/kmeans_linear.py states them), so
 * labels and centroids are reproducible bit for bit; against wvn_kmeans_cosine_pixels they differ only at pixels whose two best
 * similarities are closer than fp32 rounding.  C in {16, 90}, K <= 32 (wvn_kmeans_pixels_linear_supported_shape); scratch as above: its
 * first B*K*C floats hold the final centroids on return. */
int wvn_kmeans_pixels_linear_supported_shape(int G, int H, int C, int K);
size_t wvn_kmeans_pixels_linear_scratch_bytes(int B, int G, int H, int C, int K);
int wvn_kmeans_cosine_pixels_linear(const float* code, int* labels, int* nseg, void* scratch, int B, int G, int H, int C, int K,
                                    int iters, int relabel, void* stream);
/* The same with the OTHER reading of how the code is interpolated before it is clustered (round 6): align_corners = 0 takes ATen's half-pixel taps
 * (F.interpolate(code, size, mode="bilinear", align_corners=False): what the public STEGO evaluation code uses) instead of the align_corners=True
 * taps WVN's own later up-sample uses (stego_interface.py:107).  The reference hands postprocess() to an absent package (stego_interface.py:94-100), so
 * which one runs upstream cannot be decided here: StegoInterface(code_align_corners=...) selects it, oracle/kmeans_linear.py states both, the GPU is
 * bit-exact against either.  align_corners != 0: identical to wvn_kmeans_cosine_pixels_linear. */
int wvn_kmeans_cosine_pixels_linear_ac(const float* code, int* labels, int* nseg, void* scratch, int B, int G, int H, int C, int K,
                                       int iters, int relabel, int align_corners, void* stream);
/* labels[b][y][x] (int32, [B, H, H]) = argmax over k < K (lowest k wins ties) of the bilinear interpolation (align_corners=True, the
 * fixed operation order of wvn_upsample_bilinear) of table[b][.][k] to H x H: the STEGO cluster probe / linear probe applied at PIXEL
 * resolution (stego_interface.py:94-100, 107-109: postprocess acts on the up-sampled code; a probe is linear in the code and the
 * interpolation weights sum to one, so interpolating its K outputs per patch equals evaluating it on the interpolated code).
 * table: [B, G*G, wvn_table_argmax_slots(K)] fp32, 16-byte aligned, slots >= K ignored; K <= 32. */
int wvn_table_argmax_slots(int K);
int wvn_table_bilerp_argmax(const float* table, int* labels, int B, int G, int H, int K, void* stream);
/* the same with the taps of align_corners (0: half-pixel coordinates; see wvn_kmeans_cosine_pixels_linear_ac) */
int wvn_table_bilerp_argmax_ac(const float* table, int* labels, int B, int G, int H, int K, int align_corners, void* stream);
/* out = 0.5 * (a + flip_x(mirrored)) on [B, G, G, C] fp32 patch maps: the code of a frame averaged with the flipped-back code of
 * its mirror image (the second pass of the upstream Stego.get_code; the mirror pass itself is wvn_vit_forward_frames with a
 * reversed column table).  out may alias a. */
int wvn_flip_average(const float* a, const float* mirrored, float* out, int B, int G, int C, void* stream);
int wvn_kmeans_cosine(const float* xn, int* labels, int* nseg, void* scratch, int B, int P, int C, int K, int iters,
                      int relabel, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Traversability MLP  (model/simple_mlp.py:10-39, utils/loss.py:93-160,
 * utils/confidence_generator.py:78-82,182-193, traversability_estimator.py:100,464-477)
 * Parameters live in ONE flat fp32 buffer [W1 | b1 | W2 | b2 | W3 | b3], Wi in Linear layout;
 * gradients / Adam moments use the same layout (so the data-parallel exchange is one all-reduce).
 * ------------------------------------------------------------------------------------------- */
typedef struct wvn_mlp_desc {
  int D;  /* input features (384 dino / 90 stego) */
  int H1; /* 256 */
  int H2; /* 32  */
  int reserved;
} wvn_mlp_desc;
size_t wvn_mlp_param_count(const wvn_mlp_desc* d);
size_t wvn_mlp_workspace_bytes(const wvn_mlp_desc* d, int rows);

/* SimpleMLP.forward: x [R,D] -> out [R,1+D] (sigmoid on column 0).  h1/h2 (post-ReLU) may be NULL
 * for inference, in which case they are carved from the workspace. */
int wvn_mlp_forward(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, int R, float* out, float* h1,
                    float* h2, void* workspace, size_t workspace_bytes, void* stream);
/* One optimisation step, split at the two points where data-parallel ranks exchange:
 *  A: forward + per-row reconstruction loss + local statistic  stats[4] (double) =
 *     { n_labelled, sum loss_reco, sum loss_reco^2, R }                    -> all-reduce(sum) stats
 *  B: loss gradient + backward: grads[param_count + 2] (last two = sum weighted trav loss, sum raw
 *     trav loss)                                                            -> all-reduce(sum) grads
 *  C: Adam (lr, betas (0.9,0.999), eps 1e-8) + losses[5] = {total, trav, reco, conf mean, conf std} */
int wvn_mlp_train_phase_a(const wvn_mlp_desc* d, const float* params, const float* x, int ldx,
                          const unsigned char* y_valid, int R, double* stats, void* workspace, size_t workspace_bytes,
                          void* stream);
int wvn_mlp_train_phase_b(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, const float* y,
                          const unsigned char* y_valid, int R, const double* stats, float std_factor, float w_trav,
                          float w_reco, float* grads, float* confidence_out, void* workspace, size_t workspace_bytes,
                          void* stream);
int wvn_mlp_train_phase_c(const wvn_mlp_desc* d, float* params, const float* grads, float* adam_m, float* adam_v,
                          int step, float lr, const double* stats, float w_trav, float w_reco, float* losses,
                          void* stream);
/* The step on a COMPACTED batch whose row count lives in device memory (no host synchronisation between segmentation and
 * training): R rows are passed, the first *rows_dev of them are real; the others contribute nothing to statistics, losses or
 * gradients (rows_dev == NULL: all R rows, = the entry points above).  wvn_compact_segment_rows builds such a batch from the
 * pooled features of a frame batch: feat [B][S][D] (+ optional per-row side data [B][S][Dside], e.g. labels), nseg [B] = number
 * of segments that exist per frame -> x_out [B*S][D] / side_out [B*S][Dside] holding the rows (b, s < nseg[b]) front to back in
 * (b, s) order, zeros behind them, and *rows_dev = sum(nseg).  Replaces feat[keep] (traversability_estimator.py:432-446 builds
 * the batch from nodes that all exist; a batched extractor has to drop the ids a frame did not produce). */
int wvn_compact_segment_rows(const float* feat, int D, const float* side, int Dside, const int* nseg, int B, int S, float* x_out,
                             float* side_out, int* rows_dev, void* stream);
/* sync_word (phase A) != NULL and fused (phase B) != 0 select the FOUR-LAUNCH step for the SimpleMLP geometry (H1 = 256, H2 = 32, R <=
 * 2048; csrc/mlp_train.hip): phase A = one launch (all three layers, row losses, statistic -- the last row tile to arrive folds the
 * per-tile partials in tile order; sync_word: a device word that is ZERO on first use and is left at zero), phase B = two launches
 * (gradient seed + dL/dh2 + dL/dh1; the three weight / bias gradients and the loss sums), phase C = one launch (Adam + losses)
 * against 17-20 launches of the general path.  Both phases of a step must take the same path and the same workspace; other
 * geometries fall back to the general path whatever the flags say.  Fixed summation orders: bit-reproducible. */
int wvn_mlp_train_phase_a_rows(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, const unsigned char* y_valid,
                               int R, const int* rows_dev, double* stats, void* workspace, size_t workspace_bytes,
                               unsigned int* sync_word, void* stream);
int wvn_mlp_train_phase_b_rows(const wvn_mlp_desc* d, const float* params, const float* x, int ldx, const float* y,
                               const unsigned char* y_valid, int R, const int* rows_dev, const double* stats, float std_factor,
                               float w_trav, float w_reco, float* grads, float* confidence_out, void* workspace,
                               size_t workspace_bytes, int fused, void* stream);
/* quick_start.py:194-210 / loss.py:162-164: trav[r] = out[r][0], conf[r] = confidence(mse(out[r][1:], x[r])) */
int wvn_mlp_confidence(const float* out, int ldo, const float* x, int ldx, float mean, float std, float std_factor,
                       float* trav, float* conf, int R, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-pixel traversability inference: the live node's per-frame path with prediction_per_pixel
 * (wvn_feature_extractor_node.py:319-363, quick_start.py:183-210):
 *   dense = bilinear(align_corners) upsample of the patch tokens to [out_h,out_w]  (dino_interface.py:87-90)
 *   out   = SimpleMLP(dense rows); trav = out[:,0] (after its sigmoid); conf = confidence(mse(out[:,1:], dense))
 * without building the dense tensor (308 MB / frame at 448^2) and with layer 1 evaluated at token resolution
 * (csrc/pixel_mlp.hip).  bf16 MFMA operands, fp32 accumulation: the speed mode; the exact mode is
 * wvn_upsample_bilinear + wvn_mlp_forward + wvn_mlp_confidence (or the exact-mode form below).  H1 = 256, H2 = 32 and D = 384 (DINO
 * ViT-S features) or D = 90 (STEGO code, the live node's default feature_type); 0 / WVN_ERR_ARG otherwise.
 *
 * packed : wvn_pixel_mlp_pack_bytes() bytes, rebuilt by wvn_pixel_mlp_pack whenever the parameters change
 *          (the node reloads them at 1 Hz, wvn_feature_extractor_node.py:407-432).
 * zx     : [batch*grid*grid rows][ldzx >= wvn_pixel_mlp_zx_cols()] bf16 (640 for D = 384, 384 for D = 90).  Columns from 256 on
 *          hold the features on entry (D = 384: hand wvn_vit_forward tokens_lowp = zx + 256, ld_lowp = ldzx; D = 90: the 90
 *          code values followed by ZEROS up to column 384); columns [0,256) are scratch (layer-1 pre-activations).
 * trav / conf / loss_reco : [batch][out_h][out_w] fp32, each may be NULL.  mean/std/std_factor: ConfidenceGenerator state;
 * conf_state (may be NULL): the same three floats in DEVICE memory, read by the kernel instead of the scalars -- lets the call
 * sit in a captured HIP graph while the confidence statistics keep moving.
 * Requires 15*(grid-1)/(out-1) < 2 in both directions (out >= ~7.5 x grid: 224/28, 448/56 ...), WVN_ERR_ARG otherwise.
 * ------------------------------------------------------------------------------------------- */
#define WVN_PIXEL_ZX_COLS 640
#define WVN_PIXEL_X_COL 256
size_t wvn_pixel_mlp_pack_bytes(const wvn_mlp_desc* d);
int wvn_pixel_mlp_zx_cols(const wvn_mlp_desc* d);
int wvn_pixel_mlp_pack(const wvn_mlp_desc* d, const float* params, void* packed, void* stream);
int wvn_pixel_mlp_infer(const wvn_mlp_desc* d, const void* packed, void* zx, int ldzx, int batch, int grid, int out_h,
                        int out_w, float mean, float std, float std_factor, const float* conf_state, float* trav,
                        float* conf, float* loss_reco, void* stream);

/* Exact-mode form of the same fused kernel: every MFMA operand is split into hi + lo bf16 parts and every product is formed
 * as hi*hi + hi*lo + lo*hi (fp32 accumulation), the token-resolution layer-1 GEMM runs on the fp32 FMA path: results agree
 * with the fp32 reference sequence to ~1e-5 relative (the 1e-3 bar of the exact mode), at 2.3x the MFMA work of the bf16
 * form.  tokens: [batch*grid*grid][ld_tokens >= D] fp32 features (D = 384: wvn_vit_forward tokens_f32; D = 90: the STEGO code);
 * params: the flat fp32 parameter buffer (W1 is read from it); packed: wvn_pixel_mlp_exact_pack_bytes() bytes from
 * wvn_pixel_mlp_exact_pack; workspace: wvn_pixel_mlp_exact_workspace_bytes() bytes, no initialisation needed. */
size_t wvn_pixel_mlp_exact_pack_bytes(const wvn_mlp_desc* d);
size_t wvn_pixel_mlp_exact_workspace_bytes(const wvn_mlp_desc* d, int batch, int grid);
int wvn_pixel_mlp_exact_pack(const wvn_mlp_desc* d, const float* params, void* packed, void* stream);
int wvn_pixel_mlp_infer_exact(const wvn_mlp_desc* d, const float* params, const void* packed, const float* tokens,
                              int ld_tokens, int batch, int grid, int out_h, int out_w, float mean, float std,
                              float std_factor, const float* conf_state, float* trav, float* conf, float* loss_reco,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A -> B wire format: the per-frame wild_visual_navigation_msgs/ImageFeatures message built at
 * wvn_feature_extractor_node.py:373-393 (seg.cpu().numpy().astype(np.int32) + feat.cpu().numpy().flatten().tolist()) and
 * decoded at wvn_learning_node.py:651-656.  wvn_wire_pack lays the frame out on the device as the message carries it:
 *   [ 64-byte header {magic "WVNF", version, H, W, S, D, seg_offset, feat_offset} | int32 seg[H*W] | float32 feat[S*D] ]
 * (the two payload sections are the byte arrays of Image.data, step 4*W, and Float32MultiArray.data): one contiguous
 * device->host copy per frame instead of two copies, a host cast and a Python list.  wvn_wire_unpack is the inverse on the
 * learner's GPU (segments as int64, what MissionNode stores, and/or int32).  out / in: 16-byte aligned, wvn_wire_bytes() bytes.
 * ------------------------------------------------------------------------------------------- */
size_t wvn_wire_bytes(int H, int W, int S, int D);
int wvn_wire_pack(const void* seg, int seg_is_i64, const float* feat, int ldf, void* out, int H, int W, int S, int D, void* stream);
int wvn_wire_unpack(const void* in, long long* seg_i64, int* seg_i32, float* feat, int H, int W, int S, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Instrumentation (scripts/a384_timing.py, scripts/attn_timing.py): in-kernel s_memtime phase timings of
 * the two MFMA kernels.  Not part of the drop-in surface.
 * ------------------------------------------------------------------------------------------- */
/* wvn_gemm_bf16 with per-wave phase counters: dbg[(workgroup * 8 + wave) * 4 + {0 wait+barrier, 1 MFMA block,
 * 2 epilogue part, 3 total}] in shader cycles (K == 384 shapes only; dbg may be NULL). */
int wvn_debug_gemm_bf16_timed(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                              int N, int K, int epi, long long* dbg, void* stream);
/* The A-stationary split-operand GEMM for K = 384 (csrc/gemm_a384_x3.hip: what WVN_PREC_MIX / WVN_PREC_X3 run for QKV, projection
 * and fc1 from 8192 rows on), callable on its own: A / W as hi + lo bf16 planes ([M][384], [N][384]; the lo planes at A_lo / W_lo,
 * W_lo behind W in one allocation), epi = 1 (erf GELU, hi / lo bf16 planes C / C_lo [M][ldc]) or 4 (fp32 C [M][ldc] += result).
 * dbg != NULL: the instrumented build writes dbg[(workgroup * 4 + wave) * 4 + {0 wait + barrier, 1 DMA issue, 2 MFMA steps with the
 * epilogue chunks, 3 total}] in shader cycles (scripts/bench_a384_x3.py).  WVN_ERR_ARG when the shape is not eligible. */
int wvn_debug_gemm_a384_x3(const void* A, const void* A_lo, int lda, const void* W, const void* W_lo, const float* bias, void* C,
                           void* C_lo, int ldc, int M, int N, int epi, long long* dbg, void* stream);
/* The row-panel split-operand GEMM for N = 384 residual updates (csrc/gemm_n384_x3.hip: fc2 and the attention projection of
 * WVN_PREC_MIX / WVN_PREC_X3 from 8192 rows on): C [M][ldc] fp32 += (A W^T + bias) (* ls), A [M][lda] / W [384][K] as hi + lo bf16 planes
 * (lo planes behind the hi planes in one allocation each), K % 32 == 0.  dbg != NULL: the instrumented build writes
 * dbg[(workgroup * 4 + wave) * 4 + {0 wait + barrier, 1 k-steps, 2 epilogue, 3 total}] in shader cycles. */
int wvn_debug_gemm_n384_x3(const void* A, const void* A_lo, int lda, const void* W, const void* W_lo, const float* bias, const float* ls,
                           float* C, int ldc, int M, int K, long long* dbg, void* stream);
/* The split-operand block MLP with the fragment-major hand-over (what WVN_PREC_MIX / WVN_PREC_X3 run at D = 384 from 8192 rows on):
 * hid = gelu(xn W1^T + b1) written by csrc/gemm_a384_x3.hip as MFMA operand fragments (hi / lo planes, ceil(M / 32) * 32 rows of F each),
 * x [M][384] fp32 += hid W2^T + b2 by csrc/gemm_n384_x3.hip; xn / W1: hi + lo bf16 planes, W2p: wvn_vit_layer.fc2_w_fused in its
 * WVN_PREC_X3 / WVN_PREC_MIX layout.  dbg1 / dbg2 (optional): per-wave cycle counters of the two instrumented builds. */
int wvn_debug_mlp_x3_frag(const void* xn, const void* xn_lo, const void* W1, const void* W1_lo, const float* b1, void* hid, void* hid_lo,
                          const void* W2p, const float* b2, float* x, int M, int F, long long* dbg1, long long* dbg2, void* stream);
/* The MX form of the row-panel kernel (round 6; what WVN_PREC_MIX runs for fc2 and the attention projection by default): every operand as
 * an fp16 value h, the e5m2 image l8 of its rounding residue * 2^12 and the e5m2 image h8 of the value; the product = A_h W_h on fp16 MFMAs
 * + 2^-12 (A_h8 W_l8 + A_l8 W_h8) on scaled 8-bit MFMAs of K = 64 (csrc/gemm_n384_x3.hip: gemm_n384_mx_pair_kernel).  A_h: fragment-major
 * fp16 [ceil(M / 32)][K / 16][64 lanes][8]; A_l8: [ceil(M / 32)][K / 64][2 halves][64 lanes][16 bytes] (backbone.mx_fragments states
 * the element order), behind A_h inside one 4 GB span; A_h8 is ignored (may be NULL): the kernel derives e5m2(A_h) in registers; Wp: backbone.pack_n384_mx(W [384][K]); C [M][ldc] fp32 += (A W^T + bias) (* ls);
 * K % 128 == 0.  dbg as wvn_debug_gemm_n384_x3. */
int wvn_debug_gemm_n384_mx(const void* A_h, const void* A_l8, const void* A_h8, const void* Wp, const float* bias, const float* ls, float* C,
                           int ldc, int M, int K, long long* dbg, void* stream);
/* The block MLP of WVN_PREC_MIX in its MX form: hid = gelu(LayerNorm(x) W1^T + b1) by csrc/gemm_a384_x3.hip (LayerNorm formed on load from
 * ln_stats[m] = {mean, 1 / sqrt(var + eps)}; W1p = backbone.pack_a384_mx(fc1.weight)), written as the MX operand planes hid_h / hid_l8 (hid_h8 is
 * ignored and may be NULL) (fragment-major, ceil(M / 32) * 32 rows of F), then xout [M][384] fp32 += hid W2^T + b2 by the MX row-panel kernel (W2p =
 * backbone.pack_n384_mx(fc2.weight)).  W2p == NULL: fc1 only.  dbg1 / dbg2: per-wave cycle counters of the instrumented builds. */
int wvn_debug_mlp_mx(const float* x, int ldx, const float* ln_stats, const float* ln_g, const float* ln_b, const void* W1p, const float* b1,
                     void* hid_h, void* hid_l8, void* hid_h8, const void* W2p, const float* b2, float* xout, int M, int F, long long* dbg1,
                     long long* dbg2, void* stream);
/* LayerNorm-on-load + QKV in the MX form: q (pre-scaled by q_scale; q_lo != NULL: its fp16 rounding residue as a second plane) | k | v^T fp16 in
 * the layouts of wvn_attention_f16 (Wp = backbone.pack_a384_mx(qkv.weight)). */
int wvn_debug_qkv_mx(const float* x, int ldx, const float* ln_stats, const float* ln_g, const float* ln_b, const void* Wp, const float* bias, void* q,
                     void* q_lo, void* k, void* vt, int heads, int npad, int ntok_s, float q_scale, int M, long long* dbg, void* stream);
/* subsequent wvn_attention_bf16 launches write dbg[(workgroup * 4 + wave) * 5 + {0 wait, 1 QK^T, 2 softmax, 3 PV,
 * 4 total}]; NULL switches the instrumented build off again. */
/* (test hook) out_f16[i] = fp16(in[i]) as the kernels convert: finite values beyond the fp16 range saturate to +-65504 (MODE.FP16_OVFL) */
int wvn_debug_f16_saturate(const float* in, void* out_f16, int n, void* stream);
int wvn_debug_attention_timing(long long* dbg);
/* which form of the pre-scaled (scale == 0) attention kernel subsequent launches use: 0 = exact per-tile row max, 1 = lazy (no
 * per-tile max; the row sums raise the alarm and the tile is redone exactly -- the default), < 0 = back to the default.  Same
 * results within the kernel's tolerance; tests/test_gpu_attention_lazy.py runs both, bench.py --attn-variant A/Bs them. */
int wvn_debug_attention_variant(int variant);
/* which assignment kernel subsequent wvn_kmeans_cosine_pixels calls use: -1 / 5 = the PACKED VALU form where it is eligible (K <= 20;
 * default): the fmaf chains of two clusters ride in the two halves of v_pk_fma_f32, the interpolation runs on channel pairs; 4 = packed
 * dot products, plain interpolation; 0 = the plain VALU form (one v_fma_f32 per cluster and channel); 1 = the SCREENED form where it is
 * eligible (K <= 20): similarities from split-operand bf16 MFMAs decide every pixel whose best beats the runner-up by more than the
 * proven error bound, the exact fmaf chains of the definition decide the rest -- faster where the code has cluster structure, slower
 * on structure-less code (csrc/stego.hip); 2 = the screened kernel with every row sent down its exact path (tests); 3 = the screened
 * kernel counting its exact rows (wvn_debug_kmeans_screen_stats).  Bit-identical labels and centroids in all of them;
 * tests/test_gpu_stego_pixels.py runs them, scripts/bench_pixel_kmeans.py A/Bs them. */
/* the fragment form of the split-operand row-panel kernel (fc2 / projection of WVN_PREC_MIX / WVN_PREC_X3): 1 = a wave PAIR per 32 rows, two waves per
 * SIMD (round 5, default: 12 % fewer cycles per k-step, fc2 -4 % wall time), 0 = one wave per SIMD (round 4).  Bit-identical C; A/B and tests */
int wvn_debug_n384_pair(int on);
int wvn_debug_kmeans_assign_form(int form);
/* image rows of a band the linear form's assign kernel works on at a time (LDS per workgroup against barriers per band; default 4) */
int wvn_debug_kmeans_linear_rows(int rows);

/* statistics of the screened kernel (synchronises the device): out[0] = 64-pixel row groups it re-did with the exact chains, out[1] = row
 * groups it saw, since the last call with reset != 0 */
int wvn_debug_kmeans_screen_stats(unsigned long long* out, int reset);
/* subsequent wvn_qkv_fused launches write dbg[(workgroup * 4 + wave) * 4 + {0 LayerNorm prologue, 1 MFMA slices, 2 tile epilogues,
 * 3 total}] in shader cycles (scripts/bench_qkv_fused.py); NULL switches the instrumented build off again. */
int wvn_debug_qkv_fused_timing(long long* dbg);
/* the same for wvn_mlp_fused with the LayerNorm inside: dbg[(workgroup * 4 + wave) * 6 + {0 row prologue (LayerNorm), 1 fc1 slices,
 * 2 GELU + pack, 3 fc2 slices, 4 epilogue, 5 total}] (scripts/bench_mlp_fused.py) */
int wvn_debug_mlp_fused_timing(long long* dbg);
/* the row-panel N = 384 residual GEMM on every row block, whatever M (the dispatcher of wvn_gemm_bf16 only uses it from about
 * 0.75 x #CU row blocks of 256 on); tests */
int wvn_debug_gemm_n384(const void* A, int lda, const void* W, int ldw, const float* bias, float* C, int ldc, int M, int K,
                        void* stream);


/* ---- step scheduling (no reference counterpart: the reference runs one frame at a time on one CUDA stream,
 * wild_visual_navigation_ros/scripts/wvn_feature_extractor_node.py:319-363) ----
 * A HIP stream whose kernels may only occupy the compute units named by `mask` (bit i of the `words` 32-bit words = CU i in the
 * driver's numbering, which interleaves the 8 XCDs of an MI355X: bit i lies on XCD i mod 8), so that the persistent backbone
 * kernels and the many small kernels of clustering / pooling / the learner PARTITION the chip instead of sharing every CU
 * (bench.py's two-stream schedule; scripts/ab_cu_mask.py measures the split).  The handle is a hipStream_t: pass it wherever this
 * header takes `void* stream` (PyTorch: torch.cuda.ExternalStream(handle)).  wvn_stream_destroy releases it. */
int wvn_stream_create_cu_mask(void** stream, const unsigned int* mask, int words);
int wvn_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WVN_HIP_H */
