"""ctypes binding of libwvn_hip.so (include/wvn_hip.h).

PyTorch supplies device memory and streams; every compute call on the hot path goes through this
module into hand-written HIP kernels.  There is NO fallback: if the shared library is missing or a
call returns non-zero the caller gets an exception.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (must be imported first: its bundled libamdhip64.so.7 is the one HIP runtime of the process)

_HERE = os.path.dirname(os.path.abspath(__file__))
# WVN_LIB_PATH: load another build of the same library (same-box A/B runs of compiler flags, scripts/ab_lib.sh); never a fallback
LIB_PATH = os.environ.get("WVN_LIB_PATH") or os.path.join(_HERE, "lib", "libwvn_hip.so")
if os.environ.get("WVN_LIB_PATH"):
    import sys as _sys

    print(f"[wild_visual_navigation_amd] WVN_LIB_PATH is set: loading {LIB_PATH} instead of the in-tree build", file=_sys.stderr)

WVN_MAX_DEPTH = 32
PREC_F32, PREC_BF16, PREC_X3, PREC_FP8, PREC_F16, PREC_MIX = 0, 1, 2, 3, 4, 5
VIT_MLP_FUSED = 1
VIT_QKV_FUSED = 2
VIT_FUSE_ANY_SIZE = 4
VIT_NO_PROJ_IN_MLP = 8
VIT_NO_LN_HANDOVER = 16
VIT_NO_A384_X3 = 32
VIT_NO_MX = 1024


def vit_qsplit_blocks(n: int) -> int:
    """WVN_VIT_QSPLIT_BLOCKS(n) of include/wvn_hip.h: the flag bits that make the first n blocks of WVN_PREC_MIX take q as two planes."""
    return ((int(n) + 1) & 63) << 16

PROF_CATS = ("patchify", "patch_gemm", "layernorm", "qkv_gemm", "attention", "proj_gemm", "fc1_gemm", "fc2_gemm")

# epilogue codes (wvn_internal.h)
EPI_BF16, EPI_GELU_BF16, EPI_RELU_BF16, EPI_F32, EPI_RESID_F32, EPI_ACCUM_F32 = range(6)
EPI_QKV = 7   # (q | k | v^T in the attention kernels' layouts: wvn_gemm_a768_fp8)
EPI_GELU_MX8 = 9   # (gelu -> e4m3 with MX block scales: wvn_gemm_a768_fp8 -> wvn_gemm_fp8_mx)
F32_NONE, F32_RELU, F32_GELU, F32_RESID, F32_SIGMOID0, F32_RELUMASK = range(6)


class WvnError(RuntimeError):
    pass


class VitLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_b", "proj_b", "fc1_b", "fc2_b", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ls1", "ls2", "qkv_s", "proj_s", "fc1_s", "fc2_s", "fc2_w_fused", "fc1_w_fused", "qkv_w_fused", "proj_w_frag",
        "qkv_w_mx", "proj_w_mx", "fc1_w_mx", "fc2_w_mx")]


class VitModel(C.Structure):
    _fields_ = [
        ("img_size", C.c_int), ("patch", C.c_int), ("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int),
        ("mlp_dim", C.c_int), ("precision", C.c_int), ("flags", C.c_int),
        ("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("cls_pos", C.c_void_p), ("pos", C.c_void_p),
        ("norm_g", C.c_void_p), ("norm_b", C.c_void_p),
        ("layers", VitLayer * WVN_MAX_DEPTH),
    ]


class MlpDesc(C.Structure):
    _fields_ = [("D", C.c_int), ("H1", C.c_int), ("H2", C.c_int), ("reserved", C.c_int)]


_lib = None
_lock = threading.Lock()

_i, _f, _p, _ll, _sz = C.c_int, C.c_float, C.c_void_p, C.c_longlong, C.c_size_t
_SIGNATURES = {
    "wvn_version": ([], _i),
    "wvn_vit_workspace_bytes": ([_p, _i], _sz),
    "wvn_vit_forward": ([_p, _p, _i, _p, _p, _i, _p, _sz, _p], _i),
    "wvn_vit_forward_u8": ([_p, _p, _i, _p, _p, _i, _p, _sz, _p], _i),
    "wvn_vit_forward_frames": ([_p, _p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p, _sz, _p], _i),
    "wvn_vit_forward_frames_pair": ([_p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _sz, _p], _i),
    "wvn_resize_nearest_crop": ([_p, _p, _ll, _i, _i, _p, _p, _i, _i, _i, _p], _i),
    "wvn_prof_enable": ([_i], _i),
    "wvn_prof_collect": ([_p, _p], _i),
    "wvn_gemm_bf16": ([_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "wvn_debug_attention_variant": ([_i], _i),
    "wvn_debug_kmeans_assign_form": ([_i], _i),
    "wvn_debug_kmeans_screen_stats": ([_p, _i], _i),
    "wvn_debug_mlp_x3_frag": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p], _i),
    "wvn_debug_gemm_n384_x3": ([_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p], _i),
    "wvn_debug_gemm_n384_mx": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p], _i),
    "wvn_debug_mlp_mx": ([_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p], _i),
    "wvn_debug_qkv_mx": ([_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _p], _i),
    "wvn_debug_gemm_a384_x3": ([_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p], _i),
    "wvn_stream_create_cu_mask": ([_p, _p, _i], _i),
    "wvn_stream_destroy": ([_p], _i),
    "wvn_debug_qkv_fused_timing": ([_p], _i),
    "wvn_debug_mlp_fused_timing": ([_p], _i),
    "wvn_qkv_fused": ([_p, _i, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p], _i),
    "wvn_proj_mlp_fused": ([_p, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_mlp_fused": ([_p, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_gemm_f16": ([_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "wvn_qkv_fused_f16": ([_p, _i, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p], _i),
    "wvn_proj_mlp_resident": ([_p, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _f, _p, _p], _i),
    "wvn_proj_mlp_resident_f16": ([_p, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _f, _p, _p], _i),
    "wvn_qkv_prenorm": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p], _i),
    "wvn_qkv_prenorm_f16": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p], _i),
    "wvn_proj_mlp_fused_f16": ([_p, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_mlp_fused_f16": ([_p, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_attention_f16": ([_p, _p, _p, _p, _i, _i, _i, _i, _f, _p], _i),
    "wvn_gemm_x3": ([_p, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "wvn_quantize_rows_fp8": ([_p, _i, _i, _p, _i, _p, _i, _i, _p], _i),
    "wvn_layernorm_fp8": ([_p, _p, _p, _p, _i, _p, _i, _i, _f, _p], _i),
    "wvn_gemm_fp8": ([_p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "wvn_gemm_a768_fp8": ([_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _f, _p], _i),
    "wvn_gemm_fp8_mx": ([_p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "wvn_split_planes": ([_p, _i, _p, _p, _i, _i, _i, _p], _i),
    "wvn_attention_x3": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p], _i),
    "wvn_gemm_f32": ([_p, _i, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p], _i),
    "wvn_layernorm": ([_p, _p, _p, _p, _i, _i, _i, _f, _p], _i),
    "wvn_attention_bf16": ([_p, _p, _p, _p, _i, _i, _i, _i, _f, _p], _i),
    "wvn_attention_f32": ([_p, _p, _p, _p, _i, _i, _i, _i, _f, _p], _i),
    "wvn_patchify": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "wvn_patchify_u8": ([_p, _p, _i, _i, _i, _p], _i),
    "wvn_cast_f32_to_bf16": ([_p, _p, _ll, _p], _i),
    "wvn_upsample_bilinear": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "wvn_upsample_nearest_i32": ([_p, _p, _i, _i, _i, _p], _i),
    "wvn_segpool_bilinear_mean": ([_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "wvn_segpool_patch_labels": ([_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "wvn_segmean_scratch_bytes": ([_i, _i, _i, _i], _sz),
    "wvn_segmean_tokens": ([_p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _p], _i),
    "wvn_label_pool": ([_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_label_pool_batched": ([_p, _i, _i, _i, _i, _i, _p, _p, _p], _i),
    "wvn_project_render_fmin": ([_p, _i, _p, _i, _i, _i, _i, _i, _p, _f, _p], _i),
    "wvn_wire_bytes": ([_i, _i, _i, _i], _sz),
    "wvn_wire_pack": ([_p, _i, _p, _i, _p, _i, _i, _i, _i, _p], _i),
    "wvn_wire_unpack": ([_p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "wvn_slic_num_clusters": ([_i, _i, _i], _i),
    "wvn_slic_scratch_bytes": ([_i, _i, _i], _sz),
    "wvn_slic": ([_p, _i, _i, _i, _i, _f, _i, _p, _p, _p, _p, _sz, _p], _i),
    "wvn_seg_centers": ([_p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_seg_adjacency": ([_p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "wvn_normalize_rows": ([_p, _i, _p, _i, _i, _p], _i),
    "wvn_argmax_rows": ([_p, _i, _i, _i, _p, _p], _i),
    "wvn_kmeans_scratch_bytes": ([_i, _i, _i, _i], _sz),
    "wvn_kmeans_pixels_scratch_bytes": ([_i, _i, _i, _i, _i], _sz),
    "wvn_kmeans_cosine_pixels": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "wvn_flip_average": ([_p, _p, _p, _i, _i, _i, _p], _i),
    "wvn_kmeans_pixels_linear_supported_shape": ([_i, _i, _i, _i], _i),
    "wvn_kmeans_pixels_linear_scratch_bytes": ([_i, _i, _i, _i, _i], _sz),
    "wvn_kmeans_cosine_pixels_linear": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "wvn_debug_kmeans_linear_rows": ([_i], _i),
    "wvn_debug_n384_pair": ([_i], _i),
    "wvn_table_argmax_slots": ([_i], _i),
    "wvn_table_bilerp_argmax": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "wvn_table_bilerp_argmax_ac": ([_p, _p, _i, _i, _i, _i, _i, _p], _i),
    "wvn_kmeans_cosine_pixels_linear_ac": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "wvn_cast_rows": ([_p, _i, _p, _i, _i, _i, _i, _p], _i),
    "wvn_kmeans_cosine": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "wvn_mlp_param_count": ([_p], _sz),
    "wvn_mlp_workspace_bytes": ([_p, _i], _sz),
    "wvn_mlp_forward": ([_p, _p, _p, _i, _i, _p, _p, _p, _p, _sz, _p], _i),
    "wvn_mlp_train_phase_a": ([_p, _p, _p, _i, _p, _i, _p, _p, _sz, _p], _i),
    "wvn_mlp_train_phase_b": ([_p, _p, _p, _i, _p, _p, _i, _p, _f, _f, _f, _p, _p, _p, _sz, _p], _i),
    "wvn_compact_segment_rows": ([_p, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p], _i),
    "wvn_mlp_train_phase_a_rows": ([_p, _p, _p, _i, _p, _i, _p, _p, _p, _sz, _p, _p], _i),
    "wvn_mlp_train_phase_b_rows": ([_p, _p, _p, _i, _p, _p, _i, _p, _p, _f, _f, _f, _p, _p, _p, _sz, _i, _p], _i),
    "wvn_mlp_train_phase_c": ([_p, _p, _p, _p, _p, _i, _f, _p, _f, _f, _p, _p], _i),
    "wvn_mlp_confidence": ([_p, _i, _p, _i, _f, _f, _f, _p, _p, _i, _i, _p], _i),
    "wvn_pixel_mlp_pack_bytes": ([_p], _sz),
    "wvn_pixel_mlp_zx_cols": ([_p], _i),
    "wvn_pixel_mlp_pack": ([_p, _p, _p, _p], _i),
    "wvn_pixel_mlp_exact_pack_bytes": ([_p], _sz),
    "wvn_pixel_mlp_exact_workspace_bytes": ([_p, _i, _i], _sz),
    "wvn_pixel_mlp_exact_pack": ([_p, _p, _p, _p], _i),
    "wvn_pixel_mlp_infer_exact": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p, _p, _p, _sz, _p], _i),
    "wvn_pixel_mlp_infer": ([_p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _p, _p, _p, _p, _p], _i),
    "wvn_debug_gemm_bf16_timed": ([_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p], _i),
    "wvn_debug_f16_saturate": ([_p, _p, _i, _p], _i),
    "wvn_debug_attention_timing": ([_p], _i),
    "wvn_debug_gemm_n384": ([_p, _i, _p, _i, _p, _p, _i, _i, _i, _p], _i),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib() -> C.CDLL:
    """Load (once) and return the shared library.  Raises WvnError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise WvnError(
                        f"{LIB_PATH} not found: build it with `python -m wild_visual_navigation_amd.csrc.build` "
                        "(there is no CPU / eager fallback for the HIP path)"
                    )
                h = C.CDLL(LIB_PATH)
                for name, (args, res) in _SIGNATURES.items():
                    fn = getattr(h, name)  # AttributeError if the header and the library disagree
                    fn.argtypes = args
                    fn.restype = res
                _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        kind = {1001: "bad argument", 1002: "workspace too small"}.get(rc, f"hipError_t {rc}")
        raise WvnError(f"libwvn_hip: {what} failed: {kind}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous where a dense layout is assumed."""
    return 0 if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t: "torch.Tensor", name: str = "tensor") -> None:
    if not t.is_cuda:
        raise WvnError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
