"""The C-ABI library loads and exports every symbol include/wvn_hip.h declares (no compute calls:
this runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from wild_visual_navigation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "wvn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wvn_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run python -c 'import __graft_entry__ as g; g.build()'"


def test_every_declared_symbol_is_exported_and_bound():
    h = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(h, name), f"{name} declared in wvn_hip.h but not exported"
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS), set(declared) ^ set(_lib.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.VitLayer) == 26 * 8      # (22 pointers + the four MX weights of round 6)
    assert C.sizeof(_lib.VitModel) == 8 * 4 + 6 * 8 + _lib.WVN_MAX_DEPTH * 26 * 8
    assert C.sizeof(_lib.MlpDesc) == 16


def test_host_only_queries():
    h = _lib.lib()
    assert h.wvn_version() >= 100
    assert h.wvn_mlp_param_count(C.byref(_lib.MlpDesc(384, 256, 32, 0))) == 119489
    assert h.wvn_mlp_param_count(C.byref(_lib.MlpDesc(90, 256, 32, 0))) == 34523
    m = _lib.VitModel()
    m.img_size, m.patch, m.dim, m.depth, m.heads, m.mlp_dim, m.precision = 448, 8, 384, 12, 6, 1536, _lib.PREC_BF16
    one = h.wvn_vit_workspace_bytes(C.byref(m), 1)
    two = h.wvn_vit_workspace_bytes(C.byref(m), 2)
    # x fp32 + xn bf16 + q/k/v^T bf16 (3200-padded) + hidden bf16 + patches bf16, per frame
    # (3137 tokens are stored with a per-frame row stride of 3152 = next multiple of 16)
    expect = 3152 * 384 * 4 + 3152 * 384 * 2 + 3 * 6 * 3200 * 64 * 2 + 3152 * 1536 * 2 + 3136 * 192 * 2
    assert expect <= one <= expect + 8 * 256 and two > one
    # exact mode on the matrix pipe: two bf16 planes per operand = the fp32 mode's footprint
    m.precision = _lib.PREC_X3
    x3 = h.wvn_vit_workspace_bytes(C.byref(m), 1)
    m.precision = _lib.PREC_F32
    # (+ the hidden activation and the normalised / attention rows rounded up to whole 32-row groups: the fragment-major hand-overs of the
    #  split-operand block; + {mean, rstd} per row: the LayerNorm statistics its kernels hand across their boundaries)
    assert x3 == h.wvn_vit_workspace_bytes(C.byref(m), 1) + (3168 - 3152) * (1536 + 384) * 4 + (3152 * 8 + 255) // 256 * 256
    # DINOv2 ViT-B/14 at 518^2 (BASELINE configs[4]): 1370 tokens, patch rows 588 -> 640 for the MFMA precisions
    m.img_size, m.patch, m.dim, m.heads, m.mlp_dim, m.precision = 518, 14, 768, 12, 3072, _lib.PREC_BF16
    v2 = h.wvn_vit_workspace_bytes(C.byref(m), 1)
    expect = 1376 * 768 * 4 + 1376 * 768 * 2 + 3 * 12 * 1408 * 64 * 2 + 1376 * 3072 * 2 + 1369 * 640 * 2
    assert expect <= v2 <= expect + 8 * 256


def test_per_pixel_host_queries():
    """Sizes of the fused per-pixel kernels' packed weights / workspace (host-only arithmetic)."""
    h = _lib.lib()
    d = _lib.MlpDesc(384, 256, 32, 0)
    w23 = 16 * 2 * 32 * 16 + 13 * 2 * 2 * 32 * 16                    # W2 and W3 fragment images (bf16)
    nbias = (256 + 32 + 13 * 32) * 4
    assert h.wvn_pixel_mlp_pack_bytes(C.byref(d)) == 256 * 384 * 2 + w23 + nbias
    assert h.wvn_pixel_mlp_exact_pack_bytes(C.byref(d)) == 2 * w23 + nbias
    rows = 2 * 56 * 56
    ws = h.wvn_pixel_mlp_exact_workspace_bytes(C.byref(d), 2, 56)
    assert rows * 256 * 4 + 2 * rows * 640 * 2 <= ws <= rows * 256 * 4 + 2 * rows * 640 * 2 + 256
    assert h.wvn_pixel_mlp_zx_cols(C.byref(d)) == 640
    d90 = _lib.MlpDesc(90, 256, 32, 0)                                # STEGO code: x zero-padded to 128 columns, 3 + 1 W3 tiles
    w23_90 = 16 * 2 * 32 * 16 + 4 * 2 * 2 * 32 * 16
    nbias_90 = (256 + 32 + 4 * 32) * 4
    assert h.wvn_pixel_mlp_zx_cols(C.byref(d90)) == 384
    assert h.wvn_pixel_mlp_pack_bytes(C.byref(d90)) == 256 * 128 * 2 + w23_90 + nbias_90
    assert h.wvn_pixel_mlp_exact_pack_bytes(C.byref(d90)) == 2 * w23_90 + nbias_90
    d64 = _lib.MlpDesc(64, 256, 32, 0)
    assert h.wvn_pixel_mlp_pack_bytes(C.byref(d64)) == 0 and h.wvn_pixel_mlp_exact_pack_bytes(C.byref(d64)) == 0
    assert h.wvn_pixel_mlp_infer(C.byref(d64), None, None, 640, 1, 28, 224, 224, 0.0, 1.0, 0.5, None, None, None, None, None) == 1001
    assert h.wvn_pixel_mlp_infer_exact(C.byref(d), None, None, None, 384, 1, 28, 224, 224, 0.0, 1.0, 0.5, None, None, None, None,
                                       None, 0, None) == 1001


def test_argument_validation_without_gpu():
    h = _lib.lib()
    assert h.wvn_gemm_bf16(None, 0, None, 0, None, None, 0, 1, 1, 64, 0, None) == 1001
    assert h.wvn_gemm_x3(None, None, 0, None, None, 0, None, None, None, 0, 1, 1, 64, 0, None) == 1001
    assert h.wvn_attention_x3(None, None, None, None, None, None, None, None, 1, 6, 100, 128, 0.125, None) == 1001
    assert h.wvn_split_planes(None, 0, None, None, 0, 1, 1, None) == 1001
    assert h.wvn_gemm_fp8(None, 0, None, 0, None, None, None, None, 0, 1, 1, 128, 0, None) == 1001
    assert h.wvn_quantize_rows_fp8(None, 0, 0, None, 0, None, 1, 4, None) == 1001
    assert h.wvn_project_render_fmin(None, 1, None, 0, 4, 3, 8, 8, None, 1.0, None) == 1001
    assert h.wvn_label_pool_batched(None, 1, 3, 8, 8, 4, None, None, None) == 1001
    assert h.wvn_slic(None, 1, 8, 8, 4, 10.0, 10, None, None, None, None, 0, None) == 1001
    assert h.wvn_wire_pack(None, 1, None, 4, None, 8, 8, 2, 4, None) == 1001
    assert h.wvn_slic_num_clusters(448, 448, 100) == 100 and h.wvn_slic_num_clusters(224, 224, 100) == 100
    assert h.wvn_wire_bytes(224, 224, 100, 384) == 64 + 224 * 224 * 4 + 100 * 384 * 4
    m = _lib.VitModel()
    m.img_size, m.patch, m.dim, m.depth, m.heads, m.mlp_dim, m.precision = 518, 14, 768, 12, 12, 3072, _lib.PREC_FP8
    bf = _lib.VitModel()
    bf.img_size, bf.patch, bf.dim, bf.depth, bf.heads, bf.mlp_dim, bf.precision = 518, 14, 768, 12, 12, 3072, _lib.PREC_BF16
    # fp8 workspace = the bf16 one + e4m3 images of the GEMM inputs (M x 768, M x 3072) + per-token scales
    extra = h.wvn_vit_workspace_bytes(C.byref(m), 2) - h.wvn_vit_workspace_bytes(C.byref(bf), 2)
    assert 2 * 1376 * (768 + 3072 + 4) <= extra <= 2 * 1376 * (768 + 3072 + 4) + 3 * 256
    assert h.wvn_vit_forward(None, None, 1, None, None, 0, None, 0, None) == 1001
    with pytest.raises(_lib.WvnError):
        _lib.check(1002, "x")


def test_no_cpu_fallback():
    import torch

    from wild_visual_navigation_amd import ops
    from wild_visual_navigation_amd.backbone import VitBackbone, synthetic_vit_state_dict

    with pytest.raises(_lib.WvnError):
        ops.upsample_bilinear(torch.zeros(1, 4, 8), 2, 4)  # CPU tensor: must refuse, not fall back
    with pytest.raises(_lib.WvnError):
        VitBackbone(synthetic_vit_state_dict(depth=1, pretrain_grid=2), 16, 8, 6, device="cpu")
