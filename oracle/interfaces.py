"""Oracle: DinoInterface / StegoInterface pre- and post-processing.  TEST INFRASTRUCTURE ONLY.

Follows wild_visual_navigation/feature_extractor/dino_interface.py:52-59,70-92 and
stego_interface.py:51-58,73-111.  torchvision is absent here, so ``T.Resize(NEAREST)``,
``T.CenterCrop`` and ``T.Normalize`` are restated from their documented tensor semantics.
"""
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import vit

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize_nearest_center_crop(img: torch.Tensor, size: int) -> torch.Tensor:
    """T.Resize(size, NEAREST) (smaller edge -> size, aspect kept) then T.CenterCrop(size).

    dino_interface.py:54-57.  Identity when the input is already size x size.
    """
    H, W = img.shape[-2:]
    if (H, W) != (size, size):
        if H <= W:
            nh, nw = size, int(size * W / H)
        else:
            nh, nw = int(size * H / W), size
        if (nh, nw) != (H, W):
            img = F.interpolate(img, size=(nh, nw), mode="nearest")
        top = int(round((nh - size) / 2.0))
        left = int(round((nw - size) / 2.0))
        img = img[..., top : top + size, left : left + size]
    return img


def normalize(img: torch.Tensor) -> torch.Tensor:
    """T.Normalize(IMAGENET mean/std) -- dino_interface.py:52."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=img.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=img.dtype).view(1, 3, 1, 1)
    return (img - mean) / std


def dino_transform(img: torch.Tensor, input_size: int) -> torch.Tensor:
    return normalize(resize_nearest_center_crop(img, input_size))


def upsample_bilinear_ac(feat: torch.Tensor, out: int) -> torch.Tensor:
    """F.interpolate(features, (H, H), mode='bilinear', align_corners=True) -- dino_interface.py:87-90.
    Note the reference uses H (the image height) for BOTH output dims."""
    return F.interpolate(feat, (out, out), mode="bilinear", align_corners=True)


def dino_inference(sd, img: torch.Tensor, input_size: int, patch: int, heads: int) -> torch.Tensor:
    """DinoInterface.inference: [B,3,H,W] in [0,1] -> dense [B,D,H,H] fp32."""
    x = dino_transform(img, input_size)
    feat = vit.vit_features(sd, x, patch, heads)
    return upsample_bilinear_ac(feat, img.shape[2])


# ----------------------------------------------------------------------------------------------
# STEGO head + per-image clustering.  The arithmetic lives in the absent third-party ``stego``
# package (stego_interface.py:14,43,91,94-100) -> PARITY UNPINNED; the definition below is the one
# this build documents (DESIGN.md "STEGO definition"): published STEGO segmentation head
# (1x1 conv D->C linear branch + 1x1 conv D->D, ReLU, 1x1 conv D->C non-linear branch, summed),
# averaged with the flipped-back code of the mirrored frame, and, for run_clustering=True, a deterministic
# per-image cosine k-means (n_image_clusters centroids, fixed iteration count, lowest-index tie-break)
# over the code up-sampled to the input size; the single-pass / patch-resolution forms (labels then
# nearest-upsampled by stego_interface.py:108) are the documented cheap options.
# ----------------------------------------------------------------------------------------------
STEGO_CODE_DIM = 90
KMEANS_ITERS = 10
KMEANS_CHUNK = 64
KMEANS_SUPER = None  # (tests) overrides kmeans_super() when set


def kmeans_super(P: int) -> int:
    """Chunk partials are folded in groups of this many consecutive chunks: 8 (512 points) up to 8192 points, 16 above (the
    pixel-resolution clustering of a 448 x 448 frame then folds 196 group partials instead of 392; csrc/stego.hip: km_super)."""
    if KMEANS_SUPER is not None:
        return KMEANS_SUPER
    return 16 if P > 8192 else 8


def make_stego_head_state_dict(D: int = 384, C: int = STEGO_CODE_DIM, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(1000 + seed)

    def rn(*shape, std):
        return torch.randn(*shape, generator=g) * std

    return {
        "cluster1.0.weight": rn(C, D, std=0.08),
        "cluster1.0.bias": rn(C, std=0.02),
        "cluster2.0.weight": rn(D, D, std=0.06),
        "cluster2.0.bias": rn(D, std=0.02),
        "cluster2.2.weight": rn(C, D, std=0.08),
        "cluster2.2.bias": rn(C, std=0.02),
    }


def stego_code_tokens(head: Dict[str, torch.Tensor], tok: torch.Tensor) -> torch.Tensor:
    """Patch tokens [B, G*G, D] -> code [B, G*G, C] (1x1 convs == per-token linears)."""
    lin = F.linear(tok, head["cluster1.0.weight"], head["cluster1.0.bias"])
    hid = F.relu(F.linear(tok, head["cluster2.0.weight"], head["cluster2.0.bias"]))
    return lin + F.linear(hid, head["cluster2.2.weight"], head["cluster2.2.bias"])


def _fma32(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Correctly rounded fp32 fused multiply-add of fp32 arrays.  The product of two fp32 values is exact in fp64; the fp64 sum
    with c can carry a rounding error, which matters only when the fp64 sum lands exactly on an fp32 rounding midpoint -- the
    error term of the addition (TwoSum) then decides the direction."""
    a64, b64, c64 = a.astype(np.float64), b.astype(np.float64), np.broadcast_to(c, np.broadcast_shapes(a.shape, b.shape, np.shape(c))).astype(np.float64)
    p = a64 * b64
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)
    r = s.astype(np.float32)
    r64 = r.astype(np.float64)
    other = np.where(s > r64, np.nextafter(r, np.float32(np.inf)), np.nextafter(r, np.float32(-np.inf))).astype(np.float32)
    tie = (s != r64) & (np.abs(s - r64) == np.abs(other.astype(np.float64) - s))
    # on a tie numpy rounded to even; the true value s + err lies on err's side of the midpoint
    want_other = tie & (((err > 0) & (other.astype(np.float64) > r64)) | ((err < 0) & (other.astype(np.float64) < r64)))
    return np.where(want_other, other, r).astype(np.float32)


def _seq_dot_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """sum_d a[..., d] * b[..., d] as a FUSED multiply-add chain over the index, acc = fma(a_d, b_d, acc) from +0: one fp32
    rounding per step -- the exact order and rounding of the HIP k-means kernels (v_fma_f32), so integer outputs can be compared
    bit for bit.  (Round 2 defined it with separate multiply / add roundings: twice the instructions on the GPU for no benefit.)"""
    acc = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), dtype=np.float32)
    for d in range(a.shape[-1]):
        acc = _fma32(np.broadcast_to(a[..., d], acc.shape), np.broadcast_to(b[..., d], acc.shape), acc)
    return acc


def _normalize_rows_f32(x: np.ndarray) -> np.ndarray:
    """x * (1 / max(||x||, 1e-12)): one correctly rounded reciprocal per row, one fp32 multiply per element (the HIP kernels'
    form -- the pixel-resolution k-means re-creates its rows in every pass, and 90 multiplies are cheaper than 90 divisions)."""
    n2 = _seq_dot_f32(x, x)
    n = np.sqrt(n2).astype(np.float32)
    n = np.maximum(n, np.float32(1e-12))
    rinv = (np.float32(1.0) / n).astype(np.float32)
    return (x * rinv[..., None]).astype(np.float32)


_ORACLE_LIB = None


def _oracle_lib():
    """oracle/_build/libwvn_oracle.so (oracle/kmeans_ref.c, built by __graft_entry__.build() / oracle.build_oracle), or None."""
    global _ORACLE_LIB
    if _ORACLE_LIB is None:
        import ctypes
        import os

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libwvn_oracle.so")
        _ORACLE_LIB = False
        if os.path.exists(path):
            h = ctypes.CDLL(path)
            h.wvn_oracle_kmeans_cosine.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            h.wvn_oracle_kmeans_cosine.restype = ctypes.c_int
            h.wvn_oracle_kmeans_cosine_ex.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p]
            h.wvn_oracle_kmeans_cosine_ex.restype = ctypes.c_int
            if hasattr(h, "wvn_oracle_kmeans_pixels_linear_ac"):   # (round 6: the tap rule as a parameter)
                h.wvn_oracle_kmeans_pixels_linear_ac.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
                h.wvn_oracle_kmeans_pixels_linear_ac.restype = ctypes.c_int
            if hasattr(h, "wvn_oracle_kmeans_pixels_linear"):   # (oracle/kmeans_linear_ref.c; absent from a library built before round 5)
                h.wvn_oracle_kmeans_pixels_linear.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
                h.wvn_oracle_kmeans_pixels_linear.restype = ctypes.c_int
            _ORACLE_LIB = h
    return _ORACLE_LIB or None


def kmeans_cosine_labels(code: np.ndarray, K: int, iters: int = KMEANS_ITERS, force_numpy: bool = False) -> np.ndarray:
    """The k-means below through its C restatement (oracle/kmeans_ref.c: the same operations with libm's fmaf, seconds instead of
    minutes at 448 x 448 pixels) when that library has been built, else -- and with ``force_numpy`` -- the numpy statement
    itself; tests/test_oracle_stego.py holds the two against each other."""
    h = None if force_numpy else _oracle_lib()
    if h is None:
        return kmeans_cosine_labels_numpy(code, K, iters)
    code = np.ascontiguousarray(code, dtype=np.float32)
    labels = np.empty(code.shape[0], dtype=np.int32)
    if h.wvn_oracle_kmeans_cosine(code.ctypes.data, code.shape[0], code.shape[1], K, iters, labels.ctypes.data) != 0:
        raise MemoryError("oracle k-means")
    return labels


def kmeans_cosine_labels_numpy(code: np.ndarray, K: int, iters: int = KMEANS_ITERS) -> np.ndarray:
    """Deterministic cosine k-means over one image's code vectors.

    code: [P, C] fp32.  Returns int32 labels [P] in 0..K-1 (not yet compacted).
      x_p   = code_p / max(||code_p||, 1e-12)
      c_k^0 = x at index floor((2k+1) P / (2K))
      (normalisation everywhere: x * (1 / max(||x||, 1e-12)), _normalize_rows_f32)
      repeat ``iters`` times: label_p = argmax_k <x_p, c_k> (lowest k wins ties);
                              c_k = normalise(sum of its x_p in ascending p); empty cluster keeps c_k.
      final labels = one more assignment against the last centroids.
    All arithmetic fp32; norms and similarities are fma chains over the channel index (see _seq_dot_f32), centroid sums plain
    additions in the order given below.
    """
    code = np.ascontiguousarray(code, dtype=np.float32)
    P, C = code.shape
    x = _normalize_rows_f32(code)
    init = [((2 * k + 1) * P) // (2 * K) for k in range(K)]
    cent = x[init].copy()

    def assign(cent):
        sim = _seq_dot_f32(x[:, None, :], cent[None, :, :])  # [P, K]
        return np.argmax(sim, axis=1).astype(np.int32)  # first maximum == lowest index

    for _ in range(iters):
        lab = assign(cent)
        # centroid sums in the kernel's fixed order: chunks of KMEANS_CHUNK consecutive points, members of a
        # cluster added in ascending point order inside a chunk (from 0); chunk partials added in ascending
        # chunk order inside groups of kmeans_super(P) consecutive chunks (from 0); group partials added in
        # ascending group order (from 0) -- one fp32 rounding per addition
        sums = np.zeros((K, C), dtype=np.float32)
        sup = kmeans_super(P)
        for g0 in range(0, P, KMEANS_CHUNK * sup):
            grp = np.zeros((K, C), dtype=np.float32)
            for p0 in range(g0, min(P, g0 + KMEANS_CHUNK * sup), KMEANS_CHUNK):
                part = np.zeros((K, C), dtype=np.float32)
                np.add.at(part, lab[p0:p0 + KMEANS_CHUNK], x[p0:p0 + KMEANS_CHUNK])  # unbuffered: ascending order
                grp = (grp + part).astype(np.float32)
            sums = (sums + grp).astype(np.float32)
        cnt = np.bincount(lab, minlength=K)
        new = _normalize_rows_f32(sums)
        cent = np.where((cnt > 0)[:, None], new, cent).astype(np.float32)
    return assign(cent)


def bilinear_taps_fixed(G: int, H: int, align_corners: bool = True):
    """(i0, i1, w0, w1) per output index o < H of the fixed-order bilinear interpolation (csrc/common.h: lerp_tap / lerp_tap_ac), every operation
    rounded to fp32 on its own.  align_corners=True (WVN's own up-sample, dino_interface.py:87-90 / stego_interface.py:107): ATen's coordinates
    src = o (G - 1) / (H - 1).  align_corners=False (the other reading of the absent STEGO package's code interpolation; ATen's
    area_pixel_compute_source_index): src = max((G / H) (o + 0.5) - 0.5, 0).  Both: i0 = floor(src), i1 = min(i0 + 1, G - 1), w1 = src - i0, w0 = 1 - w1."""
    f32 = np.float32
    o = np.arange(H, dtype=np.float32)
    if align_corners:
        scale = (f32(G - 1) / f32(H - 1)) if H > 1 else f32(0)
        sc = (scale * o).astype(np.float32)
    else:
        scale = f32(G) / f32(H)
        m = (scale * (o + f32(0.5)).astype(np.float32)).astype(np.float32)
        sc = np.maximum((m - f32(0.5)).astype(np.float32), f32(0))
    i0 = sc.astype(np.int32)
    i1 = i0 + (i0 < G - 1)
    w1 = (sc - i0.astype(np.float32)).astype(np.float32)
    w0 = (f32(1) - w1).astype(np.float32)
    return i0, i1, w0, w1


def upsample_bilinear_fixed(code: np.ndarray, H: int, align_corners: bool = True) -> np.ndarray:
    """[G, G, C] fp32 patch map -> [H, H, C] fp32: F.interpolate(.., (H, H), mode="bilinear", align_corners=...) in the ONE
    fixed operation order of the HIP kernels (csrc/common.h lerp_tap / bilerp_fixed: ATen's coordinates, then
    t0 = fma(wx1, v01, wx0 * v00), t1 = fma(wx1, v11, wx0 * v10), out = fma(wy1, t1, wy0 * t0)), bit for bit -- ATen's own CPU
    kernel rounds in another order, so integer results derived from the up-sampled code are pinned through this form."""
    code = np.ascontiguousarray(code, dtype=np.float32)
    G = code.shape[0]
    i0, i1, w0, w1 = bilinear_taps_fixed(G, H, align_corners)
    wx0, wx1 = w0[None, :, None], w1[None, :, None]
    out = np.empty((H, H, code.shape[2]), dtype=np.float32)
    for y in range(H):   # row by row: the whole [H, H, C] set of temporaries would be several GB at 448^2 x 90
        r0, r1 = code[i0[y]][None], code[i1[y]][None]                   # [1, G, C]
        t0 = _fma32(wx1, r0[:, i1], (wx0 * r0[:, i0]).astype(np.float32))
        t1 = _fma32(wx1, r1[:, i1], (wx0 * r1[:, i0]).astype(np.float32))
        out[y] = _fma32(np.full_like(t1, w1[y]), t1, (w0[y] * t0).astype(np.float32))[0]
    return out


def stego_code_flip_average(head: Dict[str, torch.Tensor], tok: torch.Tensor, tok_mirror: torch.Tensor, G: int) -> torch.Tensor:
    """The flip pass of the upstream Stego.get_code as this build reads it: the code of the frame averaged with the
    flipped-back code of its mirror image.  tok / tok_mirror: patch tokens [B, G*G, D] of the frame and of img.flip(-1)."""
    code = stego_code_tokens(head, tok)
    B, P, C = code.shape
    c2 = stego_code_tokens(head, tok_mirror).reshape(B, G, G, C).flip(2).reshape(B, P, C)
    return (code + c2) * 0.5


def kmeans_cosine_labels_pixels(code_tokens: np.ndarray, G: int, H: int, K: int, iters: int = KMEANS_ITERS, align_corners: bool = True) -> np.ndarray:
    """cluster_resolution="pixel": the k-means above over the H x H up-sampled (fixed-order bilinear, align_corners as given) code
    pixels of one frame.  code_tokens [G*G, C] -> int32 labels [H*H] (not compacted)."""
    dense = upsample_bilinear_fixed(code_tokens.reshape(G, G, -1), H, align_corners)
    return kmeans_cosine_labels(dense.reshape(H * H, -1), K, iters)


def relabel_ascending(seg: np.ndarray) -> np.ndarray:
    """feature_extractor.py:245-246: replace the sorted unique ids by 0..K'-1."""
    uniq = np.unique(seg)
    lut = np.zeros(int(uniq.max()) + 1, dtype=np.int64)
    lut[uniq] = np.arange(len(uniq))
    return lut[seg]


def upsample_nearest(lab: torch.Tensor, out: int) -> torch.Tensor:
    """F.interpolate(pred[None].float(), (H,H), mode='nearest').int() -- stego_interface.py:108-109."""
    return F.interpolate(lab[None].float(), (out, out), mode="nearest").int()


def stego_inference(
    sd, head, img: torch.Tensor, input_size: int, patch: int, heads: int, n_image_clusters: int,
    flip_tta: bool = True, cluster_resolution: str = "pixel", kmeans_form: str = "linear",
) -> Tuple[torch.Tensor, torch.Tensor]:
    """StegoInterface.inference as used by FeatureExtractor (run_clustering=True, run_crf=False).
    Returns (code [B,C,H,H] fp32, cluster_pred [1,B,H,H] int32).  Defaults = the upstream behaviour as this build reads it (the code
    averaged with the flipped-back code of the mirrored frame, k-means over the H x H up-sampled code pixels); flip_tta=False /
    cluster_resolution="patch" are the cheap forms (single pass, k-means over the patch codes, labels nearest-upsampled).
    kmeans_form: which statement of the pixel-resolution k-means assigns the labels -- "linear" (oracle/kmeans_linear.py, the
    product default) or "direct" (kmeans_cosine_labels_pixels above); the same clustering, equal maps except at fp32 rounding ties."""
    x = dino_transform(img, input_size)
    tok = vit.vit_tokens(sd, x, patch, heads)[:, 1:]
    B, P, D = tok.shape
    G = input_size // patch
    if flip_tta:
        tok_m = vit.vit_tokens(sd, x.flip(-1), patch, heads)[:, 1:]
        code = stego_code_flip_average(head, tok, tok_m, G)
    else:
        code = stego_code_tokens(head, tok)  # [B, P, C]
    code_map = code.reshape(B, G, G, -1).permute(0, 3, 1, 2)
    H = img.shape[2]
    if cluster_resolution == "pixel":
        if kmeans_form == "linear":
            from . import kmeans_linear

            km = kmeans_linear.kmeans_cosine_labels_pixels_linear
        else:
            km = kmeans_cosine_labels_pixels
        labels = np.stack([km(code[b].numpy(), G, input_size, n_image_clusters) for b in range(B)])
        labels = torch.from_numpy(labels).reshape(B, input_size, input_size)
    else:
        labels = np.stack([kmeans_cosine_labels(code[b].numpy(), n_image_clusters) for b in range(B)])
        labels = torch.from_numpy(labels).reshape(B, G, G)
    return upsample_bilinear_ac(code_map, H), upsample_nearest(labels, H)
