"""Functional wrappers (torch tensors in / out) around the C-ABI kernels of libwvn_hip.so.
Allocation and stream selection happen here; arithmetic happens in HIP.  No fallbacks."""
import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def upsample_bilinear(tokens: torch.Tensor, grid: int, out_size: int) -> torch.Tensor:
    """tokens [B,G*G,D] fp32 -> dense [B,D,H,H] (align_corners=True; dino_interface.py:87-90)."""
    require_cuda(tokens, "tokens")
    tokens = tokens.contiguous()
    B, P, D = tokens.shape
    out = torch.empty(B, D, out_size, out_size, dtype=torch.float32, device=tokens.device)
    check(lib().wvn_upsample_bilinear(ptr(tokens), ptr(out), B, grid, D, out_size, stream()), "wvn_upsample_bilinear")
    return out


def upsample_nearest_labels(labels: torch.Tensor, out_size: int) -> torch.Tensor:
    """labels [B,G,G] int32 -> [B,H,H] int32 (stego_interface.py:108-109)."""
    require_cuda(labels, "labels")
    labels = _i32(labels).contiguous()
    B, G, _ = labels.shape
    out = torch.empty(B, out_size, out_size, dtype=torch.int32, device=labels.device)
    check(lib().wvn_upsample_nearest_i32(ptr(labels), ptr(out), B, G, out_size, stream()), "wvn_upsample_nearest_i32")
    return out


def segpool_bilinear_mean(seg: torch.Tensor, tokens: torch.Tensor, grid: int, n_seg: int,
                          return_counts: bool = False):
    """Fused up-sample + per-segment mean (feature_extractor.py:390-396 on dino_interface.py:87-90).
    seg [B,H,W] int (-1 ignored), tokens [B,G*G,D] fp32 -> feat [B,S,D] fp32 (NaN row for an empty id)."""
    require_cuda(tokens, "tokens")
    seg = _i32(seg).contiguous()
    B, H, W = seg.shape
    tokens = tokens.contiguous()
    D = tokens.shape[-1]
    dev = tokens.device
    feat = torch.empty(B, n_seg, D, dtype=torch.float32, device=dev)
    wbuf = torch.empty(B * n_seg * grid * grid, dtype=torch.int64, device=dev)  # 2^-40 fixed-point tap weights
    cnt = torch.empty(B * n_seg, dtype=torch.int32, device=dev)
    check(lib().wvn_segpool_bilinear_mean(ptr(seg), ptr(tokens), D, ptr(feat), ptr(wbuf), ptr(cnt), B, H, W, grid,
                                          n_seg, D, stream()), "wvn_segpool_bilinear_mean")
    return (feat, cnt.reshape(B, n_seg)) if return_counts else feat


_STENCILS = {}


def patch_stencil_tables(grid: int, out_size: int, device):
    """[G][3] stencil weights (offsets -1, 0, +1) of the block-averaged align_corners=True bilinear
    up-sampling grid -> out_size for patch size P = out_size // grid; None when the geometry is not
    patch-aligned (out_size != grid * P) or a tap falls outside the 3-wide stencil."""
    key = (grid, out_size, str(device))
    if key not in _STENCILS:
        import numpy as np

        P = out_size // grid
        tab = None
        if P * grid == out_size and P >= 1:
            scale = np.float32(grid - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0)
            tab = np.zeros((grid, 3), dtype=np.float32)
            for y in range(out_size):
                g = y // P
                src = np.float32(scale * np.float32(y))
                y0 = int(src)
                y1 = y0 + (1 if y0 < grid - 1 else 0)
                w1 = np.float32(src - np.float32(y0))
                for yy, ww in ((y0, np.float32(1) - w1), (y1, w1)):
                    if not -1 <= yy - g <= 1:
                        tab = None
                        break
                    tab[g, yy - g + 1] += ww
                if tab is None:
                    break
            if tab is not None:
                tab = torch.from_numpy(tab / np.float32(P)).to(device)
        _STENCILS[key] = tab
    return _STENCILS[key]


def segpool_patch_labels(labels: torch.Tensor, tokens: torch.Tensor, grid: int, out_size: int, n_seg: int):
    """Patch-aligned fast path of segpool_bilinear_mean: labels [B,G,G] (or [B,G*G]) int32 at patch
    resolution, tokens [B,G*G,D] -> feat [B,S,D].  Returns None if the geometry is not patch-aligned."""
    require_cuda(tokens, "tokens")
    tab = patch_stencil_tables(grid, out_size, tokens.device)
    if tab is None or n_seg * 65 * 4 > 60 * 1024:
        return None
    tokens = tokens.contiguous()
    B, P, D = tokens.shape
    labels = _i32(labels).reshape(B, P).contiguous()
    feat = torch.empty(B, n_seg, D, dtype=torch.float32, device=tokens.device)
    check(lib().wvn_segpool_patch_labels(ptr(labels), ptr(tokens), D, ptr(tab), ptr(tab), ptr(feat), B, grid, n_seg, D,
                                         stream()), "wvn_segpool_patch_labels")
    return feat


def segmean_tokens(seg: torch.Tensor, tokens: torch.Tensor, n_seg: int) -> torch.Tensor:
    """Plain per-segment mean of a pixel-resolution map: seg [B,H,W] / [B,P], tokens [B,P,D] -> [B,S,D]."""
    require_cuda(tokens, "tokens")
    tokens = tokens.contiguous().float()
    B, P, D = tokens.shape
    seg = _i32(seg).reshape(B, P).contiguous()
    out = torch.empty(B, n_seg, D, dtype=torch.float32, device=tokens.device)
    cnt = torch.empty(B * n_seg, dtype=torch.int32, device=tokens.device)
    scratch = torch.empty(lib().wvn_segmean_scratch_bytes(B, P, n_seg, D), dtype=torch.uint8, device=tokens.device)
    check(lib().wvn_segmean_tokens(ptr(seg), ptr(tokens), ptr(out), ptr(cnt), ptr(scratch), scratch.numel(), B, P, n_seg, D,
                                   stream()), "wvn_segmean_tokens")
    return out


def label_pool(mask: torch.Tensor, seg: torch.Tensor, n_seg: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """MissionNode.update_supervision_signal (nodes.py:400-440). mask [C,H,W] fp32 w/ NaN, seg [H,W]."""
    require_cuda(mask, "mask")
    mask = mask.contiguous().float()
    if mask.dim() == 2:
        mask = mask[None]
    seg = _i32(seg).contiguous()
    Cc, H, W = mask.shape
    dev = mask.device
    signal = torch.empty(n_seg, dtype=torch.float32, device=dev)
    valid = torch.empty(n_seg, dtype=torch.uint8, device=dev)
    ssum = torch.empty(n_seg, dtype=torch.int64, device=dev)   # 2^-32 fixed-point sums (deterministic integer atomics)
    scnt = torch.empty(n_seg, dtype=torch.int32, device=dev)
    check(lib().wvn_label_pool(ptr(mask), Cc, ptr(seg), ptr(signal), ptr(valid), ptr(ssum), ptr(scnt), H, W, n_seg,
                               stream()), "wvn_label_pool")
    return signal, valid.bool()


def _record_table(rows, dev) -> torch.Tensor:
    """Device array of C records whose fields are 8-byte words (pointers / int pairs), built from Python ints."""
    return torch.tensor(rows, dtype=torch.int64).to(dev)


def label_pool_batched(masks, segs, n_segs):
    """``label_pool`` for a list of nodes in ONE launch pair.  masks: list of [C,H,W] fp32 CUDA tensors (same shape), segs:
    list of [H,W] int32 CUDA tensors, n_segs: list of ints -> list of (signal [S_i] fp32, valid [S_i] bool)."""
    n = len(masks)
    dev = masks[0].device
    Cc, H, W = masks[0].shape
    smax = max(n_segs)
    sig = [torch.empty(s, dtype=torch.float32, device=dev) for s in n_segs]
    val = [torch.empty(s, dtype=torch.uint8, device=dev) for s in n_segs]
    for m, sg in zip(masks, segs):
        require_cuda(m, "mask")
        if m.dtype != torch.float32 or not m.is_contiguous() or sg.dtype != torch.int32 or not sg.is_contiguous():
            raise _lib.WvnError("label_pool_batched: masks must be contiguous fp32 [C,H,W], segment maps contiguous int32 [H,W]")
    table = _record_table([[ptr(m), ptr(sg), ptr(a), ptr(b), s] for m, sg, a, b, s in zip(masks, segs, sig, val, n_segs)], dev)
    ssum = torch.empty(n * smax, dtype=torch.int64, device=dev)
    scnt = torch.empty(n * smax, dtype=torch.int32, device=dev)
    check(lib().wvn_label_pool_batched(ptr(table), n, Cc, H, W, smax, ptr(ssum), ptr(scnt), stream()), "wvn_label_pool_batched")
    return [(a, b.bool()) for a, b in zip(sig, val)]


def project_render_fmin(Ks, poses, masks, points: torch.Tensor, value, want_projected: bool = False):
    """ImageProjector.project_and_render + torch.fmin merge for n nodes in one launch (csrc/supervision.hip).
    Ks / poses: lists of [4,4] fp32 CUDA tensors (scaled camera matrix, pose_cam_in_world); masks: list of contiguous
    [C,H,W] fp32 CUDA tensors, UPDATED IN PLACE; points [N,3] (shared) or [n,N,3]; value: float or 1-element CUDA tensor
    (colour * traversability).  ``want_projected``: returns (projected [n,N,2] raw pinhole coordinates, depth [n,N] camera-frame z)."""
    n = len(masks)
    dev = masks[0].device
    Cc, H, W = masks[0].shape
    points = points.to(dev, torch.float32).contiguous()
    batched = points.dim() == 3
    N = points.shape[-2]
    proj = torch.empty(n, N, 2, dtype=torch.float32, device=dev) if want_projected else None
    depth = torch.empty(n, N, dtype=torch.float32, device=dev) if want_projected else None
    keep = []
    rows = []
    for i in range(n):
        K = Ks[i].to(dev, torch.float32).contiguous()
        T = poses[i].to(dev, torch.float32).contiguous()
        keep += [K, T]
        m = masks[i]
        if m.dtype != torch.float32 or not m.is_contiguous() or tuple(m.shape) != (Cc, H, W):
            raise _lib.WvnError("project_render_fmin: masks must be contiguous fp32 [C,H,W] of one shape")
        rows.append([ptr(K), ptr(T), ptr(m), ptr(proj[i]) if proj is not None else 0, ptr(depth[i]) if depth is not None else 0])
    table = _record_table(rows, dev)
    vdev = value if isinstance(value, torch.Tensor) else None
    if vdev is not None:
        vdev = vdev.to(dev, torch.float32).reshape(-1).contiguous()
    check(lib().wvn_project_render_fmin(ptr(table), n, ptr(points), int(batched), N, Cc, H, W, ptr(vdev),
                                        0.0 if vdev is not None else float(value), stream()), "wvn_project_render_fmin")
    return (proj, depth) if want_projected else None


_SLIC_TABLES = {}


def slic_tables(device):
    """sRGB -> linear (x 4095) and CIELAB f(t) (x 4096) lookup tables of the integer SLIC (csrc/slic.hip), built in double
    precision on the host."""
    key = str(device)
    if key not in _SLIC_TABLES:
        import numpy as np

        c = np.arange(256, dtype=np.float64) / 255.0
        lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
        t = np.arange(4096, dtype=np.float64) / 4095.0
        f = np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
        _SLIC_TABLES[key] = (torch.from_numpy(np.rint(lin * 4095.0).astype(np.int32)).to(device),
                             torch.from_numpy(np.rint(f * 4096.0).astype(np.int32)).to(device))
    return _SLIC_TABLES[key]


def slic(img: torch.Tensor, num_components: int = 100, compactness: float = 10.0, iters: int = 10) -> torch.Tensor:
    """img [3,H,W] uint8 or float in [0,1] (CUDA) -> SLIC label map [H,W] int32 in [0, slic_num_clusters)."""
    require_cuda(img, "img")
    u8 = img.dtype == torch.uint8
    img = img.contiguous() if u8 else img.contiguous().float()
    _, H, W = img.shape
    lin, f = slic_tables(img.device)
    labels = torch.empty(H, W, dtype=torch.int32, device=img.device)
    scratch = torch.empty(lib().wvn_slic_scratch_bytes(H, W, num_components), dtype=torch.uint8, device=img.device)
    check(lib().wvn_slic(ptr(img), int(u8), H, W, num_components, float(compactness), iters, ptr(lin), ptr(f), ptr(labels),
                         ptr(scratch), scratch.numel(), stream()), "wvn_slic")
    return labels


def slic_num_clusters(H: int, W: int, num_components: int) -> int:
    return int(lib().wvn_slic_num_clusters(H, W, num_components))


def seg_centers(seg: torch.Tensor, n_seg: int) -> torch.Tensor:
    require_cuda(seg, "seg")
    seg = _i32(seg).contiguous()
    H, W = seg.shape[-2:]
    out = torch.empty(n_seg, 2, dtype=torch.float32, device=seg.device)
    scratch = torch.empty(3 * n_seg, dtype=torch.int64, device=seg.device)
    check(lib().wvn_seg_centers(ptr(seg), ptr(out), ptr(scratch), H, W, n_seg, stream()), "wvn_seg_centers")
    return out


def seg_adjacency(seg: torch.Tensor, n_seg: int) -> torch.Tensor:
    require_cuda(seg, "seg")
    seg = _i32(seg).contiguous()
    H, W = seg.shape[-2:]
    dev = seg.device
    max_edges = n_seg * n_seg
    edges = torch.empty(max_edges, 2, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    bitmap = torch.empty(n_seg * n_seg, dtype=torch.uint8, device=dev)
    check(lib().wvn_seg_adjacency(ptr(seg), ptr(edges), ptr(count), ptr(bitmap), H, W, n_seg, max_edges, stream()),
          "wvn_seg_adjacency")
    return edges[: int(count.item())]  # the edge count fixes the output shape: one host sync, as in the reference


def kmeans_cosine(code: torch.Tensor, K: int, iters: int = 10, relabel: bool = True, return_centroids: bool = False):
    """code [B,P,C] fp32 (any row stride) -> (labels [B,P] int32, n_segments [B] int32[, final centroids [B,K,C]])."""
    require_cuda(code, "code")
    B, P, Cc = code.shape
    if code.stride(2) != 1 or code.stride(0) != P * code.stride(1):
        code = code.contiguous()
    dev = code.device
    xn = torch.empty(B, P, Cc, dtype=torch.float32, device=dev)
    check(lib().wvn_normalize_rows(ptr(code), code.stride(1), ptr(xn), B * P, Cc, stream()), "wvn_normalize_rows")
    labels = torch.empty(B, P, dtype=torch.int32, device=dev)
    nseg = torch.empty(B, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib().wvn_kmeans_scratch_bytes(B, P, Cc, K), dtype=torch.uint8, device=dev)
    check(lib().wvn_kmeans_cosine(ptr(xn), ptr(labels), ptr(nseg), ptr(scratch), B, P, Cc, K, iters, int(relabel),
                                  stream()), "wvn_kmeans_cosine")
    if return_centroids:   # (the centroids are the first B*K*C floats of the scratch area, include/wvn_hip.h)
        return labels, nseg, scratch.view(torch.float32)[: B * K * Cc].reshape(B, K, Cc).clone()
    return labels, nseg


def kmeans_cosine_pixels_supported(G: int, H: int, C: int, K: int) -> bool:
    """Whether ``wvn_kmeans_cosine_pixels`` has an instantiation for this shape (csrc/stego.hip: code dimension 90 or 16, up to 64
    clusters, the two staged code rows within 96 KB of LDS); callers fall back to the dense rows + ``kmeans_cosine`` otherwise."""
    return C in (90, 16) and 0 < K <= 64 and 2 * G * C * 4 <= 96 * 1024 and G > 0 and H > 0


def kmeans_pixels_linear_supported(G: int, H: int, C: int, K: int) -> bool:
    """Whether the linear form of the pixel k-means (``wvn_kmeans_cosine_pixels_linear``, csrc/stego_linear.hip) has an instantiation
    for this shape (code dimension 90 or 16, up to 32 clusters, its per-band tables within the LDS)."""
    return bool(lib().wvn_kmeans_pixels_linear_supported_shape(G, H, C, K))


def kmeans_cosine_pixels(code: torch.Tensor, G: int, H: int, K: int, iters: int = 10, relabel: bool = True,
                         return_centroids: bool = False, form: str = "linear", align_corners: bool = True):
    """code [B, G*G, C] fp32 patch codes -> (labels [B, H*H] int32, n_segments [B] int32): the k-means of ``kmeans_cosine`` over
    the H x H bilinearly up-sampled, normalised code pixels (the [B, H*H, C] array is never built).
    form="linear" (default, oracle/kmeans_linear.py): assignment from a per-pass similarity table interpolated per pixel, centroid
    sums from summed tap weights times the patch codes -- the same clustering at ~1/20 of the arithmetic; form="direct"
    (oracle/interfaces.py::kmeans_cosine_labels_pixels): every row re-created and multiplied out in every pass.  Both are
    deterministic and bit-exact against their oracle; they differ from each other only where two similarities tie within fp32 rounding.
    align_corners=False (linear form only): the code pixels are the half-pixel (align_corners=False) bilinear interpolation of the patch codes --
    the other reading of the absent STEGO package (include/wvn_hip.h: wvn_kmeans_cosine_pixels_linear_ac)."""
    require_cuda(code, "code")
    if not align_corners and form != "linear":
        raise _lib.WvnError("kmeans_cosine_pixels: align_corners=False exists in the linear form only")
    B, P, Cc = code.shape
    if P != G * G:
        raise _lib.WvnError(f"kmeans_cosine_pixels: {P} code rows for a {G} x {G} grid")
    if form not in ("linear", "direct"):
        raise _lib.WvnError(f"kmeans_cosine_pixels: form must be 'linear' or 'direct', not {form!r}")
    code = code.contiguous()
    dev = code.device
    labels = torch.empty(B, H * H, dtype=torch.int32, device=dev)
    nseg = torch.empty(B, dtype=torch.int32, device=dev)
    if form == "linear":
        if not kmeans_pixels_linear_supported(G, H, Cc, K):
            raise _lib.WvnError(f"kmeans_cosine_pixels(form='linear'): no instantiation for G={G}, H={H}, C={Cc}, K={K} "
                                "(C in {16, 90}, K <= 32); use form='direct'")
        scratch = torch.empty(lib().wvn_kmeans_pixels_linear_scratch_bytes(B, G, H, Cc, K), dtype=torch.uint8, device=dev)
        check(lib().wvn_kmeans_cosine_pixels_linear_ac(ptr(code), ptr(labels), ptr(nseg), ptr(scratch), B, G, H, Cc, K, iters,
                                                       int(relabel), int(bool(align_corners)), stream()), "wvn_kmeans_cosine_pixels_linear_ac")
    else:
        scratch = torch.empty(lib().wvn_kmeans_pixels_scratch_bytes(B, G, H, Cc, K), dtype=torch.uint8, device=dev)
        check(lib().wvn_kmeans_cosine_pixels(ptr(code), ptr(labels), ptr(nseg), ptr(scratch), B, G, H, Cc, K, iters, int(relabel),
                                             stream()), "wvn_kmeans_cosine_pixels")
    if return_centroids:
        return labels, nseg, scratch.view(torch.float32)[: B * K * Cc].reshape(B, K, Cc).clone()
    return labels, nseg


def table_bilerp_argmax(table: torch.Tensor, G: int, H: int, align_corners: bool = True) -> torch.Tensor:
    """table [B, G*G, K] fp32 (K <= 32 scores per patch) -> labels [B, H, H] int32: per pixel the first-maximum argmax of the
    bilinearly interpolated (align_corners as given, fixed operation order) scores -- a linear probe at pixel resolution."""
    require_cuda(table, "table")
    B, P, K = table.shape
    if P != G * G:
        raise _lib.WvnError(f"table_bilerp_argmax: {P} rows for a {G} x {G} grid")
    KP = lib().wvn_table_argmax_slots(K)
    if KP <= 0:
        raise _lib.WvnError(f"table_bilerp_argmax: 1..32 scores per patch are supported, not {K}")
    t = table.float()
    t = torch.nn.functional.pad(t, (0, KP - K)).contiguous() if KP != K else t.contiguous()
    labels = torch.empty(B, H, H, dtype=torch.int32, device=table.device)
    check(lib().wvn_table_bilerp_argmax_ac(ptr(t), ptr(labels), B, G, H, K, int(bool(align_corners)), stream()), "wvn_table_bilerp_argmax_ac")
    return labels


def flip_average(code: torch.Tensor, mirrored: torch.Tensor, G: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """0.5 * (code + flip_x(mirrored)) on [B, G*G, C] fp32 patch maps (in place on ``code`` unless ``out`` is given)."""
    require_cuda(code, "code")
    B, P, Cc = code.shape
    if not (code.is_contiguous() and mirrored.is_contiguous()) or mirrored.shape != code.shape or P != G * G:
        raise _lib.WvnError("flip_average: contiguous [B, G*G, C] fp32 pairs expected")
    dst = code if out is None else out
    if out is not None and (not out.is_contiguous() or out.shape != code.shape or out.dtype != torch.float32):
        raise _lib.WvnError("flip_average: out must be a contiguous fp32 tensor of the inputs' shape")
    check(lib().wvn_flip_average(ptr(code), ptr(mirrored), ptr(dst), B, G, Cc, stream()), "wvn_flip_average")
    return dst


def cast_rows_bf16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[r, :cols] = src[r, :] rounded to dst's 16-bit format (bf16 / fp16); fp32 rows with any row stride."""
    require_cuda(src, "src")
    R, Cc = src.shape
    if src.stride(1) != 1 or dst.stride(1) != 1 or dst.shape[0] != R or dst.shape[1] < Cc or dst.dtype not in (torch.bfloat16, torch.float16):
        raise _lib.WvnError("cast_rows_bf16: fp32 [R, C] rows -> 16-bit [R, >= C] rows expected")
    check(lib().wvn_cast_rows(ptr(src), src.stride(0), ptr(dst), dst.stride(0), R, Cc, int(dst.dtype == torch.float16), stream()),
          "wvn_cast_rows")
    return dst


def compact_segment_rows(feat: torch.Tensor, nseg: torch.Tensor, side: Optional[torch.Tensor] = None):
    """feat [B,S,D] fp32 (rows of ids a frame did not produce are NaN), nseg [B] int32 (ids that exist per frame), side
    [B,S,Ds] fp32 (optional per-row data) -> (x [B*S, D], side_out [B*S, Ds] | None, rows_dev int32 [1]): the existing rows
    front to back, zeros behind, the count on the device -- no host synchronisation (``MlpTrainer.train_step(rows_dev=...)``)."""
    require_cuda(feat, "feat")
    B, S, D = feat.shape
    feat = feat.contiguous()
    nseg = nseg.to(torch.int32).contiguous()
    x = torch.empty(B * S, D, dtype=torch.float32, device=feat.device)
    so, Ds = None, 0
    if side is not None:
        side = side.float().contiguous()
        Ds = side.shape[-1]
        so = torch.empty(B * S, Ds, dtype=torch.float32, device=feat.device)
    cnt = torch.empty(1, dtype=torch.int32, device=feat.device)
    check(lib().wvn_compact_segment_rows(ptr(feat), D, ptr(side), Ds, ptr(nseg), B, S, ptr(x), ptr(so), ptr(cnt), stream()),
          "wvn_compact_segment_rows")
    return x, so, cnt


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    """x [R, C] fp32 -> int32 [R] index of the row maximum (lowest index wins ties)."""
    require_cuda(x, "x")
    if x.stride(1) != 1:
        x = x.contiguous()
    R, Cc = x.shape
    out = torch.empty(R, dtype=torch.int32, device=x.device)
    check(lib().wvn_argmax_rows(ptr(x), x.stride(0), R, Cc, ptr(out), stream()), "wvn_argmax_rows")
    return out


def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """x [R, C] fp32 -> x / max(||x||, 1e-12) (sequential fp32, the k-means kernels' normalisation)."""
    require_cuda(x, "x")
    R, Cc = x.shape
    out = torch.empty(R, Cc, dtype=torch.float32, device=x.device)
    check(lib().wvn_normalize_rows(ptr(x), x.stride(0), ptr(out), R, Cc, stream()), "wvn_normalize_rows")
    return out


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epi: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epilogue(a[M,K] @ w[N,K]^T + bias); a, w bf16 or fp16 (both the same; row strides allowed).  The operand format picks
    the library entry: wvn_gemm_bf16 / wvn_gemm_f16 (the same tile kernel compiled per format, csrc/operand.h)."""
    M, K = a.shape
    N = w.shape[0]
    if a.dtype != w.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        raise _lib.WvnError(f"gemm_bf16: operands must both be bfloat16 or both float16, got {a.dtype} / {w.dtype}")
    if out is None:
        dt = a.dtype if epi in (_lib.EPI_BF16, _lib.EPI_GELU_BF16, _lib.EPI_RELU_BF16) else torch.float32
        out = torch.empty(M, N, dtype=dt, device=a.device)
    fn, name = (lib().wvn_gemm_f16, "wvn_gemm_f16") if a.dtype == torch.float16 else (lib().wvn_gemm_bf16, "wvn_gemm_bf16")
    check(fn(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), M, N, K, epi, stream()), name)
    return out


gemm_lowp = gemm_bf16


def resize_nearest_crop(img: torch.Tensor, tables) -> torch.Tensor:
    """out[..., y, x] = img[..., tables.rows[y], tables.cols[x]] (uint8 / fp32 / int32 images [..., H, W]): T.Resize(NEAREST) +
    T.CenterCrop of ImageProjector.resize_image (image_projector.py:199-200) as one gather (``transforms.ingest_tables``)."""
    require_cuda(img, "img")
    if img.shape[-2] != tables.src_h or img.shape[-1] != tables.src_w or img.element_size() not in (1, 4):
        raise _lib.WvnError(f"resize_nearest_crop: image {tuple(img.shape)} {img.dtype} does not match the tables ({tables.src_h} x {tables.src_w})")
    img = img.contiguous()
    planes = img.numel() // (tables.src_h * tables.src_w)
    out = torch.empty(*img.shape[:-2], tables.out_h, tables.out_w, dtype=img.dtype, device=img.device)
    check(lib().wvn_resize_nearest_crop(ptr(img), ptr(out), planes, tables.src_h, tables.src_w, ptr(tables.rows), ptr(tables.cols),
                                        tables.out_h, tables.out_w, img.element_size(), stream()), "wvn_resize_nearest_crop")
    return out


def gemm_f32(a, b, bias=None, epi=_lib.F32_NONE, trans_a=False, trans_b=True, out=None, mask=None):
    """fp32 GEMM; trans_b=True means b is stored [N,K] (Linear layout)."""
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    check(lib().wvn_gemm_f32(ptr(a), a.stride(0), int(trans_a), ptr(b), b.stride(0), int(trans_b), ptr(bias), ptr(out),
                             out.stride(0), M, N, K, epi, ptr(mask), 0 if mask is None else mask.stride(0), stream()),
          "wvn_gemm_f32")
    return out


def quantize_rows_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [R, C] fp32 or bf16 (row stride allowed) -> (q [R, C] float8_e4m3fn, scale [R] fp32): x ~ q * scale, scale = amax / 448."""
    require_cuda(x, "x")
    R, Cc = x.shape
    q = torch.empty(R, Cc, dtype=torch.float8_e4m3fn, device=x.device)
    sc = torch.empty(R, dtype=torch.float32, device=x.device)
    check(lib().wvn_quantize_rows_fp8(ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), ptr(q), Cc, ptr(sc), R, Cc, stream()),
          "wvn_quantize_rows_fp8")
    return q, sc


def layernorm_fp8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """LayerNorm(x [R, D] fp32 contiguous) quantised per row -> (q [R, D] float8_e4m3fn, scale [R]): LN(x) ~ q * scale."""
    require_cuda(x, "x")
    R, D = x.shape
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise _lib.WvnError("layernorm_fp8: contiguous fp32 rows expected")
    q = torch.empty(R, D, dtype=torch.float8_e4m3fn, device=x.device)
    sc = torch.empty(R, dtype=torch.float32, device=x.device)
    check(lib().wvn_layernorm_fp8(ptr(x), ptr(gamma), ptr(beta), ptr(q), D, ptr(sc), R, D, eps, stream()), "wvn_layernorm_fp8")
    return q, sc


def gemm_fp8(a_q: torch.Tensor, sa: torch.Tensor, w_q: torch.Tensor, sw: torch.Tensor, bias: Optional[torch.Tensor], epi: int,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epilogue((a_q @ w_q^T) * sa[:, None] * sw[None, :] + bias); a_q [M,K], w_q [N,K] float8_e4m3fn, K % 128 == 0."""
    M, K = a_q.shape
    N = w_q.shape[0]
    if out is None:
        dt = torch.bfloat16 if epi in (_lib.EPI_BF16, _lib.EPI_GELU_BF16) else torch.float32
        out = torch.empty(M, N, dtype=dt, device=a_q.device)
    check(lib().wvn_gemm_fp8(ptr(a_q), a_q.stride(0), ptr(w_q), w_q.stride(0), ptr(sa), ptr(sw), ptr(bias), ptr(out), out.stride(0),
                             M, N, K, epi, stream()), "wvn_gemm_fp8")
    return out


def split_planes(x: torch.Tensor) -> torch.Tensor:
    """fp32 [R, C] (row stride allowed) -> bf16 [2, R, C]: hi = bf16(x), lo = bf16(x - hi) -- the operand representation of
    the exact-mode MFMA kernels (gemm_x3 / attention_x3)."""
    require_cuda(x, "x")
    R, Cc = x.shape
    out = torch.empty(2, R, Cc, dtype=torch.bfloat16, device=x.device)
    check(lib().wvn_split_planes(ptr(x), x.stride(0), ptr(out[0]), ptr(out[1]), Cc, R, Cc, stream()), "wvn_split_planes")
    return out


def gemm_x3(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], epi: int,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Exact-mode GEMM: a [2, M, K] and w [2, N, K] are hi / lo bf16 planes (ops.split_planes / backbone.split_planes);
    returns bf16 planes [2, M, N] for epi in {EPI_BF16, EPI_GELU_BF16, EPI_RELU_BF16}, fp32 [M, N] otherwise."""
    _, M, K = a.shape
    N = w.shape[1]
    planes = epi in (_lib.EPI_BF16, _lib.EPI_GELU_BF16, _lib.EPI_RELU_BF16)
    if out is None:
        out = torch.empty((2, M, N) if planes else (M, N), dtype=torch.bfloat16 if planes else torch.float32, device=a.device)
    c, c_lo, ldc = (out[0], out[1], out.stride(1)) if planes else (out, None, out.stride(0))
    check(lib().wvn_gemm_x3(ptr(a[0]), ptr(a[1]), a.stride(1), ptr(w[0]), ptr(w[1]), w.stride(1), ptr(bias), ptr(c), ptr(c_lo),
                            ldc, M, N, K, epi, stream()), "wvn_gemm_x3")
    return out


def qkv_fused(x: torch.Tensor, ln: Tuple[torch.Tensor, torch.Tensor, float], w: torch.Tensor, bias: Optional[torch.Tensor],
              frames: int, ntok_s: int, npad: int, q_scale: float = 0.0):
    """LayerNorm + QKV projection in one launch (csrc/qkv_fused.hip).  x [frames * ntok_s, 384] fp32, w [1152, 384] bf16 ->
    q, k [frames * 6, npad, 64] bf16 and v^T [frames * 6, 64, npad] bf16 in the attention kernel's layouts.  Slots of tokens
    >= ntok_s and the padding are left untouched (returned zero-initialised)."""
    M, heads = x.shape[0], 6
    n = frames * heads * npad * 64   # the three outputs are carved from ONE allocation: the kernel addresses them through one
    buf = torch.zeros(3 * n, dtype=torch.bfloat16, device=x.device)   # 32-bit buffer descriptor (they must lie within 2 GB)
    q, k, vt = buf[:n].view(frames * heads, npad, 64), buf[n:2 * n].view(frames * heads, npad, 64), buf[2 * n:].view(frames * heads, 64, npad)
    g, b, eps = ln
    check(lib().wvn_qkv_fused(ptr(x), x.stride(0), ptr(g), ptr(b), float(eps), ptr(w), ptr(bias), ptr(q), ptr(k), ptr(vt), heads, npad,
                              ntok_s, float(q_scale), M, stream()), "wvn_qkv_fused")
    return q, k, vt


def qkv_prenorm(xn_frag: torch.Tensor, w_perm: torch.Tensor, bias: Optional[torch.Tensor], M: int, frames: int, ntok_s: int, npad: int,
                q_scale: float = 0.0):
    """The QKV projection of ``qkv_fused`` from rows that are already normalised and laid out as operand fragments
    (``proj_mlp_resident(..., next_ln=...)`` writes them; include/wvn_hip.h).  ``w_perm = w[:, vt_token_order(384)]``."""
    heads = 6
    n = frames * heads * npad * 64
    buf = torch.zeros(3 * n, dtype=w_perm.dtype, device=w_perm.device)
    q, k, vt = buf[:n].view(frames * heads, npad, 64), buf[n:2 * n].view(frames * heads, npad, 64), buf[2 * n:].view(frames * heads, 64, npad)
    fn = lib().wvn_qkv_prenorm_f16 if w_perm.dtype == torch.float16 else lib().wvn_qkv_prenorm
    check(fn(ptr(xn_frag), ptr(w_perm), ptr(bias), ptr(q), ptr(k), ptr(vt), heads, npad, ntok_s, float(q_scale), M, stream()), "wvn_qkv_prenorm")
    return q, k, vt


def unpack_row_fragments(frag: torch.Tensor, M: int) -> torch.Tensor:
    """[M, 384] rows from the fragment layout of ``proj_mlp_resident(next_ln=...)`` (tests): fragment (R, s) holds, for lane l, row
    32 R + (l & 31), columns 16 s + 4 (l >> 5) + {0..3} and 16 s + 8 + 4 (l >> 5) + {0..3}."""
    G = (M + 31) // 32
    f = frag.view(G, 24, 2, 32, 2, 4)          # [R, s, hi, row, half, e]
    rows = f.permute(0, 3, 1, 4, 2, 5)          # [R, row, s, half, hi, e]: column = 16 s + 8 half + 4 hi + e
    return rows.reshape(G * 32, 384)[:M]


def proj_mlp_fused(attn: torch.Tensor, wp: torch.Tensor, bp: Optional[torch.Tensor], ln: Tuple[torch.Tensor, torch.Tensor, float], w1: torch.Tensor,
                   b1: Optional[torch.Tensor], w2p: torch.Tensor, b2: Optional[torch.Tensor], x: torch.Tensor,
                   ls1: Optional[torch.Tensor] = None, ls2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x += (attn @ wp^T + bp) (* ls1); x += MLP(LayerNorm(x)) in ONE launch, in place (csrc/mlp_fused.hip with the projection in its
    prologue)."""
    M, F = x.shape[0], w1.shape[0]
    g, b, eps = ln
    check(lib().wvn_proj_mlp_fused(ptr(attn), attn.stride(0), ptr(wp), ptr(bp), ptr(ls1), ptr(g), ptr(b), float(eps), ptr(w1), ptr(b1),
                                   ptr(w2p), ptr(b2), ptr(ls2), ptr(x), x.stride(0), M, F, stream()), "wvn_proj_mlp_fused")
    return x


def proj_mlp_resident(attn: torch.Tensor, wp: torch.Tensor, bp: Optional[torch.Tensor], ln: Tuple[torch.Tensor, torch.Tensor, float],
                      w1p: torch.Tensor, b1: Optional[torch.Tensor], w2p: torch.Tensor, b2: Optional[torch.Tensor], x: torch.Tensor,
                      next_ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None):
    """``proj_mlp_fused`` without LayerScale, the residual rows resident in the kernel's accumulators (x read once, written once).
    ``w1p`` = fc1.weight with its column index in the fused order: ``w1[:, vt_token_order(384)]``; bf16 or fp16 operands.
    ``next_ln=(gamma, beta, eps)``: also returns LayerNorm(x_new) as operand fragments for ``qkv_prenorm`` (x, fragments)."""
    M, F = x.shape[0], w1p.shape[0]
    g, b, eps = ln
    f16 = attn.dtype == torch.float16
    fn = lib().wvn_proj_mlp_resident_f16 if f16 else lib().wvn_proj_mlp_resident
    frag = None
    ng, nb, ne = (None, None, 0.0)
    if next_ln is not None:
        ng, nb, ne = next_ln
        frag = torch.zeros((M + 31) // 32 * 24 * 512, dtype=attn.dtype, device=x.device)
    check(fn(ptr(attn), attn.stride(0), ptr(wp), ptr(bp), ptr(g), ptr(b), float(eps), ptr(w1p), ptr(b1), ptr(w2p), ptr(b2), ptr(x),
             x.stride(0), M, F, ptr(ng), ptr(nb), float(ne), ptr(frag), stream()), "wvn_proj_mlp_resident")
    return x if next_ln is None else (x, frag)


def mlp_fused(xn: Optional[torch.Tensor], w1: torch.Tensor, b1: Optional[torch.Tensor], w2p: torch.Tensor, b2: Optional[torch.Tensor],
              x: torch.Tensor, ls: Optional[torch.Tensor] = None, ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> torch.Tensor:
    """x [M,384] fp32 += gelu(xn [M,384] bf16 @ w1[F,384]^T + b1) @ w2^T + b2 in ONE launch, in place.  ``w2p`` is fc2.weight
    [384, F] with its hidden index in the fused order: ``w2[:, vt_token_order(F)]`` (include/wvn_hip.h WVN_VIT_MLP_FUSED).
    ``xn=None, ln=(gamma, beta, eps)``: the kernel normalises the rows of ``x`` itself."""
    M, F = x.shape[0], w1.shape[0]
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    if (xn is None) == (ln is None):
        raise _lib.WvnError("mlp_fused takes either xn or ln=(gamma, beta, eps)")
    check(lib().wvn_mlp_fused(ptr(xn), xn.stride(0) if xn is not None else 0, ptr(g), ptr(b), float(eps), ptr(w1), ptr(b1), ptr(w2p),
                              ptr(b2), ptr(ls), ptr(x), x.stride(0), M, F, stream()), "wvn_mlp_fused")
    return x


def vt_token_order(npad: int, device=None) -> torch.Tensor:
    """Index map of the bf16 attention kernel's V^T layout (include/wvn_hip.h): stored position p of every
    aligned group of 16 tokens holds token p with bits 2 and 3 swapped (an involution).
    ``vt_stored = v.transpose(-1, -2)[..., vt_token_order(npad)]``."""
    p = torch.arange(npad, device=device)
    return (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1)


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().wvn_cast_f32_to_bf16(ptr(x), ptr(out), x.numel(), stream()), "wvn_cast_f32_to_bf16")
    return out


def prof_enable(on: bool) -> None:
    check(lib().wvn_prof_enable(int(on)), "wvn_prof_enable")


def prof_collect():
    n = len(_lib.PROF_CATS)
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    check(lib().wvn_prof_collect(ms, cnt), "wvn_prof_collect")
    return {c: (ms[i], cnt[i]) for i, c in enumerate(_lib.PROF_CATS)}


def cu_mask_words(cus_per_xcd_lo: int, cus_per_xcd_hi: int, layout: str = "rr", xcds: int = 8, cus_per_xcd: int = 32):
    """32-bit mask words selecting CUs [lo, hi) of every XCD.  ``layout``: how the driver numbers mask bits -- "rr": bit i is CU
    i // xcds of XCD i % xcds (amdkfd spreads a queue's mask over the XCCs round-robin), "lin": bit i is CU i % cus_per_xcd of XCD
    i // cus_per_xcd."""
    n = xcds * cus_per_xcd
    bits = 0
    for i in range(n):
        cu = i // xcds if layout == "rr" else i % cus_per_xcd
        if cus_per_xcd_lo <= cu < cus_per_xcd_hi:
            bits |= 1 << i
    return [(bits >> (32 * w)) & 0xFFFFFFFF for w in range((n + 31) // 32)]


class MaskedStream:
    """A HIP stream restricted to a set of compute units (``wvn_stream_create_cu_mask``), usable as a torch stream."""

    def __init__(self, words, device=None):
        import ctypes as C
        arr = (C.c_uint32 * len(words))(*words)
        h = C.c_void_p()
        check(lib().wvn_stream_create_cu_mask(C.byref(h), arr, len(words)), "wvn_stream_create_cu_mask")
        self.handle = h.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=device)

    def close(self):
        if self.handle:
            torch.cuda.synchronize()
            check(lib().wvn_stream_destroy(self.handle), "wvn_stream_destroy")
            self.handle = None
