"""Oracle: segmentation maps, graph structure and per-segment pooling.  TEST INFRASTRUCTURE ONLY.

Restates (vectorised, CPU) wild_visual_navigation/feature_extractor/feature_extractor.py:179-249,
310-398, segment_extractor.py:39-92 and traversability_estimator/nodes.py:400-440.  PINNED against
the reference's own code by oracle/pin_reference.py (fixtures in tests/golden/).
"""
from typing import Tuple

import numpy as np
import torch


def segment_grid(H: int, W: int, cell_size: int = 32) -> torch.Tensor:
    """feature_extractor.py:198-219 (kornia extract/combine_tensor_patches with window==stride):
    id = row-major index of the cell_size x cell_size cell.  Returns [1,1,H,W] int64.
    The reference's patch extraction only covers whole cells; H, W must be multiples of cell_size."""
    assert H % cell_size == 0 and W % cell_size == 0
    gy = torch.arange(H) // cell_size
    gx = torch.arange(W) // cell_size
    seg = gy[:, None] * (W // cell_size) + gx[None, :]
    return seg[None, None].to(torch.int64)


def segment_random(H: int, W: int, indices: torch.Tensor) -> torch.Tensor:
    """feature_extractor.py:227-235 given the already-drawn pixel permutation prefix ``indices``:
    seg = -1 everywhere, seg.flat[indices[j]] = j.  Returns [H,W] int64."""
    seg = torch.full((H * W,), -1, dtype=torch.long)
    seg[indices] = torch.arange(indices.numel())
    return seg.reshape(H, W)


def adjacency_list(seg: torch.Tensor) -> torch.Tensor:
    """segment_extractor.py:39-67.  seg [1,1,H,W] int64 -> [E,2] int64.

    The reference runs four 3x3 difference filters with replicate padding 3 (conv output
    (H+4)x(W+4)) and crops [2:-2], which leaves HxW masks aligned with the image whose
    out-of-image neighbours are clamped.  Net effect at image pixel (y,x):
        m0 = s[y,x] != s[y,x+1]    m1 = s[y,x-1] != s[y,x]
        m2 = s[y,x] != s[y+1,x]    m3 = s[y-1,x] != s[y,x]
    left ids  = s[m0] ++ s[m2], right ids = s[m1] ++ s[m3] (row-major order each), paired
    position-wise, key = left + right*(max+1), unique-sorted, decoded as (key % div, key // div).
    """
    s = seg[0, 0]
    H, W = s.shape
    xr = torch.cat([s[:, 1:], s[:, -1:]], dim=1)  # s[y, x+1] clamped
    xl = torch.cat([s[:, :1], s[:, :-1]], dim=1)  # s[y, x-1] clamped
    yd = torch.cat([s[1:, :], s[-1:, :]], dim=0)  # s[y+1, x] clamped
    yu = torch.cat([s[:1, :], s[:-1, :]], dim=0)  # s[y-1, x] clamped
    m0, m1, m2, m3 = s != xr, xl != s, s != yd, yu != s
    left = torch.cat([s[m0], s[m2]])
    right = torch.cat([s[m1], s[m3]])
    div = int(s.max()) + 1
    key = torch.unique(left + right * div)
    return torch.stack([key % div, key // div], dim=1)


def centers(seg: torch.Tensor) -> torch.Tensor:
    """segment_extractor.py:70-92: per-segment mean of pixel coordinates, returned as (x, y) because
    the reference transposes the map before torch.nonzero.  seg [1,1,H,W] -> [S,2] fp32 (NaN for an
    empty id)."""
    s = seg[0, 0]
    H, W = s.shape
    S = int(s.max()) + 1
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    flat = s.reshape(-1)
    keep = flat >= 0
    cnt = torch.zeros(S, dtype=torch.float64).index_add_(0, flat[keep], torch.ones(int(keep.sum()), dtype=torch.float64))
    sx = torch.zeros(S, dtype=torch.float64).index_add_(0, flat[keep], xs.reshape(-1)[keep].double())
    sy = torch.zeros(S, dtype=torch.float64).index_add_(0, flat[keep], ys.reshape(-1)[keep].double())
    return torch.stack([sx / cnt, sy / cnt], dim=1).float()


def sparsify_features(dense: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
    """feature_extractor.py:390-396: feat[i] = mean over {seg == i} of dense[0, :, y, x];
    NaN row for an id with no pixel.  dense [1,D,H,W] fp32, seg [H,W] int64 -> [S,D] fp32.
    (Accumulated in fp64 here; the reference's fp32 mean agrees to ~1e-6.)"""
    D = dense.shape[1]
    S = int(seg.max()) + 1
    flat = seg.reshape(-1)
    keep = flat >= 0
    vals = dense[0].permute(1, 2, 0).reshape(-1, D).double()[keep]
    sums = torch.zeros(S, D, dtype=torch.float64).index_add_(0, flat[keep], vals)
    cnt = torch.zeros(S, dtype=torch.float64).index_add_(0, flat[keep], torch.ones(int(keep.sum()), dtype=torch.float64))
    return (sums / cnt[:, None]).float()


def update_supervision_signal(mask: torch.Tensor, seg: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """nodes.py:400-440 (label pooling).  mask [3,H,W] fp32 with NaN = unlabeled, seg [H,W] int64.
    signal = nanmean over channels; per segment: sum(non-NaN signal) / count(non-NaN) ; NaN -> 0;
    valid = signal > 0.  Returns ([S] fp32, [S] bool)."""
    isn = torch.isnan(mask)
    c = (~isn).sum(0)
    sig = torch.where(isn, torch.zeros_like(mask), mask).sum(0) / c  # NaN where all channels NaN
    S = int(seg.max()) + 1
    flat = seg.reshape(-1)
    sflat = sig.reshape(-1)
    ok = ~torch.isnan(sflat)
    cnt = torch.zeros(S, dtype=torch.float64).index_add_(0, flat[ok], torch.ones(int(ok.sum()), dtype=torch.float64))
    tot = torch.zeros(S, dtype=torch.float64).index_add_(0, flat[ok], sflat[ok].double())
    mean = (tot / cnt).float()
    mean = torch.nan_to_num(mean, nan=0.0)
    return mean, mean > 0


def bilinear_ac_taps(out: int, grid: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Source index / weight of align_corners=True bilinear resampling grid -> out
    (ATen upsample_bilinear2d semantics: src = dst * (grid-1)/(out-1) in fp32, i0 = floor,
    i1 = min(i0+1, grid-1), w1 = src - i0)."""
    scale = np.float32(grid - 1) / np.float32(out - 1) if out > 1 else np.float32(0)
    src = (np.arange(out, dtype=np.float32) * scale).astype(np.float32)
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, grid - 1)
    i1 = np.minimum(i0 + 1, grid - 1)
    w1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, w1
