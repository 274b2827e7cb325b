"""Input geometry used by DinoInterface / StegoInterface (dino_interface.py:52-59): torchvision's
``T.Resize(size, NEAREST)`` + ``T.CenterCrop(size)`` restated on torch tensors (torchvision is not a
dependency), and the same geometry as INDEX TABLES for the HIP path: the library gathers the patches
straight from the camera frame (``wvn_vit_forward_frames``), so on the product path no
resized / cropped image is ever built -- ``resize_nearest_center_crop`` itself only runs once per camera
geometry, on two tiny index images, to derive the tables (identical to the image op by construction).
The ImageNet normalisation is fused into the same patch gather."""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def resize_nearest_center_crop(img: torch.Tensor, size: int) -> torch.Tensor:
    H, W = img.shape[-2:]
    if (H, W) == (size, size):
        return img
    if H <= W:
        nh, nw = size, int(size * W / H)
    else:
        nh, nw = int(size * H / W), size
    if (nh, nw) != (H, W):
        img = F.interpolate(img, size=(nh, nw), mode="nearest")
    top = int(round((nh - size) / 2.0))
    left = int(round((nw - size) / 2.0))
    return img[..., top : top + size, left : left + size]


class IngestTables:
    """rows [out_h], cols [out_w] int32 on ``device``: output pixel (y, x) is source pixel (rows[y], cols[x]) of a
    [.., src_h, src_w] frame."""

    def __init__(self, rows: torch.Tensor, cols: torch.Tensor, src_h: int, src_w: int):
        self.rows, self.cols, self.src_h, self.src_w = rows, cols, src_h, src_w
        self.out_h, self.out_w = rows.numel(), cols.numel()
        self.identity = (self.out_h == src_h and self.out_w == src_w
                         and bool((rows.cpu() == torch.arange(src_h, dtype=torch.int32)).all())
                         and bool((cols.cpu() == torch.arange(src_w, dtype=torch.int32)).all()))


_CACHE: Dict[Tuple, IngestTables] = {}


def ingest_tables(src_h: int, src_w: int, size: int, device, out_w: Optional[int] = None, flip: bool = False) -> IngestTables:
    """The geometry of ``resize_nearest_center_crop(img, size)`` (``out_w`` None), or of a plain NEAREST resize to
    [size, out_w] (ImageProjector.resize_image with an explicit width, image_projector.py:56-59), as gather tables; ``flip``:
    of the horizontally mirrored result (STEGO's flip pass reads the same frame through reversed column indices).  Cached per
    geometry: the index images below are the only place the host-side image op ever runs."""
    key = (src_h, src_w, size, out_w, flip, str(device))
    t = _CACHE.get(key)
    if t is None:
        ys = torch.arange(src_h, dtype=torch.float32).view(1, 1, src_h, 1).expand(1, 1, src_h, src_w)
        xs = torch.arange(src_w, dtype=torch.float32).view(1, 1, 1, src_w).expand(1, 1, src_h, src_w)
        if out_w is None:
            ry, rx = resize_nearest_center_crop(ys, size), resize_nearest_center_crop(xs, size)
        else:
            ry, rx = (F.interpolate(v, size=(size, out_w), mode="nearest") for v in (ys, xs))
        rows, cols = ry[0, 0, :, 0].to(torch.int32), rx[0, 0, 0, :].to(torch.int32)
        if flip:
            cols = cols.flip(0)
        t = IngestTables(rows.contiguous().to(device), cols.contiguous().to(device), src_h, src_w)
        _CACHE[key] = t
    return t
