"""Input geometry used by DinoInterface / StegoInterface (dino_interface.py:52-59): torchvision's
``T.Resize(size, NEAREST)`` + ``T.CenterCrop(size)`` restated on torch tensors (torchvision is not a
dependency).  Pure indexing / plumbing; the ImageNet normalisation itself is fused into the HIP
patchify kernel."""
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
