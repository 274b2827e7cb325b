"""DinoInterface -- same constructor / ``inference`` contract as
wild_visual_navigation/feature_extractor/dino_interface.py:15-108, running on the HIP backbone."""
import warnings
from typing import Dict, Optional

import torch

from .. import _lib, ops
from ..backbone import ARCH, VitBackbone, synthetic_vit_state_dict
from .transforms import resize_nearest_center_crop


class _Cfg(dict):
    """dict with attribute access (stand-in for the OmegaConf node the reference keeps in _cfg)."""

    __getattr__ = dict.__getitem__

    def is_empty(self):
        return len(self) == 0


def _load_state_dict(pretrained_weights, backbone_type, patch_size, seed=0, dinov2=False) -> Dict[str, torch.Tensor]:
    """A DINO / DINOv2 state dict: the given dict, a checkpoint file (released DINO ``*_pretrain.pth`` / ``*_checkpoint.pth``
    or DINOv2 hub ``dinov2_vit?14_pretrain.pth`` layouts), or -- there is no network here -- seeded synthetic weights."""
    if isinstance(pretrained_weights, dict):
        return pretrained_weights
    if isinstance(pretrained_weights, str):
        sd = torch.load(pretrained_weights, map_location="cpu")
        for key in ("state_dict", "teacher", "model"):
            if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
                sd = sd[key]
        return {k.replace("module.", "").replace("backbone.", ""): v for k, v in sd.items()}
    # The reference lets the external package download DINO weights; no network here.
    return synthetic_vit_state_dict(backbone_type, patch_size, pretrain_grid=37 if dinov2 else 28, seed=seed, dinov2=dinov2)


class DinoInterface:
    def __init__(
        self,
        device: str,
        backbone: str = "dino",
        input_size: int = 448,
        backbone_type: str = "vit_small",
        patch_size: int = 8,
        projection_type: str = None,
        dropout_p: float = 0,
        pretrained_weights=None,  # path to a DINO checkpoint, or a state dict; None -> seeded synthetic
        cfg=None,
        precision: str = "mixed",  # extension: "mixed" (default: the <= 1e-3 mode the north star's parity clause asks for, on MFMA) | "exact" (every product split) | "fp32" (same gate, FMA) | "fp16" | "bf16" (opt-in speed paths, 11 / 8 significand bits: 4.5e-3 / 3e-2 token error)
        max_chunk: int = 16,
        allow_synthetic: bool = False,
        fuse_mlp: Optional[bool] = None,
        fuse_qkv: Optional[bool] = None,
        fuse_proj: bool = True,  # None: fused block MLP wherever it applies (bf16, dim 384); False: the un-fused pair
        pos_embed_rule: str = "dino",  # position-table resampling: "dino" (scale_factor (G + 0.1) / g, as published) | "size" (HuggingFace's / torch < 1.6's reading)
    ):
        if cfg is None or len(cfg) == 0:
            self._cfg = _Cfg(backbone=backbone, backbone_type=backbone_type, input_size=input_size,
                             patch_size=patch_size, projection_type=projection_type, dropout_p=dropout_p,
                             pretrained_weights=pretrained_weights)
        else:
            self._cfg = _Cfg(cfg)
        if self._cfg.backbone not in ("dino", "dinov2"):
            raise _lib.WvnError(f"backbone '{self._cfg.backbone}' not supported by the MI355X path (dino, dinov2)")
        c = self._cfg
        _, _, heads = ARCH[c.backbone_type]
        dinov2 = c.backbone == "dinov2"
        if dinov2 and c.patch_size != 14:
            raise _lib.WvnError("backbone 'dinov2' has patch size 14 (dinov2_vit{s,b}14)")
        if c.pretrained_weights is None and not allow_synthetic:
            warnings.warn(f"DinoInterface: no pretrained_weights given -- running with seeded SYNTHETIC {c.backbone} "
                          f"{c.backbone_type}/{c.patch_size} weights (the reference downloads the released checkpoint; there is "
                          "no network here): shapes and speed are real, the features are meaningless.  Pass "
                          "pretrained_weights=<path | state dict> or allow_synthetic=True to silence", stacklevel=2)
        sd = _load_state_dict(c.pretrained_weights, c.backbone_type, c.patch_size, dinov2=dinov2)
        self._precision = precision
        self._device = torch.device(device)
        self._model = VitBackbone(sd, c.input_size, c.patch_size, heads, device=self._device, precision=precision,
                                  max_chunk=max_chunk, fuse_mlp=fuse_mlp, fuse_qkv=fuse_qkv, fuse_proj=fuse_proj, pos_embed_rule=pos_embed_rule)

    def change_device(self, device):
        """dino_interface.py:61-68: move the model to another device (another GPU: the HIP path has no CPU form)."""
        device = torch.device(device)
        self._model = self._model.to(device)
        self._device = device

    @torch.no_grad()
    def inference_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] in [0,1] -> patch tokens [B,G*G,D] fp32 (the un-upsampled feature map, NHWC)."""
        # T.Resize(NEAREST) + T.CenterCrop + T.Normalize of dino_interface.py:52-59 all happen inside the patch gather
        return self._model.forward_tokens(img.to(self._device))

    @torch.no_grad()
    def inference(self, img: torch.Tensor) -> torch.Tensor:
        """dino_interface.py:70-92: returns dense per-pixel features [B,D,H,H] fp32 (H = img height for
        BOTH dims, as the reference does)."""
        tok = self.inference_tokens(img)
        return ops.upsample_bilinear(tok, self._model.grid, img.shape[2])

    @property
    def input_size(self):
        return self._cfg.input_size

    @property
    def backbone(self):
        return self._cfg.backbone

    @property
    def backbone_type(self):
        return self._cfg.backbone_type

    @property
    def vit_patch_size(self):
        return self._cfg.patch_size

    @property
    def grid(self):
        return self._model.grid

    @property
    def feature_dim(self):
        return self._model.dim
