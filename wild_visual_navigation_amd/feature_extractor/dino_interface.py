"""DinoInterface -- same constructor / ``inference`` contract as
wild_visual_navigation/feature_extractor/dino_interface.py:15-108, running on the HIP backbone."""
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


def _load_state_dict(pretrained_weights, backbone_type, patch_size, seed=0) -> Dict[str, torch.Tensor]:
    if isinstance(pretrained_weights, dict):
        return pretrained_weights
    if isinstance(pretrained_weights, str):
        sd = torch.load(pretrained_weights, map_location="cpu")
        for key in ("state_dict", "teacher", "model"):
            if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
                sd = sd[key]
        return {k.replace("module.", "").replace("backbone.", ""): v for k, v in sd.items()}
    # The reference lets the external package download DINO weights; no network here.
    return synthetic_vit_state_dict(backbone_type, patch_size, seed=seed)


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
        precision: str = "bf16",  # extension: "bf16" (MFMA) | "fp32" (exact parity mode)
        max_chunk: int = 16,
    ):
        if cfg is None or len(cfg) == 0:
            self._cfg = _Cfg(backbone=backbone, backbone_type=backbone_type, input_size=input_size,
                             patch_size=patch_size, projection_type=projection_type, dropout_p=dropout_p,
                             pretrained_weights=pretrained_weights)
        else:
            self._cfg = _Cfg(cfg)
        if self._cfg.backbone != "dino":
            raise _lib.WvnError(f"backbone '{self._cfg.backbone}' not supported by the MI355X path (dino only)")
        c = self._cfg
        _, _, heads = ARCH[c.backbone_type]
        sd = _load_state_dict(c.pretrained_weights, c.backbone_type, c.patch_size)
        self._precision = precision
        self._device = torch.device(device)
        self._model = VitBackbone(sd, c.input_size, c.patch_size, heads, device=self._device, precision=precision,
                                  max_chunk=max_chunk)

    def change_device(self, device):
        device = torch.device(device)
        if device != self._device:
            raise _lib.WvnError("change_device: weights are bound to the GPU they were built on")

    @torch.no_grad()
    def inference_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] in [0,1] -> patch tokens [B,G*G,D] fp32 (the un-upsampled feature map, NHWC)."""
        img = img.to(self._device)
        return self._model.forward_tokens(resize_nearest_center_crop(img, self._cfg.input_size))

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
