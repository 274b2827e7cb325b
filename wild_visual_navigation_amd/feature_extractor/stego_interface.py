"""StegoInterface -- same contract as wild_visual_navigation/feature_extractor/stego_interface.py:18-135.

The STEGO network itself (``stego.stego.Stego``: backbone + segmentation head + cluster / linear
probes, k-means / CRF post-processing) is an external, absent package; this build implements the
published STEGO head (1x1-conv linear branch + 1x1-conv/ReLU/1x1-conv branch, summed) on the HIP
ViT backbone and a deterministic per-image cosine k-means for ``run_clustering=True`` (definition
in DESIGN.md).  ``run_crf=True`` (pydensecrf, CPU) is not available.
"""
from typing import Dict, Optional

import torch

from .. import _lib, ops
from ..backbone import ARCH, VitBackbone
from .dino_interface import _Cfg, _load_state_dict
from .transforms import resize_nearest_center_crop

STEGO_CODE_DIM = 90
KMEANS_ITERS = 10
N_LINEAR_CLASSES = 27


def synthetic_stego_head(D: int = 384, Cc: int = STEGO_CODE_DIM, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(1000 + seed)

    def rn(*s, std):
        return torch.randn(*s, generator=g) * std

    return {
        "cluster1.0.weight": rn(Cc, D, std=0.08), "cluster1.0.bias": rn(Cc, std=0.02),
        "cluster2.0.weight": rn(D, D, std=0.06), "cluster2.0.bias": rn(D, std=0.02),
        "cluster2.2.weight": rn(Cc, D, std=0.08), "cluster2.2.bias": rn(Cc, std=0.02),
    }


class StegoInterface:
    def __init__(
        self,
        device: str,
        input_size: int = 448,
        model_path: Optional[str] = None,
        n_image_clusters: int = 40,
        run_crf: bool = True,
        run_clustering: bool = False,
        cfg=None,
        backbone_type: str = "vit_small",  # extension (the reference's released ckpt is ViT-Base)
        patch_size: int = 8,
        precision: str = "bf16",
        backbone_weights=None,
        head_weights: Optional[Dict[str, torch.Tensor]] = None,
        max_chunk: int = 16,
    ):
        if cfg is None or len(cfg) == 0:
            self._cfg = _Cfg(model_path=model_path, input_size=input_size, run_crf=run_crf,
                             run_clustering=run_clustering, n_image_clusters=n_image_clusters)
        else:
            self._cfg = _Cfg(cfg)
        if self._cfg.run_crf:
            raise _lib.WvnError("run_crf=True needs pydensecrf (CPU, external); FeatureExtractor uses run_crf=False")
        self._device = torch.device(device)
        D, _, heads = ARCH[backbone_type]
        sd = _load_state_dict(backbone_weights, backbone_type, patch_size)
        self._bb = VitBackbone(sd, self._cfg.input_size, patch_size, heads, device=self._device, precision=precision,
                               max_chunk=max_chunk)
        self._precision = precision
        head = head_weights if head_weights is not None else synthetic_stego_head(D)
        head = {k: v.reshape(v.shape[0], -1) if v.dim() > 2 else v for k, v in head.items()}  # conv1x1 -> linear
        dev = self._device
        self._D = D
        self._b_hid = head["cluster2.0.bias"].float().to(dev).contiguous()
        self._b_code = (head["cluster1.0.bias"] + head["cluster2.2.bias"]).float().to(dev).contiguous()
        if precision == "bf16":
            self._w_hid = head["cluster2.0.weight"].to(dev, torch.bfloat16).contiguous()
            self._w_code = torch.cat([head["cluster1.0.weight"], head["cluster2.2.weight"]], dim=1).to(
                dev, torch.bfloat16).contiguous()  # [C, 2D] acting on [tok | hid]
        else:
            self._w_hid = head["cluster2.0.weight"].float().to(dev).contiguous()
            self._w_lin = head["cluster1.0.weight"].float().to(dev).contiguous()
            self._w_nl = head["cluster2.2.weight"].float().to(dev).contiguous()
            self._b_lin = head["cluster1.0.bias"].float().to(dev).contiguous()
            self._b_nl = head["cluster2.2.bias"].float().to(dev).contiguous()
        g = torch.Generator().manual_seed(77)
        self._w_probe = (torch.randn(N_LINEAR_CLASSES, STEGO_CODE_DIM, generator=g) * 0.1).to(dev)
        self._model = self  # the reference exposes `.model`
        self._cmap = None
        self._code = None
        self._code_tokens = None
        self._cluster_pred = None
        self._linear_pred = None
        self._n_segments = None

    def change_device(self, device):
        if torch.device(device) != self._device:
            raise _lib.WvnError("change_device: weights are bound to the GPU they were built on")

    # ---- code (STEGO head) at patch resolution --------------------------------------------------------
    @torch.no_grad()
    def code_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] in [0,1] -> STEGO code [B, G*G, 90] fp32 (patch resolution)."""
        img = resize_nearest_center_crop(img.to(self._device), self._cfg.input_size)
        B = img.shape[0]
        P, D = self._bb.grid ** 2, self._D
        if self._precision == "bf16":
            cat = torch.empty(B * P, 2 * D, dtype=torch.bfloat16, device=self._device)  # [tok | hid]
            self._bb.forward_tokens(img, lowp_out=cat)
            ops.gemm_bf16(cat[:, :D], self._w_hid, self._b_hid, _lib.EPI_RELU_BF16, out=cat[:, D:])
            code = ops.gemm_bf16(cat, self._w_code, self._b_code, _lib.EPI_F32)
        else:
            tok = self._bb.forward_tokens(img).reshape(B * P, D)
            hid = ops.gemm_f32(tok, self._w_hid, self._b_hid, _lib.F32_RELU)
            code = ops.gemm_f32(tok, self._w_lin, self._b_lin, _lib.F32_NONE)
            ops.gemm_f32(hid, self._w_nl, self._b_nl, _lib.F32_RESID, out=code)
        return code.reshape(B, P, STEGO_CODE_DIM)

    @torch.no_grad()
    def inference(self, img: torch.Tensor, code: torch.Tensor = None):
        """stego_interface.py:73-111: returns (linear_pred, cluster_pred), both [1,B,H,H] int32, and keeps
        ``features`` = code [B,90,H,H] (bilinear, align_corners=True).  ``code``: the result of ``code_tokens(img)`` if
        the caller already ran that stage (e.g. on another stream, FeatureExtractor.backbone_stage)."""
        G = self._bb.grid
        H = img.shape[2]
        if code is None:
            code = self.code_tokens(img)
        B = code.shape[0]
        self._code_tokens = code
        if self._cfg.run_clustering:
            labels, nseg = ops.kmeans_cosine(code, self._cfg.n_image_clusters, KMEANS_ITERS, relabel=True)
            self._n_segments = nseg
        else:
            # cluster probe: cosine similarity against learned centroids -- not part of the hot path; use
            # k-means semantics with the probe as fixed centroids is not defined upstream here.
            raise _lib.WvnError("run_clustering=False (learned cluster probe) needs the STEGO checkpoint's probe")
        self._labels_patch = labels.reshape(B, G, G)  # patch-resolution cluster ids (before the nearest up-sampling)
        self._cluster_pred = ops.upsample_nearest_labels(self._labels_patch, H)[None]
        logits = ops.gemm_f32(code.reshape(B * G * G, -1), self._w_probe)
        lin = logits.argmax(dim=1).to(torch.int32).reshape(B, G, G)
        self._linear_pred = ops.upsample_nearest_labels(lin, H)[None]
        self._code = None  # dense code is produced lazily (features property): 72 MB/frame at 448^2
        self._H = H
        return self._linear_pred, self._cluster_pred

    @property
    def model(self):
        return self._model

    @property
    def cmap(self):
        return self._cmap

    @property
    def input_size(self):
        return self._cfg.input_size

    @property
    def linear_segments(self):
        return self._linear_pred

    @property
    def cluster_segments(self):
        return self._cluster_pred

    @property
    def features(self):
        if self._code is None and self._code_tokens is not None:
            self._code = ops.upsample_bilinear(self._code_tokens, self._bb.grid, self._H)
        return self._code

    @property
    def feature_tokens(self):
        return self._code_tokens

    @property
    def grid(self):
        return self._bb.grid
