"""StegoInterface -- same contract as wild_visual_navigation/feature_extractor/stego_interface.py:18-135.

The STEGO network itself (``stego.stego.Stego``: backbone + segmentation head + cluster / linear probes, k-means / CRF
post-processing) is an external, absent package (PARITY UNPINNED); this build implements the published
STEGO head (1x1-conv linear branch + 1x1-conv/ReLU/1x1-conv branch, summed) on the HIP ViT backbone, the cluster probe
(cosine-similarity argmax against the checkpoint's learned centroids, ``run_clustering=False``), the linear probe, and a
deterministic per-image cosine k-means for ``run_clustering=True``.  ``model_path`` is honoured: a Lightning checkpoint in
either the upstream STEGO (``net.model.* / net.cluster1.* / cluster_probe.clusters / linear_probe.*``) or the
self_supervised_segmentation (``backbone.* / segmentation_head.*``) key layout is loaded (``load_stego_checkpoint``); without
one, seeded synthetic weights are used and a warning says so.  ``run_crf=True`` -- the reference constructor's own default -- needs
pydensecrf (CPU, external): it raises unless the caller opts in to running WITHOUT the CRF refinement (``skip_crf=True`` or the
environment variable ``WVN_SKIP_CRF=1``), in which case a warning says that the CRF step was dropped.

Two knobs make the definition explicit instead of implicit.  Their DEFAULTS are the upstream behaviour as published (Stego.get_code
averages the code with the flipped-back code of the mirrored frame; postprocess clusters the code up-sampled to the image size);
the cheap forms are opt-in, and bench.py names which one each of its legs ran:
  flip_tta            True | False : average the code with the code of the horizontally flipped frame (a second backbone pass; the
                                     mirror is a reversed column table of the fused ingest, no flipped frame exists)
  cluster_resolution  "pixel" | "patch": k-means over the H x H bilinearly up-sampled code pixels (the dense code never exists) or
                                     over the G x G patch codes (labels then nearest-upsampled: segments are patch-aligned).  The
                                     probes of a checkpoint (run_clustering=False, linear_pred) follow the same reading: at
                                     "pixel" their per-patch scores are interpolated and the argmax is taken per pixel
  kmeans_form         "linear" | "direct": how the pixel-resolution k-means is evaluated -- through its linearity (a similarity
                                     table per pass interpolated per pixel, centroid sums from summed tap weights; ~1/20 of the
                                     arithmetic, oracle/kmeans_linear.py) or row by row (oracle/interfaces.py); same clustering,
                                     maps equal except at fp32 rounding ties
"""
import re
import warnings
from typing import Dict, Optional, Tuple

import torch

from .. import _lib, ops
from ..backbone import ARCH, VitBackbone, split_planes
from .dino_interface import _Cfg, _load_state_dict

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


def load_stego_checkpoint(path_or_dict) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Lightning checkpoint of ``Stego.load_from_checkpoint`` (stego_interface.py:43) -> (backbone state dict in the DINO
    layout, head state dict ``cluster1.0 / cluster2.0 / cluster2.2``, probes ``{clusters [K,C], linear.weight [n,C],
    linear.bias [n]}``).  Keys are located by suffix, so both the upstream STEGO naming (``net.model.blocks...``,
    ``net.cluster1.0.weight``) and the ``backbone. / segmentation_head.{linear,nonlinear}`` naming load; 1x1-conv weights
    [out, in, 1, 1] are flattened to Linear layout.  Pure host code (no GPU needed)."""
    ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu", weights_only=False)
    sd = ck.get("state_dict", ck)
    anchor = [k for k in sd if k.endswith("patch_embed.proj.weight")]
    if not anchor:
        raise _lib.WvnError("STEGO checkpoint: no '...patch_embed.proj.weight' key (not a DINO-backbone checkpoint)")
    prefix = anchor[0][: -len("patch_embed.proj.weight")]
    backbone = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}

    def find(pattern):
        hits = [k for k in sd if re.search(pattern, k) and not k.startswith(prefix)]
        return sd[hits[0]] if hits else None

    def lin(t):
        return None if t is None else t.reshape(t.shape[0], -1).float()

    head = {
        "cluster1.0.weight": lin(find(r"(cluster1|segmentation_head\.linear)\.0\.weight$")),
        "cluster1.0.bias": find(r"(cluster1|segmentation_head\.linear)\.0\.bias$"),
        "cluster2.0.weight": lin(find(r"(cluster2|segmentation_head\.nonlinear)\.0\.weight$")),
        "cluster2.0.bias": find(r"(cluster2|segmentation_head\.nonlinear)\.0\.bias$"),
        "cluster2.2.weight": lin(find(r"(cluster2|segmentation_head\.nonlinear)\.2\.weight$")),
        "cluster2.2.bias": find(r"(cluster2|segmentation_head\.nonlinear)\.2\.bias$"),
    }
    missing = [k for k, v in head.items() if v is None]
    if missing:
        raise _lib.WvnError(f"STEGO checkpoint: segmentation-head tensors not found: {missing}")
    probes = {}
    t = find(r"cluster_probe\.clusters$")
    if t is not None:
        probes["clusters"] = t.float()
    t = find(r"linear_probe\.weight$")
    if t is not None:
        probes["linear.weight"] = lin(t)
        b = find(r"linear_probe\.bias$")
        probes["linear.bias"] = b.float() if b is not None else torch.zeros(t.shape[0])
    return backbone, head, probes


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
        backbone_type: str = None,     # extension: default = what the weights say (the released ckpt is ViT-Base), else vit_small
        patch_size: int = 8,
        precision: str = "mixed",      # "mixed" (default: <= 1e-3 of the fp32 reference) | "exact" | "fp32" | "fp16" / "bf16" (opt-in speed paths) | "fp8"
        backbone_weights=None,
        head_weights: Optional[Dict[str, torch.Tensor]] = None,
        probe_weights: Optional[Dict[str, torch.Tensor]] = None,
        max_chunk: int = 16,
        flip_tta: bool = True,
        cluster_resolution: str = "pixel",
        kmeans_form: str = "linear",   # "linear" | "direct": the two statements of the pixel-resolution k-means (ops.kmeans_cosine_pixels)
        allow_synthetic: bool = False,
        fuse_mlp: Optional[bool] = None,
        fuse_qkv: Optional[bool] = None,
        fuse_proj: bool = True,
        skip_crf: Optional[bool] = None,  # run_crf=True without pydensecrf: None -> WVN_SKIP_CRF env, True -> warn and drop the CRF step
        pos_embed_rule: str = "dino",   # position-table resampling of the backbone (backbone.resample_pos_embed)
        code_align_corners: bool = True,   # how postprocess() up-samples the code BEFORE clustering / probing (the absent package's choice, stego_interface.py:94-100):
        #                                    True = the align_corners=True taps of WVN's own later up-sample (:107); False = the half-pixel taps of the public STEGO
        #                                    evaluation code.  `features` (stego_interface.py:107) and the pooling are WVN's own code: always align_corners=True
    ):
        if cfg is None or len(cfg) == 0:
            self._cfg = _Cfg(model_path=model_path, input_size=input_size, run_crf=run_crf,
                             run_clustering=run_clustering, n_image_clusters=n_image_clusters)
        else:
            self._cfg = _Cfg(cfg)
        if self._cfg.run_crf:
            import os

            if skip_crf is None:
                skip_crf = os.environ.get("WVN_SKIP_CRF", "0") not in ("", "0")
            if not skip_crf:
                raise _lib.WvnError("run_crf=True needs pydensecrf (CPU, external); FeatureExtractor uses run_crf=False.  Pass "
                                    "skip_crf=True (or set WVN_SKIP_CRF=1) to run the segmentation without the CRF refinement")
            warnings.warn("StegoInterface: run_crf=True but the dense CRF (pydensecrf, CPU) is not part of the MI355X path -- the "
                          "cluster / linear predictions are returned WITHOUT CRF refinement (skip_crf)", stacklevel=2)
        if cluster_resolution not in ("patch", "pixel"):
            raise _lib.WvnError("cluster_resolution must be 'patch' or 'pixel'")
        self._device = torch.device(device)
        probes = dict(probe_weights) if probe_weights else {}
        mp = self._cfg.get("model_path")
        if mp is not None and backbone_weights is None and head_weights is None:
            import os

            if os.path.exists(mp):
                backbone_weights, head_weights, ck_probes = load_stego_checkpoint(mp)
                probes = {**ck_probes, **probes}
            else:
                warnings.warn(f"StegoInterface: model_path '{mp}' does not exist; falling back to SYNTHETIC weights", stacklevel=2)
        sd = _load_state_dict(backbone_weights, backbone_type or "vit_small", patch_size)
        D = sd["cls_token"].shape[-1]
        if backbone_type is None:
            backbone_type = {v[0]: k for k, v in ARCH.items()}[D]
        if ARCH[backbone_type][0] != D:
            raise _lib.WvnError(f"backbone_type {backbone_type} does not match the weights (dim {D})")
        heads = ARCH[backbone_type][2]
        patch_size = sd["patch_embed.proj.weight"].shape[-1]
        if head_weights is None or backbone_weights is None:
            if not allow_synthetic:
                warnings.warn("StegoInterface: no STEGO checkpoint given -- running with seeded SYNTHETIC backbone / head weights "
                              "(shapes and speed are real, the segmentation is meaningless); pass model_path=... or "
                              "allow_synthetic=True to silence", stacklevel=2)
        self._bb = VitBackbone(sd, self._cfg.input_size, patch_size, heads, device=self._device, precision=precision,
                               max_chunk=max_chunk, fuse_mlp=fuse_mlp, fuse_qkv=fuse_qkv, fuse_proj=fuse_proj, pos_embed_rule=pos_embed_rule)
        self._precision = precision
        self._flip_tta = flip_tta
        self._cluster_resolution = cluster_resolution
        self._code_ac = bool(code_align_corners)
        if not self._code_ac and (cluster_resolution != "pixel" or kmeans_form != "linear"):
            raise _lib.WvnError("code_align_corners=False needs cluster_resolution='pixel' and kmeans_form='linear' (the patch-resolution forms interpolate nothing)")
        if kmeans_form not in ("linear", "direct"):
            raise _lib.WvnError("kmeans_form must be 'linear' or 'direct'")
        self._kmeans_form = kmeans_form
        head = head_weights if head_weights is not None else synthetic_stego_head(D)
        head = {k: v.reshape(v.shape[0], -1) if v.dim() > 2 else v for k, v in head.items()}  # conv1x1 -> linear
        self._head_sd = head
        dev = self._device
        self._D = D
        self._C = head["cluster1.0.weight"].shape[0]
        self._b_hid = head["cluster2.0.bias"].float().to(dev).contiguous()
        self._b_code = (head["cluster1.0.bias"] + head["cluster2.2.bias"]).float().to(dev).contiguous()
        self._b_lin = head["cluster1.0.bias"].float().to(dev).contiguous()
        self._b_nl = head["cluster2.2.bias"].float().to(dev).contiguous()
        if precision in ("bf16", "fp8", "fp16"):   # the (tiny) head stays on the bf16 kernels in the fp8 mode
            lp = self._bb.lowp_dtype
            self._w_hid = head["cluster2.0.weight"].to(dev, lp).contiguous()
            self._w_code = torch.cat([head["cluster1.0.weight"], head["cluster2.2.weight"]], dim=1).to(dev, lp).contiguous()  # [C, 2D] acting on [tok | hid]
        elif precision in ("exact", "mixed"):  # hi / lo planes for the x3 MFMA GEMMs
            self._w_hid = split_planes(head["cluster2.0.weight"].float().to(dev))
            self._w_lin = split_planes(head["cluster1.0.weight"].float().to(dev))
            self._w_nl = split_planes(head["cluster2.2.weight"].float().to(dev))
        else:
            self._w_hid = head["cluster2.0.weight"].float().to(dev).contiguous()
            self._w_lin = head["cluster1.0.weight"].float().to(dev).contiguous()
            self._w_nl = head["cluster2.2.weight"].float().to(dev).contiguous()
        # probes (real weights only: nothing is invented for them)
        self._clusters = None
        if "clusters" in probes:
            c = probes["clusters"].float().to(dev)
            self._clusters = ops.normalize_rows(c.contiguous())          # ClusterLookup: cosine similarity against normalised centroids
        self._w_probe = probes["linear.weight"].float().to(dev).contiguous() if "linear.weight" in probes else None
        self._b_probe = probes["linear.bias"].float().to(dev).contiguous() if "linear.bias" in probes else None
        if not self._cfg.run_clustering and self._clusters is None:
            raise _lib.WvnError("run_clustering=False uses the checkpoint's learned cluster probe; none was loaded "
                                "(pass model_path=<STEGO ckpt> or probe_weights={'clusters': ...})")
        self._model = self  # the reference exposes `.model`
        self._cmap = None
        self._code = None
        self._code_tokens = None
        self._cluster_pred = None
        self._linear_pred = None
        self._n_segments = None
        self._labels_patch = None

    def change_device(self, device):
        """stego_interface.py:60-71: move the model.  Device weights are re-homed (bf16 / plane packs are rebuilt lazily by
        VitBackbone.to)."""
        device = torch.device(device)
        if device == self._device:
            return
        self._bb = self._bb.to(device)
        for name in ("_b_hid", "_b_code", "_b_lin", "_b_nl", "_w_hid", "_w_code", "_w_lin", "_w_nl", "_clusters", "_w_probe", "_b_probe"):
            t = getattr(self, name, None)
            if t is not None:
                setattr(self, name, t.to(device))
        self._device = device

    # ---- code (STEGO head) at patch resolution --------------------------------------------------------
    def _code_once(self, img: torch.Tensor, flip: bool = False) -> torch.Tensor:
        B = img.shape[0]
        P, D = self._bb.grid ** 2, self._D
        if self._precision in ("bf16", "fp8", "fp16"):
            cat = torch.empty(B * P, 2 * D, dtype=self._bb.lowp_dtype, device=self._device)  # [tok | hid]
            self._bb.forward_tokens(img, lowp_out=cat, flip=flip)
            ops.gemm_bf16(cat[:, :D], self._w_hid, self._b_hid, _lib.EPI_RELU_BF16, out=cat[:, D:])
            code = ops.gemm_bf16(cat, self._w_code, self._b_code, _lib.EPI_F32)
        elif self._precision in ("exact", "mixed"):
            tok = ops.split_planes(self._bb.forward_tokens(img, flip=flip).reshape(B * P, D))
            hid = ops.gemm_x3(tok, self._w_hid, self._b_hid, _lib.EPI_RELU_BF16)
            code = ops.gemm_x3(tok, self._w_lin, self._b_lin, _lib.EPI_F32)
            ops.gemm_x3(hid, self._w_nl, self._b_nl, _lib.EPI_RESID_F32, out=code)
        else:
            tok = self._bb.forward_tokens(img, flip=flip).reshape(B * P, D)
            hid = ops.gemm_f32(tok, self._w_hid, self._b_hid, _lib.F32_RELU)
            code = ops.gemm_f32(tok, self._w_lin, self._b_lin, _lib.F32_NONE)
            ops.gemm_f32(hid, self._w_nl, self._b_nl, _lib.F32_RESID, out=code)
        return code.reshape(B, P, self._C)

    def _code_pair(self, img: torch.Tensor):
        """``_code_once`` for the frames and their mirrors together: (code [2B, P, C], chunk) in the layout of
        ``VitBackbone.forward_tokens_pair``."""
        B = img.shape[0]
        P, D = self._bb.grid ** 2, self._D
        if self._precision in ("bf16", "fp8", "fp16"):
            cat = torch.empty(2 * B * P, 2 * D, dtype=self._bb.lowp_dtype, device=self._device)  # [tok | hid]
            _, chunk = self._bb.forward_tokens_pair(img, lowp_out=cat)
            ops.gemm_bf16(cat[:, :D], self._w_hid, self._b_hid, _lib.EPI_RELU_BF16, out=cat[:, D:])
            code = ops.gemm_bf16(cat, self._w_code, self._b_code, _lib.EPI_F32)
        elif self._precision in ("exact", "mixed"):
            tok32, chunk = self._bb.forward_tokens_pair(img)
            tok = ops.split_planes(tok32.reshape(2 * B * P, D))
            hid = ops.gemm_x3(tok, self._w_hid, self._b_hid, _lib.EPI_RELU_BF16)
            code = ops.gemm_x3(tok, self._w_lin, self._b_lin, _lib.EPI_F32)
            ops.gemm_x3(hid, self._w_nl, self._b_nl, _lib.EPI_RESID_F32, out=code)
        else:
            tok32, chunk = self._bb.forward_tokens_pair(img)
            tok = tok32.reshape(2 * B * P, D)
            hid = ops.gemm_f32(tok, self._w_hid, self._b_hid, _lib.F32_RELU)
            code = ops.gemm_f32(tok, self._w_lin, self._b_lin, _lib.F32_NONE)
            ops.gemm_f32(hid, self._w_nl, self._b_nl, _lib.F32_RESID, out=code)
        return code.reshape(2 * B, P, self._C), chunk

    @torch.no_grad()
    def code_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] in [0,1] -> STEGO code [B, G*G, 90] fp32 (patch resolution)."""
        # T.Resize(NEAREST) + T.CenterCrop + T.Normalize (stego_interface.py:51-58, 87) happen inside the backbone's patch gather
        img = img.to(self._device)
        if not self._flip_tta:
            return self._code_once(img)
        # code averaged with the flipped-back code of the mirrored frame.  The mirror is a reversed column table of the fused ingest, and
        # both passes go through the network in ONE launch sequence per chunk (twice the rows per kernel launch)
        B, P = img.shape[0], self._bb.grid ** 2
        code2, chunk = self._code_pair(img)                      # [2B, P, C], chunk by chunk [frames | mirrors]
        out = torch.empty(B, P, self._C, dtype=torch.float32, device=self._device)
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            ops.flip_average(code2[2 * b0:2 * b0 + nb], code2[2 * b0 + nb:2 * b0 + 2 * nb], self._bb.grid, out=out[b0:b0 + nb])
        return out

    @torch.no_grad()
    def inference(self, img: torch.Tensor, code: torch.Tensor = None):
        """stego_interface.py:73-111: returns (linear_pred, cluster_pred), both [1,B,H,H] int32 (linear_pred is None when no
        linear probe was loaded), and keeps ``features`` = code [B,90,H,H] (bilinear, align_corners=True).  ``code``: the
        result of ``code_tokens(img)`` if the caller already ran that stage (e.g. on another stream)."""
        G = self._bb.grid
        H = img.shape[2]
        S = self._cfg.input_size   # postprocess() works on the resized frame (stego_interface.py:87-100); the label maps are then
        #                            nearest-resampled to the camera height (stego_interface.py:108-109)
        if code is None:
            code = self.code_tokens(img)
        B = code.shape[0]
        self._code_tokens = code
        self._code = None  # dense code is produced lazily (features property): 72 MB/frame at 448^2
        self._H = H
        self._labels_patch = None
        if self._cluster_resolution == "pixel":      # cluster the S x S up-sampled code pixels
            K = self._cfg.n_image_clusters
            if self._cfg.run_clustering:
                # the dense code is never built: per pass a similarity table interpolated per pixel and summed tap weights times the
                # patch codes (kmeans_form="linear", csrc/stego_linear.hip), or every row re-created from the patch codes and
                # multiplied out ("direct", csrc/stego.hip) -- the same clustering, fixed summation orders in both
                form = self._kmeans_form
                if form == "linear" and not ops.kmeans_pixels_linear_supported(G, S, self._C, K):
                    if not self._code_ac:
                        raise _lib.WvnError(f"StegoInterface(code_align_corners=False): the linear form has no instantiation for C={self._C}, K={K}")
                    form = "direct"
                if form == "direct" and not ops.kmeans_cosine_pixels_supported(G, S, self._C, K):
                    form = "dense"
                if form == "dense":
                    # shapes outside the fused kernels' instantiations (e.g. a 64-d head) keep working through the dense rows:
                    # the up-sampled code is materialised (H * H * C floats per frame) and clustered by the patch-resolution kernel
                    # (code dimension 16 / 64 / 90); only a shape that kernel has no instantiation for either is an error
                    if self._C not in (16, 64, 90):
                        raise _lib.WvnError(f"StegoInterface: no k-means kernel for code dimension C={self._C} (fused pixel-resolution forms: 16 or 90 with "
                                            "up to 32 / 64 clusters; dense rows and cluster_resolution='patch': 16, 64 or 90)")
                    pix = ops.upsample_bilinear(code, G, S).permute(0, 2, 3, 1).reshape(B, S * S, self._C).contiguous()
                    try:
                        labels, self._n_segments = ops.kmeans_cosine(pix, K, KMEANS_ITERS, relabel=True)
                    except _lib.WvnError as e:
                        raise _lib.WvnError(f"StegoInterface: the dense-row k-means has no instantiation for C={self._C}, K={K} ({e})") from None
                else:
                    labels, self._n_segments = ops.kmeans_cosine_pixels(code, G, S, K, KMEANS_ITERS, relabel=True, form=form, align_corners=self._code_ac)
            elif self._clusters.shape[0] <= 32:
                # the cluster probe at pixel resolution: cosine similarity is linear in the (un-normalised) code up to the pixel's
                # positive norm, so the argmax over the interpolated patch similarities IS the argmax on the interpolated code
                sim = ops.gemm_f32(code.reshape(B * G * G, self._C), self._clusters, None)
                labels = ops.table_bilerp_argmax(sim.reshape(B, G * G, -1), G, S, align_corners=self._code_ac)
                self._n_segments = None
            else:
                pix = ops.upsample_bilinear(code, G, S).permute(0, 2, 3, 1).reshape(B * S * S, self._C)   # [B, C, S, S] -> pixel rows
                labels = self._probe_labels(pix, self._clusters, None, cosine=True)
                self._n_segments = None
            labels = labels.reshape(B, S, S)
            self._cluster_pred = (labels if H == S else ops.upsample_nearest_labels(labels, H))[None]
        else:
            if self._cfg.run_clustering:
                labels, self._n_segments = ops.kmeans_cosine(code, self._cfg.n_image_clusters, KMEANS_ITERS, relabel=True)
            else:
                labels = self._probe_labels(code.reshape(B * G * G, self._C), self._clusters, None, cosine=True)
                self._n_segments = None
            self._labels_patch = labels.reshape(B, G, G)  # patch-resolution ids (before the nearest up-sampling)
            self._cluster_pred = ops.upsample_nearest_labels(self._labels_patch, H)[None]
        if self._w_probe is not None:
            # ONE reading for both probes: with cluster_resolution="pixel" the linear probe too acts on the up-sampled
            # code (stego_interface.py:94-100: postprocess works on the interpolated code) -- its logits are linear in the code and
            # the interpolation weights sum to one, so the patch logits are interpolated and the argmax taken per pixel; with
            # "patch" both probes label patches and the maps are nearest-upsampled
            rows = code.reshape(B * G * G, self._C)
            if self._cluster_resolution == "pixel" and self._w_probe.shape[0] <= 32:
                logits = ops.gemm_f32(rows if rows.is_contiguous() else rows.contiguous(), self._w_probe, self._b_probe)
                lin = ops.table_bilerp_argmax(logits.reshape(B, G * G, -1), G, S, align_corners=self._code_ac)
                self._linear_pred = (lin if H == S else ops.upsample_nearest_labels(lin, H))[None]
            else:
                lin = self._probe_labels(rows, self._w_probe, self._b_probe, cosine=False)
                self._linear_pred = ops.upsample_nearest_labels(lin.reshape(B, G, G), H)[None]
        else:
            self._linear_pred = None
        return self._linear_pred, self._cluster_pred

    def _probe_labels(self, rows: torch.Tensor, w: torch.Tensor, b, cosine: bool) -> torch.Tensor:
        x = ops.normalize_rows(rows) if cosine else (rows if rows.is_contiguous() else rows.contiguous())
        return ops.argmax_rows(ops.gemm_f32(x, w, b))

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
