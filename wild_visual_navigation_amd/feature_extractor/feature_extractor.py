"""FeatureExtractor -- same public surface as
wild_visual_navigation/feature_extractor/feature_extractor.py:19-398 (constructor, ``extract`` 5-tuple,
``compute_segments`` / ``compute_features`` / ``sparsify_features``, properties), re-designed for
MI355X: the ViT runs on MFMA kernels, and the per-segment features come from ONE fused
up-sample + segment-mean kernel pair that reads the 4.8 MB patch map instead of writing and
re-reading the 308 MB dense map (dino_interface.py:87-90 -> feature_extractor.py:390-396).
``extract_batch`` is the batched (B > 1) entry point with per-image semantics identical to B = 1.
"""
from typing import Optional, Tuple

import torch

from .. import _lib, ops
from .dino_interface import DinoInterface
from .segment_extractor import SegmentExtractor
from .stego_interface import StegoInterface


class FeatureExtractor:
    def __init__(self, device: str, segmentation_type: str = "slic", feature_type: str = "dino",
                 input_size: int = 448, **kwargs):
        self._device = torch.device(device)
        self._segmentation_type = segmentation_type
        self._feature_type = feature_type
        self._input_size = input_size
        self._stego_features_already_computed_in_segmentation = False
        self._tokens = None  # patch-resolution features of the frame(s) being processed
        self.segment_extractor = SegmentExtractor().to(self._device)
        precision = kwargs.get("precision", "mixed")   # the <= 1e-3 mode (the reference is fp32 end to end); "fp16" / "bf16" are opt-in speed paths

        if self._feature_type == "stego":
            self._feature_dim = 90
            self._extractor = StegoInterface(
                device=device, input_size=input_size,
                n_image_clusters=kwargs.get("n_image_clusters", 20),
                run_clustering=kwargs.get("run_clustering", True),
                run_crf=kwargs.get("run_crf", False),
                backbone_type=kwargs.get("backbone_type", "vit_small" if kwargs.get("model_path") is None else None),
                patch_size=kwargs.get("patch_size", 8), precision=precision,
                backbone_weights=kwargs.get("pretrained_weights"), head_weights=kwargs.get("head_weights"),
                probe_weights=kwargs.get("probe_weights"), model_path=kwargs.get("model_path"),
                max_chunk=kwargs.get("max_chunk", 16), flip_tta=kwargs.get("flip_tta", True),
                cluster_resolution=kwargs.get("cluster_resolution", "pixel"), kmeans_form=kwargs.get("kmeans_form", "linear"),
                allow_synthetic=kwargs.get("allow_synthetic", False), fuse_mlp=kwargs.get("fuse_mlp"), fuse_qkv=kwargs.get("fuse_qkv"), fuse_proj=kwargs.get("fuse_proj", True),
                pos_embed_rule=kwargs.get("pos_embed_rule", "dino"), code_align_corners=kwargs.get("code_align_corners", True),
            )
        elif "dino" in self._feature_type:
            self._feature_dim = 384  # the reference hard-codes 384 for any dino type (feature_extractor.py:56)
            self._extractor = DinoInterface(
                device=device, input_size=input_size, patch_size=kwargs.get("patch_size", 8),
                backbone=kwargs.get("backbone", "dinov2" if self._feature_type == "dinov2" else "dino"),
                backbone_type=kwargs.get("backbone_type", "vit_small"),
                pretrained_weights=kwargs.get("pretrained_weights"), precision=precision,
                max_chunk=kwargs.get("max_chunk", 16), allow_synthetic=kwargs.get("allow_synthetic", False),
                fuse_mlp=kwargs.get("fuse_mlp"), fuse_qkv=kwargs.get("fuse_qkv"), fuse_proj=kwargs.get("fuse_proj", True),
                pos_embed_rule=kwargs.get("pos_embed_rule", "dino"),
            )
            self._feature_dim = self._extractor.feature_dim
        elif self._feature_type == "none":
            self._extractor = None
        else:
            # sift / torchvision / histogram ablation extractors are outside the MI355X hot path
            raise TypeError(f"Extractor[{self._feature_type}] not supported!")

        if self.segmentation_type == "slic":
            # The reference round-trips through the CPU package fast_slic (feature_extractor.py:84-90, 221-225); here SLIC runs on
            # the GPU in integer arithmetic (csrc/slic.hip; parity with fast_slic's own variant is unpinned)
            self._slic_num_components = kwargs.get("slic_num_components", 100)
            self._slic_compactness = kwargs.get("slic_compactness", 10)
        elif self.segmentation_type == "stego" and self._feature_type != "stego":
            raise TypeError("segmentation_type 'stego' requires feature_type 'stego' (as in the reference)")

    # ------------------------------------------------------------------------------------------ props
    @property
    def feature_type(self):
        return self._feature_type

    @property
    def feature_dim(self):
        return self._feature_dim

    @property
    def segmentation_type(self):
        return self._segmentation_type

    def change_device(self, device):
        """feature_extractor.py:130-139: move every member to ``device`` (another GPU)."""
        self._device = torch.device(device)
        self.segment_extractor = self.segment_extractor.to(self._device)
        if self._extractor is not None:
            self._extractor.change_device(device)

    # ------------------------------------------------------------------------------------------ extract
    @torch.no_grad()
    def extract(self, img: torch.Tensor, **kwargs):
        """feature_extractor.py:95-128.  img [1,3,H,W] fp32 in [0,1], or the raw uint8 frame (then x/255 is fused into
        the backbone's patch gather: same features, a quarter of the upload).
        Returns (edges [2,E] i64, feat [S,D] f32, seg [H,W] i64, center [S,2] f32, dense [1,D,H,H] | None)."""
        img = img.to(self._device)
        want_dense = kwargs.get("return_dense_features", False)
        if self._segmentation_type == "random":
            tokens = self._feature_tokens(img)
            H, W = img.shape[2:]
            nr = kwargs.get("n_random_pixels", 100)
            seg = torch.full((H * W,), -1, dtype=torch.long, device=self._device)
            indices = torch.randperm(H * W, device=self._device)[:nr]
            seg[indices] = torch.arange(0, nr, device=self._device)
            seg = seg.reshape(H, W)
            # a one-pixel segment's "mean" is the interpolated feature at that pixel
            feat = ops.segpool_bilinear_mean(seg[None], tokens, self._grid(), nr)[0]
            dense = ops.upsample_bilinear(tokens, self._grid(), H) if want_dense else None
            return None, feat, seg, None, dense

        edges, seg, center = self.compute_segments(img, **kwargs)
        tokens = self._feature_tokens(img)
        n_seg = center.shape[0]
        feat = ops.segpool_bilinear_mean(seg[None], tokens, self._grid(), n_seg)[0]
        dense = ops.upsample_bilinear(tokens, self._grid(), img.shape[2]) if want_dense else None
        return edges, feat, seg, center, dense

    @torch.no_grad()
    def backbone_stage(self, img: torch.Tensor) -> torch.Tensor:
        """First stage of ``extract_batch`` on its own: the ViT (and, for stego features, the STEGO head) -> the
        patch-resolution feature tokens [B, G*G, D].  Everything after it (clustering, pooling, the MLP step) consists of
        small kernels that do not fill the GPU; a throughput pipeline runs this stage for batch i+1 on one HIP stream while
        the rest of batch i runs on another (``extract_batch(img, backbone_out=...)``, bench.py)."""
        img = img.to(self._device)
        if self._feature_type == "stego":
            return self._extractor.code_tokens(img)
        return self._extractor.inference_tokens(img)

    @torch.no_grad()
    def extract_batch(self, img: torch.Tensor, backbone_out: Optional[torch.Tensor] = None,
                      **kwargs) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Batched hot path (no graph structure): img [B,3,H,H] -> (feat [B,S,D], seg [B,H,H] int32,
        n_segments [B] int32).  Rows of ids that do not occur in an image are NaN (the reference's empty
        mean).  Supported segmentations: grid, stego.  ``backbone_out``: the tokens ``backbone_stage(img)`` returned."""
        img = img.to(self._device)
        B, _, H, W = img.shape
        G = self._grid()
        labels_patch = None  # [B,G,G] ids when the segment map is constant on every ViT patch
        if self._segmentation_type == "grid":
            cell = kwargs.get("cell_size", 32)
            seg = self.segment_grid(img, **kwargs)[0, 0].to(torch.int32)[None].expand(B, H, W).contiguous()
            n_seg = (H // cell) * (W // cell)
            nseg = torch.full((B,), n_seg, dtype=torch.int32, device=self._device)
            tokens = backbone_out if backbone_out is not None else self._feature_tokens(img)
            P = H // G
            if H == W and G * P == H and cell % P == 0:
                labels_patch = seg[:, ::P, ::P]
        elif self._segmentation_type == "stego":
            if backbone_out is not None and self._feature_type != "stego":
                raise _lib.WvnError("extract_batch: backbone_out with stego segmentation needs feature_type='stego'")
            self._extractor.inference(img, code=backbone_out)
            seg = self._extractor.cluster_segments[0]
            nseg = self._extractor._n_segments
            if nseg is None:
                raise _lib.WvnError("extract_batch with stego segmentation needs run_clustering=True (per-image k-means)")
            n_seg = self._extractor._cfg.n_image_clusters
            tokens = self._extractor.feature_tokens
            labels_patch = self._extractor._labels_patch
        elif self._segmentation_type == "slic":
            seg = torch.stack([ops.slic(img[b], self._slic_num_components, self._slic_compactness) for b in range(B)])
            n_seg = ops.slic_num_clusters(H, W, self._slic_num_components)
            nseg = torch.full((B,), n_seg, dtype=torch.int32, device=self._device)
            tokens = backbone_out if backbone_out is not None else self._feature_tokens(img)
        else:
            raise TypeError(f"extract_batch: segmentation_type [{self._segmentation_type}] not supported")
        feat = None
        if labels_patch is not None and kwargs.get("patch_aligned_pooling", True):
            feat = ops.segpool_patch_labels(labels_patch, tokens, G, H, n_seg)  # None if geometry does not allow
        if feat is None:
            feat = ops.segpool_bilinear_mean(seg, tokens, G, n_seg)
        return feat, seg, nseg

    # ------------------------------------------------------------------------------------------ pieces
    @torch.no_grad()
    def predict_per_pixel(self, img: torch.Tensor, model, confidence_generator=None, want_loss: bool = False):
        """The live node's per-frame prediction with ``prediction_per_pixel`` (wvn_feature_extractor_node.py:311-363;
        quick_start.py:176-210) in one call: img [B,3,H,W] (fp32 in [0,1] or uint8) -> (trav [B,H,H], conf [B,H,H],
        loss_reco | None).  Equivalent to ``extract(..., return_dense_features=True)`` -> ``model.forward(Data(x=dense
        rows))`` -> column 0 / ``confidence_generator.inference_without_update(mse(pred[:, 1:], x))``, but the dense
        [B,D,H,H] tensor is never built and layer 1 runs at patch resolution (csrc/pixel_mlp.hip).  DINO (384-d) or STEGO
        (90-d code) features; bf16 extractor -> bf16 MFMA kernel, fp32 (exact) extractor -> the hi + lo split form (<= 1e-3 of the reference).  Like the reference (dino_interface.py:87-90) the map is H x H for an H x W frame."""
        mean, std, f = 0.0, 1.0, 0.5
        if confidence_generator is not None:
            mean, std, f = float(confidence_generator.mean), float(confidence_generator.std), float(confidence_generator.std_factor)
        return self._predict_per_pixel(img.to(self._device), model, mean, std, f, None, want_loss)

    def _predict_per_pixel(self, img, model, mean, std, f, conf_state, want_loss):
        B, H = img.shape[0], img.shape[2]
        G = self._grid()
        stego = self._feature_type == "stego"
        prec = self._extractor._precision
        exact = prec in ("exact", "fp32", "mixed")
        if exact:   # fp32 extractor: hi + lo split MFMA operands in the fused kernel
            tokens = self.backbone_stage(img)
            return model.forward_per_pixel_exact(tokens.reshape(B * G * G, -1), B, G, (H, H), mean, std, f, want_loss=want_loss,
                                                 conf_state=conf_state)
        if stego:   # 90-d code (the live node's default feature_type), zero-padded to the 128 columns the layer-1 GEMM reads
            code = self._extractor.code_tokens(img).reshape(B * G * G, -1)
            zx = torch.zeros(B * G * G, model.ZX_COLS, dtype=torch.bfloat16, device=self._device)
            zx[:, model.X_COL: model.X_COL + code.shape[1]] = code
        else:
            zx = torch.empty(B * G * G, model.ZX_COLS, dtype=torch.bfloat16, device=self._device)
            if prec == "fp16":   # the per-pixel kernel takes bf16 features: hand it the fp32 tokens rounded once
                ops.cast_rows_bf16(self._extractor._model.forward_tokens(img).reshape(B * G * G, -1), zx[:, model.X_COL:])
            else:
                self._extractor._model.forward_tokens(img, lowp_out=zx[:, model.X_COL:])
        return model.forward_per_pixel(zx, B, G, (H, H), mean, std, f, want_loss=want_loss, conf_state=conf_state)

    def _grid(self) -> int:
        return self._extractor.grid

    def _feature_tokens(self, img: torch.Tensor) -> torch.Tensor:
        """Patch-resolution features [B, G*G, D] of the configured feature type."""
        if self._feature_type == "stego":
            if self._stego_features_already_computed_in_segmentation:
                self._stego_features_already_computed_in_segmentation = False
                return self._extractor.feature_tokens
            self._extractor.inference(img.clone())
            return self._extractor.feature_tokens
        return self._extractor.inference_tokens(img)

    def compute_segments(self, img: torch.Tensor, **kwargs):
        """feature_extractor.py:151-177 -> (edges [2,E], seg [H,W] i64, centers [S,2])."""
        if self._segmentation_type == "none" or self._segmentation_type is None:
            edges, seg, centers = self.segment_pixelwise(img, **kwargs)
            return edges.T, seg[0, 0], centers
        if self._segmentation_type == "grid":
            seg = self.segment_grid(img, **kwargs)
        elif self._segmentation_type == "slic":
            seg = self.segment_slic(img, **kwargs)
        elif self._segmentation_type == "stego":
            seg = self.segment_stego(img, **kwargs)
        elif self._segmentation_type == "random":
            seg = self.segment_random(img, **kwargs)
        else:
            raise TypeError(f"segmentation_type [{self._segmentation_type}] not supported")
        edges = self.segment_extractor.adjacency_list(seg)
        centers = self.segment_extractor.centers(seg)
        return edges.T, seg[0, 0], centers

    def segment_pixelwise(self, img, **kwargs):
        B, C, H, W = img.shape
        seg = torch.arange(0, H * W, 1, device=self._device).reshape(H, W)
        ys, xs = torch.meshgrid(torch.arange(H, device=self._device), torch.arange(W, device=self._device),
                                indexing="ij")
        centers = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=1)
        hor = torch.stack([seg[:, :-1].reshape(-1), seg[:, 1:].reshape(-1)], dim=1)
        ver = torch.stack([seg[:-1, :].reshape(-1), seg[1:, :].reshape(-1)], dim=1)
        return torch.cat([hor, ver], dim=0), seg[None, None], centers

    def segment_grid(self, img, **kwargs):
        """feature_extractor.py:198-219: id = row-major index of the cell_size cell (index arithmetic only)."""
        cell = kwargs.get("cell_size", 32)
        H, W = img.shape[2:]
        gy = torch.arange(H, device=self._device) // cell
        gx = torch.arange(W, device=self._device) // cell
        return (gy[:, None] * (W // cell) + gx[None, :])[None, None].to(torch.int64)

    def segment_slic(self, img, **kwargs):
        """feature_extractor.py:221-225 without the host round trip: [1,3,H,W] (float in [0,1], truncated to 8 bits like the
        reference's np.uint8(img * 255), or uint8) -> [1,1,H,W] int64 ids in [0, number of SLIC clusters)."""
        seg = ops.slic(img[0].to(self._device), self._slic_num_components, self._slic_compactness)
        return seg[None, None].to(torch.long)

    def segment_random(self, img, **kwargs):
        H, W = img.shape[2:]
        nr = kwargs.get("n_random_pixels", 100)
        seg = torch.full((H * W,), -1, dtype=torch.long, device=self._device)
        indices = torch.randperm(H * W, device=self._device)[:nr]
        seg[indices] = torch.arange(0, nr, device=self._device)
        return seg.reshape(H, W)[None, None]

    def segment_stego(self, img, **kwargs):
        """feature_extractor.py:237-249; the ascending relabel is done inside the k-means kernel."""
        self._extractor.inference(img.clone())
        seg = self._extractor.cluster_segments.to(torch.long)  # [1,1,H,H]
        if self._extractor._n_segments is None:  # cluster-probe ids are not compacted: relabel ascending (:245-246)
            _, inv = torch.unique(seg, return_inverse=True)
            seg = inv.reshape(seg.shape)
        self._stego_features_already_computed_in_segmentation = True
        return seg

    @torch.no_grad()
    def compute_features(self, img: torch.Tensor, seg: torch.Tensor, center: torch.Tensor, **kwargs):
        """feature_extractor.py:251-274: dense per-pixel features [B,D,H,H] (materialised on request only)."""
        if self._feature_type == "none":
            return None
        if self._feature_type == "stego":
            if self._stego_features_already_computed_in_segmentation:
                self._stego_features_already_computed_in_segmentation = False
                return self._extractor.features
            self._extractor.inference(img.clone())
            return self._extractor.features
        return self._extractor.inference(img.clone())

    @torch.no_grad()
    def sparsify_features(self, dense_features: torch.Tensor, seg: torch.Tensor, cumsum_trick=False):
        """feature_extractor.py:310-398 (default branch) on an explicit dense map: identity resampling
        through the same segment-mean kernels (grid == image size)."""
        if self._feature_type in ["histogram"] or self._segmentation_type in ["none"]:
            return dense_features
        _lib.require_cuda(dense_features, "dense_features")
        _, D, H, W = dense_features.shape
        if H != W:
            raise _lib.WvnError("sparsify_features expects the square [1,D,H,H] map DinoInterface produces")
        n_seg = int(seg.max().item()) + 1
        tokens = dense_features[0].permute(1, 2, 0).reshape(1, H * W, D).contiguous()
        return ops.segmean_tokens(seg[None], tokens, n_seg)[0]
