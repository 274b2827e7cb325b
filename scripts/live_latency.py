"""Single-frame latency of the live node's per-frame path (wvn_feature_extractor_node.py:305-363 with prediction_per_pixel):
8-bit 448x448 frame already on the GPU -> DINO ViT-S/8 (bf16) -> fused per-pixel traversability + confidence maps.
One JSON line.  (Replaying the same ~115-launch sequence from a HIP graph was measured at every size -- 1.70 vs 1.73 ms at
448x448, 1.28 vs 1.31 ms at 224x224 with stego features, bit-identical results -- and is not kept: the launches already overlap
execution, the frame time is the chain of dependent small-grid kernels.)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_visual_navigation_amd.backbone import synthetic_vit_state_dict  # noqa: E402
from wild_visual_navigation_amd.cfg import ExperimentParams  # noqa: E402
from wild_visual_navigation_amd.feature_extractor import FeatureExtractor  # noqa: E402
from wild_visual_navigation_amd.model import get_model  # noqa: E402
from wild_visual_navigation_amd.utils import ConfidenceGenerator  # noqa: E402


def wall(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def run(dev, S, ftype, precision="bf16"):
    sd = synthetic_vit_state_dict(depth=12, pretrain_grid=28)
    fe = FeatureExtractor(device=dev, segmentation_type="stego" if ftype == "stego" else "grid", feature_type=ftype, patch_size=8,
                          backbone_type="vit_small", input_size=S, pretrained_weights=sd, precision=precision)
    params = ExperimentParams()
    params.model.simple_mlp_cfg.input_size = fe.feature_dim
    model = get_model(params.model).to(dev)
    model.eval()
    cg = ConfidenceGenerator(method="latest_measurement", std_factor=0.5).to(dev)
    cg.mean[0], cg.std[0] = 0.9, 0.25
    frame = torch.randint(0, 256, (1, 3, S, S), dtype=torch.uint8, device=dev)
    eager = wall(lambda: fe.predict_per_pixel(frame, model, cg), 50)
    # the segmentation + pooling half of the node's frame (extract): k-means / grid segments, pooled rows
    seg_ms = wall(lambda: fe.extract(frame), 30)
    return {"frame": f"{S}x{S} uint8", "features": ftype, "precision": precision, "predict_per_pixel_ms": round(eager, 3), "extract_ms": round(seg_ms, 3)}


def main():
    dev = torch.device("cuda:0")
    # BASELINE's 448x448 DINO configuration, and the node's own default (default.yaml: 224x224, feature_type stego)
    # (round 5: the class default is precision="mixed", the <= 1e-3 mode; the 16-bit speed paths beside it)
    out = []
    for prec in ("mixed", "fp16", "bf16"):
        out += [run(dev, 448, "dino", prec), run(dev, 224, "stego", prec), run(dev, 224, "dino", prec)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
