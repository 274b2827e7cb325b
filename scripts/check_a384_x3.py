#!/usr/bin/env python
"""GPU check of the A-stationary split-operand K = 384 kernel (gemm_a384_x3.hip) inside the ViT: 'mixed' and 'exact' tokens with the
kernel on / off (WVN_VIT_NO_A384_X3) against each other and against the CPU oracle."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import interfaces as OI, vit as OV  # noqa: E402
from wild_visual_navigation_amd.backbone import VitBackbone  # noqa: E402
dev = torch.device("cuda:0")
S, B, depth = 448, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 3
sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=0, depth=depth)
img = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(1))
want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
for prec in ("mixed", "exact"):
    os.environ.pop("WVN_NO_A384_X3", None)
    a = VitBackbone(sd, S, 8, 6, device=dev, precision=prec, max_chunk=B).forward_tokens(img.to(dev)).cpu()
    os.environ["WVN_NO_A384_X3"] = "1"
    b = VitBackbone(sd, S, 8, 6, device=dev, precision=prec, max_chunk=B).forward_tokens(img.to(dev)).cpu()
    print(f"{prec}: a384_x3 vs oracle {(a - want).abs().max():.3e}; tiled vs oracle {(b - want).abs().max():.3e}; a384_x3 vs tiled {(a - b).abs().max():.3e}; finite {bool(torch.isfinite(a).all())}", flush=True)
