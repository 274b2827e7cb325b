#!/usr/bin/env python
"""Localise a disagreement between the pixel-resolution k-means kernels, the materialised route on the GPU and the CPU oracle
(GPU box):   python scripts/debug_pixel_kmeans.py [G H K]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import interfaces as OI  # noqa: E402
from wild_visual_navigation_amd import ops  # noqa: E402

real = "--real" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
G, H, K = (int(a) for a in argv[:3]) if len(argv) >= 3 else (56, 448, 20)
C = 90
dev = torch.device("cuda:0")
if real:   # the code of tests/test_gpu_stego_pixels.py::test_pixel_kmeans_at_448_against_oracle
    from oracle import vit as OV
    from wild_visual_navigation_amd.feature_extractor import StegoInterface

    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=2, depth=1)
    head = OI.make_stego_head_state_dict(384, 90, seed=2)
    img = torch.rand(1, 3, H, H, generator=torch.Generator().manual_seed(3))
    si = StegoInterface(dev, input_size=H, n_image_clusters=K, run_crf=False, run_clustering=True, backbone_weights=sd, head_weights=head,
                        precision="fp16", flip_tta=True, cluster_resolution="patch", allow_synthetic=True)
    code = si.code_tokens(img.to(dev)).cpu()
    print("real code: abs max", float(code.abs().max()), "row norm min/max", float(code.norm(dim=-1).min()), float(code.norm(dim=-1).max()))
else:
    code = torch.randn(1, G * G, C, generator=torch.Generator().manual_seed(3)) * (1.0 + torch.rand(1, G * G, 1, generator=torch.Generator().manual_seed(4)))
dense_gpu = ops.upsample_bilinear(code.to(dev), G, H).permute(0, 2, 3, 1).reshape(1, H * H, C).contiguous()
t = time.time()
dense_cpu = OI.upsample_bilinear_fixed(code[0].reshape(G, G, C).numpy(), H).reshape(H * H, C)
print(f"up-sampling: GPU vs fixed-order oracle: {(dense_gpu[0].cpu().numpy() != dense_cpu).sum()} of {dense_cpu.size} values differ ({time.time() - t:.1f} s)")
xn_gpu = ops.normalize_rows(dense_gpu[0])
xn_cpu = OI._normalize_rows_f32(dense_cpu)
print(f"normalised rows: {(xn_gpu.cpu().numpy() != xn_cpu).sum()} values differ")
for iters in (0, 1, 2, 10):
    fused, _ = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=iters, relabel=False)
    mat, _ = ops.kmeans_cosine(dense_gpu, K, iters=iters, relabel=False)
    t = time.time()
    want = OI.kmeans_cosine_labels(dense_cpu, K, iters=iters)
    f, m = fused[0].cpu().numpy(), mat[0].cpu().numpy()
    print(f"iters {iters:2d}: fused vs materialised {int((f != m).sum())}, fused vs oracle {int((f != want).sum())}, materialised vs oracle "
          f"{int((m != want).sum())} of {want.size} labels differ ({time.time() - t:.1f} s oracle)", flush=True)
    print(f"          cluster sizes (oracle): {np.bincount(want, minlength=K).tolist()}")
    if (f != want).any():
        bad = np.nonzero(f != want)[0]
        print(f"          first differing pixels {bad[:8].tolist()} (y, x of the first: {divmod(int(bad[0]), H)}); gpu {f[bad[:8]].tolist()} oracle {want[bad[:8]].tolist()}")
