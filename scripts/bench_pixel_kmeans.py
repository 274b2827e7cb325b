#!/usr/bin/env python
"""Time the pixel-resolution k-means on its own (GPU box): 64 frames of 56 x 56 x 90 code -> 448 x 448 labels, K = 20, 10 iterations.
    python scripts/bench_pixel_kmeans.py [frames]"""
import sys

import torch

sys.path.insert(0, ".")
from wild_visual_navigation_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
code = (torch.randn(B, 56 * 56, 90, generator=torch.Generator().manual_seed(0)) * 2 + 0.3).to(dev)
for _ in range(2):
    ops.kmeans_cosine_pixels(code, 56, 448, 20)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    ops.kmeans_cosine_pixels(code, 56, 448, 20)
b.record()
torch.cuda.synchronize()
print(f"pixel k-means, {B} frames 448^2, K = 20, 10 iterations: {a.elapsed_time(b) / 5:.2f} ms per call")
