#!/usr/bin/env python
"""The linear form of the pixel k-means alone (GPU box; for `rocprofv3 --kernel-trace --stats`): 64 frames of 56 x 56 x 90 code ->
448 x 448 labels, K = 20, 10 iterations.   python scripts/bench_kmeans_linear.py [frames] [rows per chunk] [calls]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from wild_visual_navigation_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
code = torch.randn(B, 90, 14, 14, generator=g)
code = torch.nn.functional.interpolate(code, (56, 56), mode="bicubic").permute(0, 2, 3, 1).reshape(B, 56 * 56, 90).contiguous()
code = (code * 2 + 0.3).to(dev)
_lib.lib().wvn_debug_kmeans_linear_rows(rc)
for _ in range(2):
    ops.kmeans_cosine_pixels(code, 56, 448, 20, form="linear")
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(calls):
    ops.kmeans_cosine_pixels(code, 56, 448, 20, form="linear")
b.record()
torch.cuda.synchronize()
print(f"pixel k-means [linear form, rows per chunk {rc or 'default'}], {B} frames 448^2, K = 20, 10 iterations: {a.elapsed_time(b) / calls:.3f} ms per call", flush=True)
