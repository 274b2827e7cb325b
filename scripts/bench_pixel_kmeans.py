#!/usr/bin/env python
"""Time the pixel-resolution k-means on its own (GPU box): 64 frames of 56 x 56 x 90 code -> 448 x 448 labels, K = 20, 10 iterations,
with the assignment kernels (plain / packed VALU, screened bf16 MFMA, the screened kernel's exact path: `wvn_debug_kmeans_assign_form`), and check that they agree bit for bit.
    python scripts/bench_pixel_kmeans.py [frames]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from wild_visual_navigation_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
# smooth code maps (low-pass noise) so that labels are spatially coherent like a real segmentation's
g = torch.Generator().manual_seed(0)
code = torch.randn(B, 90, 14, 14, generator=g)
code = torch.nn.functional.interpolate(code, (56, 56), mode="bicubic").permute(0, 2, 3, 1).reshape(B, 56 * 56, 90).contiguous()
code = (code * 2 + 0.3).to(dev)
res = {}
for form, name in ((0, "valu"), (4, "valu, packed dot products"), (5, "valu, packed dot products and interpolation (default)"), (1, "mfma"), (2, "mfma kernel, every row exact")):
    _lib.lib().wvn_debug_kmeans_assign_form(form)
    for _ in range(2):
        out = ops.kmeans_cosine_pixels(code, 56, 448, 20, return_centroids=True, form="direct")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.kmeans_cosine_pixels(code, 56, 448, 20, form="direct")
    b.record()
    torch.cuda.synchronize()
    res[name] = out
    if form == 1:
        import ctypes
        st = (ctypes.c_ulonglong * 2)()
        _lib.lib().wvn_debug_kmeans_screen_stats(st, 1)
        _lib.lib().wvn_debug_kmeans_assign_form(3)
        ops.kmeans_cosine_pixels(code, 56, 448, 20, form="direct")
        torch.cuda.synchronize()
        _lib.lib().wvn_debug_kmeans_assign_form(form)
        _lib.lib().wvn_debug_kmeans_screen_stats(st, 1)
        print(f"screened kernel: {st[0]} of {st[1]} 64-pixel row groups re-done exactly ({100.0 * st[0] / max(st[1], 1):.3f} %)", flush=True)
    print(f"pixel k-means [{name} assign], {B} frames 448^2, K = 20, 10 iterations: {a.elapsed_time(b) / 5:.2f} ms per call", flush=True)
_lib.lib().wvn_debug_kmeans_assign_form(-1)
# the linear form (csrc/stego_linear.hip), with the rows-per-chunk knob of its assign kernel
for rc in (5, 4, 3, 8, 10):
    _lib.lib().wvn_debug_kmeans_linear_rows(rc)
    for _ in range(2):
        lin = ops.kmeans_cosine_pixels(code, 56, 448, 20, return_centroids=True, form="linear")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.kmeans_cosine_pixels(code, 56, 448, 20, form="linear")
    b.record()
    torch.cuda.synchronize()
    print(f"pixel k-means [LINEAR form, {rc} rows per chunk], {B} frames 448^2, K = 20, 10 iterations: {a.elapsed_time(b) / 5:.2f} ms per call; "
          f"labels equal to the direct form's: {(lin[0] == res['valu'][0]).float().mean().item():.6f}", flush=True)
_lib.lib().wvn_debug_kmeans_linear_rows(0)
same = all(torch.equal(res[k][0], res["valu"][0]) and torch.equal(res[k][2], res["valu"][2]) for k in res)
print("labels and centroids identical between the two forms:", same)
sys.exit(0 if same else 1)
