#!/usr/bin/env python
"""K = 1: every group partial of the fused pixel k-means is the ordered sum of 512 consecutive normalised rows.  Compare with
the same sums formed on the CPU from the materialised rows (GPU box)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from wild_visual_navigation_amd import ops  # noqa: E402
from wild_visual_navigation_amd._lib import check, lib, ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
for G, H, Cc in ((8, 64, 90), (28, 224, 90), (56, 448, 90), (56, 448, 16)):
    P, K = H * H, 1
    code = (torch.randn(1, G * G, Cc, generator=torch.Generator().manual_seed(3)) * 2 + 0.5).to(dev)
    dense = ops.upsample_bilinear(code, G, H).permute(0, 2, 3, 1).reshape(1, P, Cc).contiguous()
    x = ops.normalize_rows(dense[0]).cpu().numpy()
    labels = torch.empty(1, P, dtype=torch.int32, device=dev)
    nseg = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(lib().wvn_kmeans_pixels_scratch_bytes(1, G, H, Cc, K) // 4, dtype=torch.float32, device=dev)
    check(lib().wvn_kmeans_cosine_pixels(ptr(code), ptr(labels), ptr(nseg), ptr(scratch), 1, G, H, Cc, K, 1, 0, stream()))
    ngroup = (P + 511) // 512
    cw = (K * Cc + 63) // 64 * 64
    part = scratch[cw: cw + ngroup * K * Cc].cpu().numpy().reshape(ngroup, Cc)
    pw = (ngroup * K * Cc + 63) // 64 * 64
    cntw = (ngroup * K + 63) // 64 * 64
    rinv = scratch[cw + pw + cntw: cw + pw + cntw + P].cpu().numpy()
    n2 = np.zeros(P, dtype=np.float32)
    dn = dense[0].cpu().numpy()
    for d in range(Cc):
        n2 = (n2 + (dn[:, d] * dn[:, d]).astype(np.float32)).astype(np.float32)
    rinv_cpu = (np.float32(1) / np.maximum(np.sqrt(n2).astype(np.float32), np.float32(1e-12))).astype(np.float32)
    want = np.zeros((ngroup, Cc), dtype=np.float32)
    for g in range(ngroup):
        grp = np.zeros(Cc, dtype=np.float32)
        for c in range(8):
            t = np.zeros(Cc, dtype=np.float32)
            for p in range(g * 512 + c * 64, min(P, g * 512 + c * 64 + 64)):
                t = (t + x[p]).astype(np.float32)
            grp = (grp + t).astype(np.float32)
        want[g] = grp
    bad = np.nonzero((part != want).any(1))[0]
    print(f"G={G} H={H} C={Cc}: rinv values that differ from the CPU's {int((rinv != rinv_cpu).sum())} of {P}; group partials that differ "
          f"{len(bad)} of {ngroup}", flush=True)
    if len(bad):
        g = int(bad[0])
        dd = np.nonzero(part[g] != want[g])[0]
        print(f"   group {g}: channels {dd[:10].tolist()}  gpu {part[g][dd[:4]]}  cpu {want[g][dd[:4]]}")
