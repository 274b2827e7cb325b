#!/usr/bin/env python
"""Find the first Lloyd iteration at which the fused pixel k-means and the materialised route disagree on a centroid, then the
group partial, the chunk and the label sequence behind it (GPU box)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import interfaces as OI, vit as OV  # noqa: E402
from wild_visual_navigation_amd import ops  # noqa: E402
from wild_visual_navigation_amd._lib import check, lib, ptr, stream  # noqa: E402
from wild_visual_navigation_amd.feature_extractor import StegoInterface  # noqa: E402

G, H, K, Cc = 56, 448, 20, 90
dev = torch.device("cuda:0")
sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=2, depth=1)
head = OI.make_stego_head_state_dict(384, 90, seed=2)
img = torch.rand(1, 3, H, H, generator=torch.Generator().manual_seed(3))
si = StegoInterface(dev, input_size=H, n_image_clusters=K, run_crf=False, run_clustering=True, backbone_weights=sd, head_weights=head,
                    precision="fp16", flip_tta=True, cluster_resolution="patch", allow_synthetic=True)
code = si.code_tokens(img.to(dev)).contiguous()
dense = ops.upsample_bilinear(code, G, H).permute(0, 2, 3, 1).reshape(1, H * H, Cc).contiguous()
xn = ops.normalize_rows(dense[0]).reshape(1, H * H, Cc)
P = H * H


def fused(iters):
    labels = torch.empty(1, P, dtype=torch.int32, device=dev)
    nseg = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(lib().wvn_kmeans_pixels_scratch_bytes(1, G, H, Cc, K) // 4, dtype=torch.float32, device=dev)
    check(lib().wvn_kmeans_cosine_pixels(ptr(code), ptr(labels), ptr(nseg), ptr(scratch), 1, G, H, Cc, K, iters, 0, stream()))
    return labels[0].cpu().numpy(), scratch


def mat(iters):
    labels = torch.empty(1, P, dtype=torch.int32, device=dev)
    nseg = torch.empty(1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(lib().wvn_kmeans_scratch_bytes(1, P, Cc, K) // 4, dtype=torch.float32, device=dev)
    check(lib().wvn_kmeans_cosine(ptr(xn), ptr(labels), ptr(nseg), ptr(scratch), 1, P, Cc, K, iters, 0, stream()))
    return labels[0].cpu().numpy(), scratch


for it in range(1, 11):
    lf, sf = fused(it)
    lm, sm = mat(it)
    cf, cm = sf[: K * Cc].cpu().numpy().reshape(K, Cc), sm[: K * Cc].cpu().numpy().reshape(K, Cc)
    nd = int((cf != cm).sum())
    print(f"iters {it}: centroid values that differ {nd}, labels that differ {int((lf != lm).sum())}", flush=True)
    if nd:
        ks, ds = np.nonzero(cf != cm)
        print("   clusters", sorted(set(ks.tolist())), " example (k, d)", (int(ks[0]), int(ds[0])), cf[ks[0], ds[0]], cm[ks[0], ds[0]])
        # labels that went into this update = final labels of a run with it - 1 iterations (identical in both routes so far)
        lab_prev, _ = fused(it - 1)
        x = xn[0].cpu().numpy()
        ngroup = P // 512
        cent_words = (K * Cc + 63) // 64 * 64
        part = sf[cent_words: cent_words + ngroup * K * Cc].cpu().numpy().reshape(ngroup, K, Cc)
        k0 = int(ks[0])
        bad_groups = []
        for g in range(ngroup):
            grp = np.zeros(Cc, dtype=np.float32)
            for c in range(8):
                p0 = g * 512 + c * 64
                t = np.zeros(Cc, dtype=np.float32)
                for p in range(p0, p0 + 64):
                    if lab_prev[p] == k0:
                        t = (t + x[p]).astype(np.float32)
                grp = (grp + t).astype(np.float32)
            if not np.array_equal(grp, part[g, k0]):
                bad_groups.append(g)
        print(f"   cluster {k0}: groups whose partial differs from the ordered CPU sum: {bad_groups[:10]} ({len(bad_groups)} of {ngroup})")
        if bad_groups:
            g = bad_groups[0]
            for c in range(8):
                p0 = g * 512 + c * 64
                print(f"   group {g} chunk {c} labels:", lab_prev[p0:p0 + 64].tolist())
        break
