"""How many Lloyd iterations until the pixel-resolution k-means of bench-like code stops changing labels (GPU box)?"""
import sys
import torch
sys.path.insert(0, ".")
from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd.feature_extractor import FeatureExtractor

dev = torch.device("cuda:0")
fe = FeatureExtractor(dev, segmentation_type="stego", feature_type="stego", input_size=448, n_image_clusters=20, precision="fp16",
                      allow_synthetic=True, max_chunk=8)
img = torch.rand(8, 3, 448, 448, generator=torch.Generator().manual_seed(1)).to(dev)
code = fe._extractor.code_tokens(img)
prev = None
for it in range(0, 21):
    lab, _ = ops.kmeans_cosine_pixels(code, 56, 448, 20, it, relabel=False)
    if prev is not None:
        ch = (lab != prev).reshape(8, -1).float().mean(1)
        print(it, " ".join(f"{c:.5f}" for c in ch.tolist()))
    prev = lab.clone()
