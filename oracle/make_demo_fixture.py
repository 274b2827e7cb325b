"""Build tests/golden/demo_frames_224.pt (and demo_frames_raw.pt: the undecimated frames) from the reference's own demo images.

Run in the build container only (needs /root/reference):   python -m oracle.make_demo_fixture

It decodes assets/demo_data/*.png exactly as quick_start.py:156-161 does (PIL -> RGB -> uint8 tensor) and applies
ImageProjector.resize_image (image_projector.py:56-59,199-200: T.Resize(224, NEAREST) + T.CenterCrop(224)) through
the oracle's restatement of those transforms.  The frames are stored as uint8 [4,3,224,224] (600 KB) so the GPU box,
which has no /root/reference, can run the quick_start sequence on the reference's real inputs.
"""
import glob
import os

import numpy as np
import torch
from PIL import Image

from . import interfaces as OI

REF = "/root/reference/assets/demo_data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_frames_224.pt")


def main():
    names, frames, raw = [], [], []
    for p in sorted(glob.glob(os.path.join(REF, "*.png"))):
        img = torch.from_numpy(np.array(Image.open(p).convert("RGB"))).permute(2, 0, 1)[None].float()
        raw.append(img[0].to(torch.uint8))
        img = OI.resize_nearest_center_crop(img, 224)  # NEAREST: values stay integers in [0,255]
        assert img.shape == (1, 3, 224, 224) and torch.equal(img, img.round())
        frames.append(img[0].to(torch.uint8))
        names.append(os.path.basename(p))
    torch.save({"names": names, "frames_u8": torch.stack(frames),
                "source": "leggedrobotics/wild_visual_navigation assets/demo_data, quick_start.py:156-174 preprocessing"}, OUT)
    print("wrote", OUT, names, torch.stack(frames).shape)
    # the same frames UNRESIZED (224 x 299, what the camera / PNG decoder hands over): inputs of the fused-ingest parity tests
    # (NEAREST resize + centre crop inside the patch gather, tests/test_gpu_ingest.py)
    out_raw = os.path.join(os.path.dirname(OUT), "demo_frames_raw.pt")
    torch.save({"names": names, "frames_u8": torch.stack(raw),
                "source": "leggedrobotics/wild_visual_navigation assets/demo_data, decoded as quick_start.py:156-161 (no resize)"}, out_raw)
    print("wrote", out_raw, torch.stack(raw).shape)
    # the one real 448 x 448 frame the reference ships (assets/graph/img.png, the visualiser demo's input): BASELINE's full size
    p448 = "/root/reference/assets/graph/img.png"
    img = torch.from_numpy(np.array(Image.open(p448).convert("RGB"))).permute(2, 0, 1).contiguous()
    assert img.shape == (3, 448, 448) and img.dtype == torch.uint8
    out448 = os.path.join(os.path.dirname(OUT), "graph_img_448.pt")
    torch.save({"name": "assets/graph/img.png", "frame_u8": img}, out448)
    print("wrote", out448, tuple(img.shape))


if __name__ == "__main__":
    main()
