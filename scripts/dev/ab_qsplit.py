import os, sys
# A/B of the two-plane-q forms inside the model: tokens vs oracle on the reference's real frame + random frames
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import interfaces as OI, vit as OV
from wild_visual_navigation_amd import _lib
from wild_visual_navigation_amd.backbone import VitBackbone
dev = torch.device("cuda:0")
sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=0)
u8 = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "graph_img_448.pt"))["frame_u8"]
img = torch.cat([(u8.float() / 255)[None], torch.rand(3, 3, 448, 448, generator=torch.Generator().manual_seed(1))])
want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
for form in (2, 1):
    _lib.lib().wvn_debug_attention_variant(16 + form)
    for nq in (6, 12):
        bb = VitBackbone(sd, 448, 8, 6, device=dev, precision="mixed", max_chunk=4, qsplit_blocks=nq)
        got = bb.forward_tokens(img.to(dev)).cpu()
        e = (got - want).abs()
        print(f"q_lo form {form} ({'scaled e5m2 MFMA' if form == 2 else 'fp16 MFMAs'}), q split in {nq} blocks: tokens max err real frame {e[0].max().item():.2e}, random frames {e[1:].max().item():.2e}", flush=True)
_lib.lib().wvn_debug_attention_variant(18)
