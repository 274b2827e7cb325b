"""Patch-aligned fused pooling (wvn_segpool_patch_labels) must equal the reference semantics -- mean over
the segment's pixels of the bilinearly (align_corners=True) up-sampled map -- whenever the segment map is
a patch-resolution label grid nearest-upsampled by the patch size."""
import pytest
import torch

from oracle import interfaces as OI, segments as OS
from wild_visual_navigation_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,P,S,D", [(8, 8, 5, 24), (28, 8, 20, 90), (56, 8, 20, 90), (56, 8, 196, 384), (14, 16, 9, 70)])
def test_patch_aligned_pooling_matches_pixel_mean(dev, G, P, S, D):
    B, H = 2, G * P
    g = torch.Generator().manual_seed(G * 100 + S)
    tok = torch.randn(B, G * G, D, generator=g)
    if S == 196:  # 32-pixel grid cells (the reference's `grid` segmentation) expressed at patch resolution
        cell = 32 // P
        gy = torch.arange(G) // cell
        lab = (gy[:, None] * (G // cell) + gy[None, :]).expand(B, G, G).contiguous()
    else:
        lab = torch.randint(0, S, (B, G, G), generator=g)
        lab[1][lab[1] == 2] = 0  # an id that never occurs -> NaN row
    seg = OI.upsample_nearest(lab.int(), H)[0].long()  # [B,H,H]
    dense = OI.upsample_bilinear_ac(tok.reshape(B, G, G, D).permute(0, 3, 1, 2), H)

    def pooled(b):
        sp = OS.sparsify_features(dense[b:b + 1], seg[b])
        return torch.cat([sp, torch.full((S - sp.shape[0], D), float("nan"))])

    want = torch.stack([pooled(b) for b in range(B)])
    got = ops.segpool_patch_labels(lab.to(dev), tok.to(dev), G, H, S)
    assert got is not None
    got = got.cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.allclose(got, want, atol=2e-5, equal_nan=True), (got - want).nan_to_num().abs().max()
    general = ops.segpool_bilinear_mean(seg.to(dev), tok.to(dev), G, S).cpu()
    assert torch.allclose(got, general, atol=2e-5, equal_nan=True)
    # deterministic: two runs are bit-identical (no atomics)
    again = ops.segpool_patch_labels(lab.to(dev), tok.to(dev), G, H, S).cpu()
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(again))


def test_non_aligned_geometry_falls_back(dev):
    assert ops.patch_stencil_tables(14, 100, dev) is None  # 100 is not a multiple of 14
    tab = ops.patch_stencil_tables(56, 448, dev)
    assert tab.shape == (56, 3) and torch.allclose(tab.sum(1).cpu(), torch.ones(56), atol=1e-5)
