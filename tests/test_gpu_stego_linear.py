"""The pixel-resolution k-means through its linearity (csrc/stego_linear.hip; the definition is oracle/kmeans_linear.py): a per-pass
similarity table interpolated per pixel, centroid sums from summed tap weights times the patch codes.  Labels AND centroids are
bit-exact against that oracle on identical code (every summation order is fixed); against the direct form (csrc/stego.hip,
oracle/interfaces.py::kmeans_cosine_labels_pixels) the maps differ only where two similarities are closer than fp32 rounding."""
import numpy as np
import pytest
import torch

from oracle import interfaces as OI, kmeans_linear as KL, vit as OV
from wild_visual_navigation_amd import _lib, ops
from wild_visual_navigation_amd.feature_extractor import StegoInterface

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture
def band_rows(request):
    _lib.lib().wvn_debug_kmeans_linear_rows(request.param)
    yield request.param
    _lib.lib().wvn_debug_kmeans_linear_rows(0)   # (out of range: back to the default)


@pytest.mark.parametrize("band_rows", [5, 3, 16], indirect=True)
@pytest.mark.parametrize("G,H,C,K,B", [(8, 64, 90, 5, 2), (7, 50, 16, 4, 3), (28, 224, 90, 20, 1), (5, 33, 90, 6, 2), (28, 224, 90, 20, 16),
                                       (9, 70, 90, 17, 9), (9, 70, 90, 19, 3), (8, 64, 90, 16, 2), (6, 100, 16, 27, 2), (12, 9, 16, 3, 8),
                                       (37, 518, 90, 20, 1)])
def test_linear_pixel_kmeans_bit_exact(dev, band_rows, G, H, C, K, B):
    """Shapes of tests/test_gpu_stego_pixels.py plus: (6, 100): 20 rows per band (several chunks of rows per band at every chunk size),
    K = 27 in the 32-slot instantiation; (12, 9): FEWER pixels than patches per side (bands without rows, patch columns without
    pixels); (37, 518): the DINOv2 grid of BASELINE configs[4]; B = 16 / 8: the frame -> XCD mapping."""
    code = torch.randn(B, G * G, C, generator=g(G * H)) * (1.0 + torch.rand(B, G * G, 1, generator=g(1)))
    lab, nseg, cent = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=10, relabel=False, return_centroids=True, form="linear")
    lab2, nseg2 = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=10, relabel=True, form="linear")
    for b in range(B if H < 200 else min(B, 2)):
        want, wcent = KL.kmeans_pixels_linear(code[b].numpy(), G, H, K, iters=10)
        assert np.array_equal(lab[b].cpu().numpy(), want), f"frame {b}: {(lab[b].cpu().numpy() != want).mean()} of the labels differ"
        assert np.array_equal(cent[b].cpu().numpy(), wcent), f"frame {b}: centroids differ by {np.abs(cent[b].cpu().numpy() - wcent).max()}"
        assert np.array_equal(lab2[b].cpu().numpy(), OI.relabel_ascending(want))
        assert int(nseg[b]) == len(np.unique(want)) == int(nseg2[b])


@pytest.mark.parametrize("iters", [0, 1, 3])
def test_linear_pixel_kmeans_odd_iteration_counts(dev, iters):
    """The centroid buffers ping-pong between passes: the final centroids must land where include/wvn_hip.h promises for any count."""
    G, H, C, K = 9, 70, 90, 7
    code = torch.randn(2, G * G, C, generator=g(iters))
    lab, _, cent = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=iters, relabel=False, return_centroids=True, form="linear")
    for b in range(2):
        want, wcent = KL.kmeans_pixels_linear(code[b].numpy(), G, H, K, iters=iters)
        assert np.array_equal(lab[b].cpu().numpy(), want) and np.array_equal(cent[b].cpu().numpy(), wcent)


def test_linear_and_direct_forms_agree_up_to_rounding_ties(dev):
    """The two statements of the clustering on the same code: identical maps except where the direct form's two best similarities are
    within fp32 rounding of each other (smooth code with cluster structure: a real segmentation's situation)."""
    G, H, C, K, B = 28, 224, 90, 20, 4
    code = torch.randn(B, C, 7, 7, generator=g(0))
    code = torch.nn.functional.interpolate(code, (G, G), mode="bicubic").permute(0, 2, 3, 1).reshape(B, G * G, C).contiguous() * 2 + 0.3
    lin, _ = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, relabel=False, form="linear")
    dire, _ = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, relabel=False, form="direct")
    agree = (lin == dire).float().mean().item()
    assert agree >= 0.999, agree


def test_linear_pixel_kmeans_at_448_against_oracle(dev):
    """BASELINE size through the class: one 448^2 frame, 56 x 56 x 90 code, K = 20 -- the StegoInterface default route."""
    G, H, C, K = 56, 448, 90, 20
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=2, depth=1)
    head = OI.make_stego_head_state_dict(384, 90, seed=2)
    img = torch.rand(1, 3, H, H, generator=g(3))
    si = StegoInterface(dev, input_size=H, n_image_clusters=K, run_crf=False, run_clustering=True, backbone_weights=sd, head_weights=head,
                        precision="fp16", flip_tta=True, cluster_resolution="pixel", allow_synthetic=True)
    _, clu = si.inference(img.to(dev))
    code = si.feature_tokens.cpu()                                                      # the GPU's own code: identical input
    want = OI.relabel_ascending(KL.kmeans_cosine_labels_pixels_linear(code[0].numpy(), G, H, K))
    assert clu.shape == (1, 1, H, H) and clu.dtype == torch.int32
    assert np.array_equal(clu[0, 0].cpu().numpy().reshape(-1), want)


@pytest.mark.parametrize("G,H,C,K,B", [(8, 64, 90, 5, 2), (7, 50, 16, 4, 3), (28, 224, 90, 20, 2), (12, 9, 16, 3, 8), (6, 100, 16, 27, 2), (37, 518, 90, 20, 1),
                                       (56, 448, 90, 20, 1)])
def test_linear_pixel_kmeans_half_pixel_taps_bit_exact(dev, G, H, C, K, B):
    """the OTHER reading of the code interpolation (align_corners=False: ATen's half-pixel taps, clamped at both edges) through
    the same kernels: labels AND centroids bit-exact against oracle/kmeans_linear.py under those taps, and a different map than the default reading."""
    code = torch.randn(B, G * G, C, generator=g(G * H + 1)) * (1.0 + torch.rand(B, G * G, 1, generator=g(2)))
    lab, nseg, cent = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=10, relabel=False, return_centroids=True, form="linear", align_corners=False)
    lab_ac, _ = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=10, relabel=False, form="linear")
    for b in range(B if H < 200 else 1):
        want, wcent = KL.kmeans_pixels_linear(code[b].numpy(), G, H, K, iters=10, align_corners=False)
        assert np.array_equal(lab[b].cpu().numpy(), want), f"frame {b}: {(lab[b].cpu().numpy() != want).mean()} of the labels differ"
        assert np.array_equal(cent[b].cpu().numpy(), wcent)
        assert int(nseg[b]) == len(np.unique(want))
    if H > G:
        assert not torch.equal(lab, lab_ac)
    with pytest.raises(_lib.WvnError):
        ops.kmeans_cosine_pixels(code.to(dev), G, H, K, form="direct", align_corners=False)


def test_probes_and_clustering_through_the_class_with_half_pixel_taps(dev):
    """StegoInterface(code_align_corners=False): the k-means map = the oracle's under the half-pixel taps on the GPU's own code; a probe at pixel
    resolution = argmax of the align_corners=False interpolation of its patch scores; `features` stays WVN's own align_corners=True up-sample."""
    S, G, K = 64, 8, 5
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=11, depth=1)
    head = OI.make_stego_head_state_dict(384, 90, seed=2)
    img = torch.rand(2, 3, S, S, generator=g(12))
    st = StegoInterface(dev, input_size=S, n_image_clusters=K, run_clustering=True, run_crf=False, backbone_weights=sd, head_weights=head, precision="exact",
                        flip_tta=False, cluster_resolution="pixel", allow_synthetic=True, code_align_corners=False)
    _, cluster = st.inference(img.to(dev))
    code = st.feature_tokens.cpu()
    for b in range(2):
        want = OI.relabel_ascending(KL.kmeans_cosine_labels_pixels_linear(code[b].numpy(), G, S, K, align_corners=False))
        assert np.array_equal(cluster[0, b].cpu().numpy().reshape(-1), want)
    dense = st.features.cpu()                                                              # [2, 90, S, S]: stego_interface.py:107, align_corners=True
    want_dense = torch.nn.functional.interpolate(code.reshape(2, G, G, 90).permute(0, 3, 1, 2), (S, S), mode="bilinear", align_corners=True)
    assert (dense - want_dense).abs().max().item() < 1e-5
    table = torch.randn(2, G * G, 7, generator=g(3))
    got = ops.table_bilerp_argmax(table.to(dev), G, S, align_corners=False).cpu()
    for b in range(2):
        up = OI.upsample_bilinear_fixed(table[b].reshape(G, G, 7).numpy(), S, align_corners=False)
        assert np.array_equal(got[b].numpy(), up.argmax(-1))
    with pytest.raises(_lib.WvnError):
        StegoInterface(dev, input_size=S, n_image_clusters=K, run_clustering=True, run_crf=False, backbone_weights=sd, head_weights=head, precision="exact",
                       cluster_resolution="patch", allow_synthetic=True, code_align_corners=False)
