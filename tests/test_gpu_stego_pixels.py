"""The other reading of the absent STEGO package (stego_interface.py:91-109): get_code averages the code with the
flipped-back code of the mirrored frame, postprocess runs the k-means over the H x H up-sampled code PIXELS.  The pixel k-means
never builds that array (csrc/stego.hip: rows interpolated on the fly from the patch codes); labels are bit-exact against the CPU
oracle on identical code, at the small sizes and at 448^2."""
import numpy as np
import pytest
import torch

from oracle import interfaces as OI, kmeans_linear as KL, vit as OV
from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd.feature_extractor import FeatureExtractor, StegoInterface

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def test_upsample_kernel_matches_fixed_order_oracle(dev):
    """wvn_upsample_bilinear and the on-the-fly interpolation share ONE explicitly rounded operation order (csrc/common.h);
    oracle/interfaces.upsample_bilinear_fixed restates it bit for bit (ATen's CPU kernel rounds in another order: <= 1e-6)."""
    code = torch.randn(2, 7 * 7, 16, generator=g(0)) * 3
    got = ops.upsample_bilinear(code.to(dev), 7, 50).cpu()                              # [2, 16, 50, 50]
    for b in range(2):
        want = OI.upsample_bilinear_fixed(code[b].reshape(7, 7, 16).numpy(), 50)
        assert np.array_equal(got[b].permute(1, 2, 0).numpy(), want)
    aten = torch.nn.functional.interpolate(code.reshape(2, 7, 7, 16).permute(0, 3, 1, 2), (50, 50), mode="bilinear", align_corners=True)
    assert (got - aten).abs().max().item() < 2e-6


@pytest.fixture
def assign_form(request):
    """The assignment kernels of the pixel k-means: the opt-in screened bf16-MFMA form (K <= 20: the label is taken from the
    split-operand similarities where their margin proves it, from the exact fmaf chains elsewhere), the same kernel with EVERY row
    sent down its exact path, the plain VALU form (one v_fma_f32 per cluster and channel), and the packed forms (default: 5) -- two
    clusters' chains in the halves of v_pk_fma_f32, with (5) or without (4) the interpolation on channel pairs."""
    from wild_visual_navigation_amd import _lib
    _lib.lib().wvn_debug_kmeans_assign_form(request.param)
    yield request.param
    _lib.lib().wvn_debug_kmeans_assign_form(-1)


@pytest.mark.parametrize("assign_form", [1, 2, 0, 4, 5], indirect=True, ids=["screened", "screened-all-exact", "valu-plain", "valu-packed-dots", "valu-packed"])
@pytest.mark.parametrize("G,H,C,K,B", [(8, 64, 90, 5, 2), (7, 50, 16, 4, 3), (28, 224, 90, 20, 1), (5, 33, 90, 6, 2), (28, 224, 90, 20, 16),
                                       (9, 70, 90, 17, 9), (9, 70, 90, 19, 3), (8, 64, 90, 16, 2)])
def test_pixel_kmeans_bit_exact(dev, assign_form, G, H, C, K, B):
    """(7, 50) and (5, 33): chunks straddle image rows, the last group is ragged; (28, 224): the live node's default size; B = 16: the
    frame -> XCD mapping of whole multiples of 8 frames; (9, 70, K = 17): two 64-pixel groups per row with a ragged second one, a
    partial last centroid block; K = 17 / 19 / 16: a half-used, a half-used last and two unused cluster PAIRS of the packed form (the
    first packed kernel mislabelled 0.5 % of the pixels exactly there -- in-flight scalar registers copied under register pressure)."""
    code = torch.randn(B, G * G, C, generator=g(G * H)) * (1.0 + torch.rand(B, G * G, 1, generator=g(1)))
    lab, nseg, cent = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=10, relabel=False, return_centroids=True, form="direct")
    lab2, nseg2 = ops.kmeans_cosine_pixels(code.to(dev), G, H, K, iters=10, relabel=True, form="direct")
    # the materialised route on the GPU: up-sample, normalise, cluster the H*H rows -- must give the same bits, labels AND
    # centroids (labels alone are robust to last-place differences of the sums: a 1-ulp square root went unnoticed that way)
    dense = ops.upsample_bilinear(code.to(dev), G, H).permute(0, 2, 3, 1).reshape(B, H * H, C).contiguous()
    lab_m, _, cent_m = ops.kmeans_cosine(dense, K, iters=10, relabel=False, return_centroids=True)
    assert torch.equal(lab, lab_m) and torch.equal(cent, cent_m)
    for b in range(B):
        want = OI.kmeans_cosine_labels_pixels(code[b].numpy(), G, H, K, iters=10)
        assert np.array_equal(lab[b].cpu().numpy(), want), f"frame {b}"
        assert np.array_equal(lab2[b].cpu().numpy(), OI.relabel_ascending(want))
        assert int(nseg[b]) == len(np.unique(want)) == int(nseg2[b])


@pytest.mark.parametrize("assign_form", [1, 2, 0, 4, 5], indirect=True, ids=["screened", "screened-all-exact", "valu-plain", "valu-packed-dots", "valu-packed"])
def test_pixel_kmeans_at_448_against_oracle(dev, assign_form):
    """BASELINE size: one 448^2 frame, 56 x 56 x 90 code, K = 20: 200 704 points x 11 assignment passes, labels bit-exact."""
    G, H, C, K = 56, 448, 90, 20
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=2, depth=1)
    head = OI.make_stego_head_state_dict(384, 90, seed=2)
    img = torch.rand(1, 3, H, H, generator=g(3))
    si = StegoInterface(dev, input_size=H, n_image_clusters=K, run_crf=False, run_clustering=True, backbone_weights=sd, head_weights=head,
                        precision="fp16", flip_tta=True, cluster_resolution="pixel", kmeans_form="direct", allow_synthetic=True)
    _, clu = si.inference(img.to(dev))
    code = si.feature_tokens.cpu()                                                      # the GPU's own code: identical input
    want = OI.relabel_ascending(OI.kmeans_cosine_labels_pixels(code[0].numpy(), G, H, K))
    assert clu.shape == (1, 1, H, H) and clu.dtype == torch.int32
    assert np.array_equal(clu[0, 0].cpu().numpy().reshape(-1), want)
    assert int(si._n_segments[0]) == len(np.unique(want))


def test_upstream_reading_end_to_end(dev):
    """flip TTA + pixel clustering + general pooling through FeatureExtractor.extract_batch (what bench.py's headline leg
    runs), exact precision: code within 1e-3 of the CPU oracle's flip-averaged code, labels bit-exact given the GPU's code,
    pooled features == per-segment means of the explicitly up-sampled code."""
    S, G, K = 64, 8, 5
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=11, depth=2)
    head = OI.make_stego_head_state_dict(384, 90, seed=2)
    img = torch.rand(2, 3, S, S, generator=g(12))
    fe = FeatureExtractor(dev, segmentation_type="stego", feature_type="stego", input_size=S, pretrained_weights=sd, head_weights=head,
                          n_image_clusters=K, precision="exact", flip_tta=True, cluster_resolution="pixel")
    feat, seg, nseg = fe.extract_batch(img.to(dev))
    x = OI.normalize(img)
    tok, tok_m = OV.vit_tokens(sd, x, 8, 6)[:, 1:], OV.vit_tokens(sd, x.flip(-1), 8, 6)[:, 1:]
    code_ref = OI.stego_code_flip_average(head, tok, tok_m, G)
    code = fe._extractor.feature_tokens.cpu()
    assert (code - code_ref).abs().max().item() < 1e-3
    dense = fe._extractor.features.cpu()                                                # [2, 90, S, S]
    for b in range(2):
        want = OI.relabel_ascending(KL.kmeans_cosine_labels_pixels_linear(code[b].numpy(), G, S, K))   # (the class default: the linear form)
        assert np.array_equal(seg[b].cpu().numpy().reshape(-1), want)
        for s_ in range(int(nseg[b])):
            m = torch.from_numpy(want.reshape(S, S) == s_)
            assert (feat[b, s_].cpu() - dense[b][:, m].mean(1)).abs().max().item() < 1e-4


def test_dense_row_fallback_for_shapes_outside_the_fused_kernels(dev):
    """a code dimension the fused pixel-resolution kernels have no instantiation for (a 64-d head) still clusters at pixel
    resolution -- through the materialised up-sampled rows and the patch-resolution kernel; labels = the oracle's cosine k-means on the same rows."""
    S, G, K = 64, 8, 5
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=11, depth=1)
    head = OI.make_stego_head_state_dict(384, 64, seed=3)
    img = torch.rand(1, 3, S, S, generator=g(13))
    st = StegoInterface(dev, input_size=S, n_image_clusters=K, run_clustering=True, run_crf=False, backbone_weights=sd, head_weights=head,
                        precision="exact", flip_tta=False, cluster_resolution="pixel", allow_synthetic=True)
    _, cluster = st.inference(img.to(dev))
    assert cluster.shape == (1, 1, S, S)
    rows = ops.upsample_bilinear(st.feature_tokens, G, S).permute(0, 2, 3, 1).reshape(S * S, 64).cpu().numpy()
    want = OI.relabel_ascending(OI.kmeans_cosine_labels(rows, K))
    assert np.array_equal(cluster[0, 0].cpu().numpy().reshape(-1), want)
