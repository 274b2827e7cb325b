"""BASELINE.json configs[0] on the GPU path: the quick_start.py sequence (quick_start.py:156-212) on the REFERENCE'S OWN
demo frames (assets/demo_data/*.png, committed as tests/golden/demo_frames_224.pt by oracle/make_demo_fixture.py),
through the drop-in classes -- FeatureExtractor.extract -> Data -> get_model(...).forward -> traversability map and
ConfidenceGenerator.inference_without_update -- against the CPU oracle on the same frames and the same (seeded,
DINO-layout) weights.  Exact mode (fp32 storage / FMA): <= 1e-3 absolute on every float output (north_star), segment
maps and edges bit-exact; the bf16 MFMA mode is checked at bf16-rounding tolerance."""
import pytest
import torch

from oracle import interfaces as OI, mlp as OM, segments as OS, vit as OV
from wild_visual_navigation_amd.cfg import ExperimentParams
from wild_visual_navigation_amd.feature_extractor import FeatureExtractor
from wild_visual_navigation_amd.model import get_model
from wild_visual_navigation_amd.utils import ConfidenceGenerator, Data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(golden):
    frames = golden("demo_frames_224.pt")
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=21)          # full 12-block ViT-S/8
    mlp_sd = OM.make_mlp_state_dict(384, seed=42)
    return frames, sd, mlp_sd


def _oracle(img, sd, mlp_sd, per_pixel):
    dense = OI.dino_inference(sd, img, 224, 8, 6)                                   # [1,384,224,224]
    seg = OS.segment_grid(224, 224, 32)
    feat = OS.sparsify_features(dense, seg[0, 0])
    x = dense[0].permute(1, 2, 0).reshape(-1, 384) if per_pixel else feat[seg[0, 0].reshape(-1)]
    pred = OM.mlp_forward(mlp_sd, x)
    loss_reco = ((pred[:, 1:] - x) ** 2).mean(1)
    return dense, seg, feat, pred, OM.confidence_from_stats(loss_reco, 0.9, 0.25, 0.5)


@pytest.mark.parametrize("prec,tol,tol_conf", [("fp32", 1e-3, 1e-3), ("bf16", 0.25, 0.12)])
def test_quick_start_sequence_on_reference_demo_frames(dev, setup, prec, tol, tol_conf):
    frames, sd, mlp_sd = setup
    params = ExperimentParams()
    fe = FeatureExtractor(device=dev, segmentation_type="grid", feature_type="dino", patch_size=8, backbone_type="vit_small",
                          input_size=224, slic_num_components=100, pretrained_weights=sd, precision=prec)
    params.model.simple_mlp_cfg.input_size = fe.feature_dim
    model = get_model(params.model).to(dev)
    model.eval()
    model.load_state_dict({**mlp_sd, "confidence_generator": {"mean": 0.9}}, strict=False)   # quick_start.py:142-150
    cg = ConfidenceGenerator(method=params.loss.method, std_factor=params.loss.confidence_std_factor).to(dev)
    cg.mean[0], cg.std[0] = 0.9, 0.25
    for i, name in enumerate(frames["names"]):
        img = frames["frames_u8"][i].float() / 255.0                                # quick_start.py:160-161
        for per_pixel in ((True, False) if i == 0 else (False,)):                   # per-pixel branch once (50k rows)
            dense_o, seg_o, feat_o, pred_o, conf_o = _oracle(img[None], sd, mlp_sd, per_pixel)
            edges, feat, seg, center, dense = fe.extract(img=img[None].to(dev), return_centers=False,
                                                         return_dense_features=True, n_random_pixels=100)
            assert torch.equal(seg.cpu(), seg_o[0, 0]), name
            assert torch.equal(edges.cpu(), OS.adjacency_list(seg_o).T), name
            assert (feat.cpu() - feat_o).abs().max().item() < tol, name
            assert (dense.cpu() - dense_o).abs().max().item() < tol, name
            x = dense[0].permute(1, 2, 0).reshape(-1, dense.shape[1]) if per_pixel else feat[seg.reshape(-1)]
            data = Data(x=x)
            prediction = model.forward(data)
            out_trav = prediction.reshape(224, 224, -1)[:, :, 0]
            assert (prediction.cpu() - pred_o).abs().max().item() < tol, name
            loss_reco = torch.nn.functional.mse_loss(prediction[:, 1:], data.x, reduction="none").mean(dim=1)
            confidence = cg.inference_without_update(x=loss_reco).reshape(224, 224)
            assert (confidence.cpu().reshape(-1) - conf_o).abs().max().item() < tol_conf, name
            assert out_trav.shape == (224, 224) and 0.0 < float(out_trav.min()) and float(out_trav.max()) < 1.0


def test_quick_start_body_through_the_reference_import_paths(dev, setup):
    """The caller's own lines -- quick_start.py:6-18 (imports), :111-128 (ConfidenceGenerator, FeatureExtractor with the
    quick_start keyword set), :131-143 (get_model, load_state_dict(strict=False) with the extra "confidence_generator" key),
    :165-181 (ImageProjector.resize_image, extract), :183-210 -- executed against ``wild_visual_navigation`` as aliased by
    ``wild_visual_navigation_amd.dropin.install()``, with precision "exact" (the <= 1e-3 mode on the matrix pipe) and the
    reference's default segmentation ("slic"); every float output against the CPU oracle."""
    import wild_visual_navigation_amd.dropin as dropin

    dropin.install(force_synthetic=True)
    from wild_visual_navigation import WVN_ROOT_DIR                                   # noqa: F401  quick_start.py:6
    from wild_visual_navigation.feature_extractor import FeatureExtractor as FE        # :7
    from wild_visual_navigation.cfg import ExperimentParams as EP                      # :8
    from wild_visual_navigation.image_projector import ImageProjector                  # :9
    from wild_visual_navigation.model import get_model as gm                           # :10
    from wild_visual_navigation.utils import ConfidenceGenerator as CG, AnomalyLoss    # noqa: F401  :11-12
    from wild_visual_navigation.utils import Data as D2                                # :18
    from oracle import slic as OSL

    frames, sd, mlp_sd = setup
    params = EP()
    confidence_generator = CG(method=params.loss.method, std_factor=params.loss.confidence_std_factor)
    feature_extractor = FE(device=dev, segmentation_type="slic", feature_type="dino", patch_size=8, backbone_type="vit_small",
                           input_size=224, slic_num_components=100, pretrained_weights=sd, precision="exact")
    params.model.simple_mlp_cfg.input_size = feature_extractor.feature_dim
    params.model.double_mlp_cfg.input_size = feature_extractor.feature_dim
    params.model.simple_gcn_cfg.input_size = feature_extractor.feature_dim
    params.model.linear_rnvp_cfg.input_size = feature_extractor.feature_dim
    model = gm(params.model).to(dev)
    model.eval()
    state = {**mlp_sd, "confidence_generator": {"mean": torch.tensor([0.9]), "var": torch.tensor([[0.06]]), "std": torch.tensor([0.25])}}
    model.load_state_dict(state, strict=False)
    cg = state["confidence_generator"]
    confidence_generator.var, confidence_generator.mean, confidence_generator.std = (
        torch.nn.Parameter(cg["var"], requires_grad=False), torch.nn.Parameter(cg["mean"], requires_grad=False),
        torch.nn.Parameter(cg["std"], requires_grad=False))                          # quick_start.py:146-150 assigns the fields
    confidence_generator = confidence_generator.to(dev)
    raw = torch.zeros(3, 224, 299, dtype=torch.uint8)
    raw[:, :, 37:261] = frames["frames_u8"][0]                                        # a 299 x 224 frame like assets/demo_data
    torch_image = raw.to(dev).float() / 255.0
    image_projector = ImageProjector(K=torch.eye(4, device=dev)[None], h=224, w=299, new_h=224, new_w=224)
    torch_image = image_projector.resize_image(torch_image)
    assert torch_image.shape == (3, 224, 224)
    _, feat, seg, center, dense_feat = feature_extractor.extract(img=torch_image[None], return_centers=False,
                                                                 return_dense_features=True, n_random_pixels=100)
    img_cpu = torch_image.cpu()
    dense_o = OI.dino_inference(sd, img_cpu[None], 224, 8, 6)
    seg_o = torch.from_numpy(OSL.slic(img_cpu.numpy(), 100, 10.0)).long()
    assert torch.equal(seg.cpu(), seg_o)                                               # bit-exact segment-index map
    assert (dense_feat.cpu() - dense_o).abs().max().item() < 1e-3
    feat_o = OS.sparsify_features(dense_o, seg_o)
    ok = ~torch.isnan(feat_o).any(1)
    assert (feat.cpu()[ok] - feat_o[ok]).abs().max().item() < 1e-3
    input_feat = torch.nan_to_num(feat)[seg.reshape(-1)]
    data = D2(x=input_feat)
    prediction = model.forward(data)
    out_trav = prediction.reshape(224, 224, -1)[:, :, 0]
    pred_o = OM.mlp_forward(mlp_sd, torch.nan_to_num(feat_o)[seg_o.reshape(-1)])
    assert (prediction.cpu() - pred_o).abs().max().item() < 1e-3
    loss_reco = torch.nn.functional.mse_loss(prediction[:, 1:], data.x, reduction="none").mean(dim=1)
    confidence = confidence_generator.inference_without_update(x=loss_reco)
    conf_o = OM.confidence_from_stats(((pred_o[:, 1:] - torch.nan_to_num(feat_o)[seg_o.reshape(-1)]) ** 2).mean(1), 0.9, 0.25, 0.5)
    assert (confidence.cpu() - conf_o).abs().max().item() < 1e-3 and out_trav.shape == (224, 224)
