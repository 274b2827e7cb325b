"""Fused per-pixel traversability inference (csrc/pixel_mlp.hip) against the CPU oracle of the
reference sequence wvn_feature_extractor_node.py:319-363 / quick_start.py:183-210:
   dense = bilinear(align_corners) upsample of the patch tokens; out = SimpleMLP(dense rows);
   trav = out[:, 0]; loss_reco = mse(out[:, 1:], dense); conf = ConfidenceGenerator.inference_without_update(loss_reco).

The kernel is the bf16 speed path (bf16 MFMA operands, fp32 accumulation, layer 1 evaluated at patch resolution, no dense
tensor).  Two oracles, two tolerances:
  * "same operands": the oracle run on the bf16-rounded tokens and bf16-rounded weights -- what remains is the rounding
    of the intermediate activations (Z, h1, h2) to bf16 and the fp32 summation order;
  * "reference": the oracle on the fp32 weights -- adds the 2^-9 operand rounding every bf16 GEMM path has.
Tolerances are written next to each assert."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle import mlp as OM, vit as OV
from wild_visual_navigation_amd import _lib
from wild_visual_navigation_amd.cfg import ExperimentParams
from wild_visual_navigation_amd.feature_extractor import FeatureExtractor
from wild_visual_navigation_amd.model import get_model
from wild_visual_navigation_amd.utils import ConfidenceGenerator, Data

pytestmark = pytest.mark.gpu

MEAN, STD, FAC = 0.9, 0.25, 0.5


def g(seed):
    return torch.Generator().manual_seed(seed)


def bf(x):
    return x.to(torch.bfloat16).float()


def _model(dev, mlp_sd, D=384):
    params = ExperimentParams()
    params.model.simple_mlp_cfg.input_size = D
    model = get_model(params.model).to(dev)
    model.eval()
    model.load_state_dict(mlp_sd, strict=False)
    return model


def _oracle(tokens, G, H, W, sd):
    """tokens [B, G*G, D] fp32 -> trav, loss_reco, conf [B,H,W] (reference order of operations, fp32)."""
    B, D = tokens.shape[0], tokens.shape[2]
    dense = F.interpolate(tokens.reshape(B, G, G, D).permute(0, 3, 1, 2), (H, W), mode="bilinear", align_corners=True)
    x = dense.permute(0, 2, 3, 1).reshape(-1, D)
    pred = OM.mlp_forward(sd, x)
    loss = ((pred[:, 1:] - x) ** 2).mean(1)
    conf = OM.confidence_from_stats(loss, MEAN, STD, FAC)
    return pred[:, 0].reshape(B, H, W), loss.reshape(B, H, W), conf.reshape(B, H, W)


def _run(dev, model, tokens_bf, B, G, H, W):
    zx = torch.full((B * G * G, 640), float("nan"), dtype=torch.bfloat16, device=dev)   # scratch columns need no init
    zx[:, 256:] = tokens_bf.reshape(B * G * G, 384).to(dev)
    trav, conf, loss = model.forward_per_pixel(zx, B, G, (H, W), MEAN, STD, FAC, want_loss=True)
    return trav.cpu(), loss.cpu(), conf.cpu()


@pytest.mark.parametrize("B,G,H,W", [(1, 28, 224, 224), (2, 28, 230, 251), (1, 56, 448, 448), (3, 14, 224, 112)])
def test_fused_per_pixel_matches_oracle(dev, B, G, H, W):
    sd = OM.make_mlp_state_dict(384, seed=7)
    tokens = 2.0 * torch.randn(B, G * G, 384, generator=g(B + G))      # DINO-like magnitude
    tokens_bf = tokens.to(torch.bfloat16)
    model = _model(dev, sd)
    trav, loss, conf = _run(dev, model, tokens_bf, B, G, H, W)
    assert torch.isfinite(trav).all() and torch.isfinite(loss).all()

    sd_bf = {k: (bf(v) if k.endswith("weight") else v) for k, v in sd.items()}
    t0, l0, c0 = _oracle(tokens_bf.float(), G, H, W, sd_bf)              # same operands
    # pre-sigmoid logits are O(1); bf16 rounding of h1/h2 (2^-9 relative, 256/32 terms) -> a few 1e-3 absolute
    assert (trav - t0).abs().max().item() < 4e-3
    assert ((loss - l0).abs() / l0).max().item() < 4e-3
    assert (conf - c0).abs().max().item() < 2e-2                        # conf = 1 - (loss - lo) / (2 std): loss error / 0.5
    t1, l1, c1 = _oracle(tokens_bf.float(), G, H, W, sd)                # reference weights (fp32)
    assert (trav - t1).abs().max().item() < 1e-2
    assert ((loss - l1).abs() / l1).max().item() < 1e-2


def test_weight_split_restores_partition_of_unity(dev):
    """Constant token field: interpolation must return the constant (to 2^-17 with the hi+lo weight split), so
    trav / loss are the same for every pixel."""
    sd = OM.make_mlp_state_dict(384, seed=9)
    row = (3.0 * torch.randn(1, 1, 384, generator=g(1))).to(torch.bfloat16)
    tokens_bf = row.expand(1, 28 * 28, 384).contiguous()
    model = _model(dev, sd)
    trav, loss, _ = _run(dev, model, tokens_bf, 1, 28, 224, 224)
    assert (trav.max() - trav.min()).item() < 2e-4
    assert ((loss.max() - loss.min()) / loss.mean()).item() < 1e-3


def test_unsupported_configurations_are_refused(dev):
    sd = OM.make_mlp_state_dict(384, seed=7)
    model = _model(dev, sd)
    zx = torch.zeros(28 * 28, 640, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.WvnError):
        model.forward_per_pixel(zx, 1, 28, (112, 112))                   # 15 * 27/111 > 2: window would exceed 4x4 tokens
    with pytest.raises(_lib.WvnError):
        model.forward_per_pixel(zx[:, :512], 1, 28, (224, 224))          # row too short
    d64 = _lib.MlpDesc(64, 256, 32, 0)
    assert _lib.lib().wvn_pixel_mlp_pack_bytes(C.byref(d64)) == 0        # only D = 384 (DINO) and D = 90 (STEGO code)


def test_predict_per_pixel_equals_unfused_sequence(dev, golden):
    """Drop-in level, on the reference's demo frames: FeatureExtractor.predict_per_pixel vs the reference call sequence on
    the same (bf16-mode) extractor: extract(return_dense_features) -> model.forward -> column 0 / confidence."""
    frames = golden("demo_frames_224.pt")["frames_u8"][:2]
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=21, depth=4)
    mlp_sd = OM.make_mlp_state_dict(384, seed=42)
    fe = FeatureExtractor(device=dev, segmentation_type="grid", feature_type="dino", patch_size=8, backbone_type="vit_small",
                          input_size=224, pretrained_weights=sd, precision="bf16")
    model = _model(dev, mlp_sd)
    cg = ConfidenceGenerator(method="latest_measurement", std_factor=FAC).to(dev)
    cg.mean[0], cg.std[0] = MEAN, STD
    trav, conf, loss = fe.predict_per_pixel(frames.to(dev), model, cg, want_loss=True)
    assert trav.shape == (2, 224, 224) and conf.shape == (2, 224, 224)
    for i in range(2):
        img = (frames[i:i + 1].float() / 255).to(dev)
        _, _, _, _, dense = fe.extract(img=img, return_centers=False, return_dense_features=True)
        x = dense[0].permute(1, 2, 0).reshape(-1, 384)
        pred = model.forward(Data(x=x))
        lr = ((pred[:, 1:] - x) ** 2).mean(1)
        c = cg.inference_without_update(lr)
        assert (trav[i].reshape(-1) - pred[:, 0]).abs().max().item() < 1e-2
        assert ((loss[i].reshape(-1) - lr).abs() / lr).max().item() < 1e-2
        assert (conf[i].reshape(-1) - c).abs().max().item() < 5e-2


def test_confidence_state_from_device_memory(dev):
    """conf_state (device {mean, std, std_factor}) overrides the scalar arguments: the form a captured HIP graph needs."""
    sd = OM.make_mlp_state_dict(384, seed=7)
    model = _model(dev, sd)
    zx = torch.zeros(28 * 28, 640, dtype=torch.bfloat16, device=dev)
    zx[:, 256:] = (2.0 * torch.randn(28 * 28, 384, generator=g(3))).to(torch.bfloat16).to(dev)
    a = model.forward_per_pixel(zx, 1, 28, (224, 224), MEAN, STD, FAC)
    state = torch.tensor([MEAN, STD, FAC], dtype=torch.float32, device=dev)
    b = model.forward_per_pixel(zx, 1, 28, (224, 224), 123.0, 456.0, 7.0, conf_state=state)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("B,G,H,W", [(1, 28, 224, 224), (2, 28, 230, 251), (1, 56, 448, 448)])
def test_exact_mode_fused_per_pixel_within_1e3(dev, B, G, H, W):
    """Exact mode (hi + lo split MFMA operands, fp32 layer-1 GEMM): <= 1e-3 absolute on the traversability logit's sigmoid and
    on the reconstruction loss against the fp32 reference sequence on the SAME fp32 tokens and fp32 weights (north_star bar)."""
    sd = OM.make_mlp_state_dict(384, seed=7)
    tokens = 2.0 * torch.randn(B, G * G, 384, generator=g(B + G))
    model = _model(dev, sd)
    trav, conf, loss = model.forward_per_pixel_exact(tokens.reshape(B * G * G, 384).to(dev), B, G, (H, W), MEAN, STD, FAC,
                                                     want_loss=True)
    t0, l0, c0 = _oracle(tokens, G, H, W, sd)
    assert (trav.cpu() - t0).abs().max().item() < 1e-3
    assert (loss.cpu() - l0).abs().max().item() < 1e-3
    assert (conf.cpu() - c0).abs().max().item() < 2e-3            # conf = 1 - (loss - lo) / (2 std), std = 0.25


def test_predict_per_pixel_exact_mode_on_demo_frames(dev, golden):
    """fp32 extractor -> predict_per_pixel takes the exact fused kernel: <= 1e-3 against the CPU oracle of the reference
    sequence (dense up-sample -> MLP -> MSE -> confidence) on the reference's demo frames."""
    from oracle import interfaces as OI

    frames = golden("demo_frames_224.pt")["frames_u8"][:1]
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=21, depth=3)
    mlp_sd = OM.make_mlp_state_dict(384, seed=42)
    fe = FeatureExtractor(device=dev, segmentation_type="grid", feature_type="dino", patch_size=8, backbone_type="vit_small",
                          input_size=224, pretrained_weights=sd, precision="fp32")
    model = _model(dev, mlp_sd)
    cg = ConfidenceGenerator(method="latest_measurement", std_factor=FAC).to(dev)
    cg.mean[0], cg.std[0] = MEAN, STD
    trav, conf, loss = fe.predict_per_pixel(frames.to(dev), model, cg, want_loss=True)
    img = frames.float() / 255
    dense = OI.dino_inference(sd, img, 224, 8, 6)
    x = dense[0].permute(1, 2, 0).reshape(-1, 384)
    pred = OM.mlp_forward(mlp_sd, x)
    lr = ((pred[:, 1:] - x) ** 2).mean(1)
    assert (trav[0].reshape(-1).cpu() - pred[:, 0]).abs().max().item() < 1e-3
    assert (loss[0].reshape(-1).cpu() - lr).abs().max().item() < 1e-3
    assert (conf[0].reshape(-1).cpu() - OM.confidence_from_stats(lr, MEAN, STD, FAC)).abs().max().item() < 2e-3


@pytest.mark.parametrize("B,G,H,W", [(1, 28, 224, 224), (2, 28, 230, 251)])
def test_stego_code_features_90d(dev, B, G, H, W):
    """The live node's default configuration (feature_type: stego, prediction_per_pixel: True, default.yaml:22-29): 90-d
    code features, MLP 90 -> 256 -> 32 -> 91.  bf16 form against the same-operand oracle, exact form within 1e-3."""
    sd = OM.make_mlp_state_dict(90, seed=17)
    tokens = 0.5 * torch.randn(B, G * G, 90, generator=g(B + G + 1))
    model = _model(dev, sd, D=90)
    assert model.ZX_COLS == 384
    # exact
    trav, conf, loss = model.forward_per_pixel_exact(tokens.reshape(B * G * G, 90).to(dev), B, G, (H, W), MEAN, STD, FAC, want_loss=True)
    t0, l0, c0 = _oracle(tokens, G, H, W, sd)
    assert (trav.cpu() - t0).abs().max().item() < 1e-3 and (loss.cpu() - l0).abs().max().item() < 1e-3
    assert (conf.cpu() - c0).abs().max().item() < 2e-3
    # bf16
    tb = tokens.to(torch.bfloat16)
    zx = torch.zeros(B * G * G, 384, dtype=torch.bfloat16, device=dev)
    zx[:, :256] = float("nan")                                        # scratch columns need no init
    zx[:, 256:346] = tb.reshape(B * G * G, 90).to(dev)
    trav, conf, loss = model.forward_per_pixel(zx, B, G, (H, W), MEAN, STD, FAC, want_loss=True)
    sd_bf = {k: (bf(v) if k.endswith("weight") else v) for k, v in sd.items()}
    t1, l1, c1 = _oracle(tb.float(), G, H, W, sd_bf)
    assert (trav.cpu() - t1).abs().max().item() < 4e-3
    assert ((loss.cpu() - l1).abs() / l1).max().item() < 4e-3


def test_predict_per_pixel_with_stego_features(dev, golden):
    """Drop-in level with feature_type='stego' (code features), both precisions, against the reference call sequence on the
    same extractor: dense code features -> model.forward -> column 0 / confidence."""
    frames = golden("demo_frames_224.pt")["frames_u8"][:1].to(dev)
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=23, depth=2)
    mlp_sd = OM.make_mlp_state_dict(90, seed=43)
    for prec, tol in (("fp32", 1e-3), ("bf16", 2e-2)):
        fe = FeatureExtractor(device=dev, segmentation_type="stego", feature_type="stego", patch_size=8, backbone_type="vit_small",
                              input_size=224, pretrained_weights=sd, precision=prec, n_image_clusters=8)
        model = _model(dev, mlp_sd, D=90)
        cg = ConfidenceGenerator(method="latest_measurement", std_factor=FAC).to(dev)
        cg.mean[0], cg.std[0] = 0.05, 0.02
        trav, conf, loss = fe.predict_per_pixel(frames, model, cg, want_loss=True)
        _, _, _, _, dense = fe.extract(img=frames.float() / 255, return_centers=False, return_dense_features=True)
        x = dense[0].permute(1, 2, 0).reshape(-1, 90)
        pred = model.forward(Data(x=x))
        lr = ((pred[:, 1:] - x) ** 2).mean(1)
        assert trav.shape == (1, 224, 224)
        assert (trav[0].reshape(-1) - pred[:, 0]).abs().max().item() < tol, prec
        assert ((loss[0].reshape(-1) - lr).abs() / lr).max().item() < 10 * tol, prec

