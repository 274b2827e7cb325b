"""Host-side mirror of the reference interface (no GPU needed): containers, config, registry,
transforms, weight preparation, sharding."""
import torch

from oracle import interfaces as OI, vit as OV
from wild_visual_navigation_amd.backbone import resample_pos_embed, synthetic_vit_state_dict
from wild_visual_navigation_amd.cfg import ExperimentParams
from wild_visual_navigation_amd.distributed import shard_range
from wild_visual_navigation_amd.feature_extractor.transforms import resize_nearest_center_crop
from wild_visual_navigation_amd.model import SimpleMLP, get_model
from wild_visual_navigation_amd.utils import Batch, ConfidenceGenerator, Data, TraversabilityLoss


def test_batch_from_data_list():
    d1 = Data(x=torch.ones(3, 2), y=torch.zeros(3), y_valid=torch.tensor([True, False, True]),
              edge_index=torch.tensor([[0, 1], [1, 2]]))
    d2 = Data(x=2 * torch.ones(2, 2), y=torch.ones(2), y_valid=torch.tensor([False, True]),
              edge_index=torch.tensor([[0], [1]]))
    b = Batch.from_data_list([d1, d2])
    assert b.x.shape == (5, 2) and b.ba == 5
    assert b.ptr.tolist() == [0, 3, 5] and b.batch.tolist() == [0, 0, 0, 1, 1]
    assert b.edge_index.tolist() == [[0, 1, 3], [1, 2, 4]]  # second graph offset by ptr[1]
    assert Batch.from_data_list([]) is None


def test_confidence_generator_matches_golden(golden):
    for name, c in golden("confidence.pt").items():
        cg = ConfidenceGenerator(std_factor=float(name[2:]), method="latest_measurement")
        out = cg.update(c["x"], c["x"][: c["n_pos"]], step=0)
        assert torch.allclose(out, c["confidence"], atol=1e-6)
        assert torch.allclose(cg.mean, c["mean"]) and torch.allclose(cg.std, c["std"])
        assert set(cg.state_dict()) == {"mean", "var", "std"}


def test_traversability_loss_matches_golden(golden):
    c = golden("mlp_train.pt")["graph_pt_D90"]
    loss_fn = TraversabilityLoss(0.03, 0.5, 0.0, True, model=None, method="latest_measurement", confidence_std_factor=0.5)
    loss, aux, res = loss_fn(Data(x=c["x"], y=c["y"], y_valid=c["y_valid"]), c["res0"].clone())
    assert abs(loss.item() - c["traj"][0][0].item()) < 1e-5
    assert torch.allclose(aux["confidence"], c["confidence0"], atol=1e-6)


def test_get_model_and_state_dict_keys():
    p = ExperimentParams()
    p.model.simple_mlp_cfg.input_size = 384
    m = get_model(p.model)
    assert isinstance(m, SimpleMLP)
    assert list(m.state_dict()) == [f"layers.{i}.{n}" for i in (0, 2, 4) for n in ("weight", "bias")]
    assert sum(v.numel() for v in m.parameters()) == 119489
    assert p["optimizer"]["lr"] == 1e-3 and p.loss.w_trav == 0.03 and p["loss"]["w_reco"] == 0.5
    flat = m.flat_params()
    assert flat.numel() == 119489 and m.layers[0].weight.data_ptr() == flat.data_ptr()
    m.load_state_dict({k: torch.zeros_like(v) for k, v in m.state_dict().items()}, strict=False)
    assert float(m.flat_params().abs().sum()) == 0.0  # loading wrote through the views into the flat buffer
    hs = [256, 32, 1]
    SimpleMLP(90, hs, True)
    assert hs == [256, 32, 1]  # unlike the reference (simple_mlp.py:21-22) the argument is not mutated


def test_same_init_as_reference_under_seed_42(golden):
    c = golden("mlp_train.pt")["synthetic_D384"]
    torch.manual_seed(42)
    m = SimpleMLP(384, [256, 32, 1], True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, c["sd0"][k]), k  # identical construction order => identical default init


def test_transforms_match_oracle():
    for shape in [(1, 3, 224, 299), (2, 3, 300, 200), (1, 3, 448, 448), (1, 3, 1080, 1440)]:
        img = torch.rand(*shape)
        for size in (224, 448):
            assert torch.equal(resize_nearest_center_crop(img, size), OI.resize_nearest_center_crop(img, size))


def test_weight_prep_matches_oracle():
    sd = synthetic_vit_state_dict("vit_small", 8, 28, seed=0, depth=1)
    ref = OV.make_vit_state_dict("vit_small", 8, 28, seed=0, depth=1)
    assert all(torch.equal(sd[k], ref[k]) for k in ref)
    assert torch.equal(resample_pos_embed(sd["pos_embed"], 56), OV.interpolate_pos_embed(sd["pos_embed"], 56))
    assert torch.equal(resample_pos_embed(sd["pos_embed"], 28), sd["pos_embed"])


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_resize_crop_commutes_with_uint8_to_float():
    """Frame ingest: NEAREST resize + centre crop is an index operation, so doing it on the raw uint8 frame
    and converting afterwards (what the fused uint8 entry point does) equals the reference order (convert, then resize)."""
    import torch

    from wild_visual_navigation_amd.feature_extractor.transforms import resize_nearest_center_crop

    g = torch.Generator().manual_seed(0)
    for (h, w, size) in [(224, 299, 224), (300, 224, 224), (448, 448, 448), (540, 720, 448)]:
        u8 = torch.randint(0, 256, (2, 3, h, w), generator=g, dtype=torch.uint8)
        a = resize_nearest_center_crop(u8, size).float() / 255
        b = resize_nearest_center_crop(u8.float() / 255, size)
        assert a.shape == (2, 3, size, size) and torch.equal(a, b)


def test_gelu_polynomial_is_within_a_quarter_bf16_ulp_of_erf_gelu():
    """The bf16 fc1 epilogue evaluates GELU as x * sigmoid(g(x)), g an odd degree-7 polynomial (csrc/gemm_a384.hip).  This
    pins the claim in its comment: at most 0.25 ulp of the bf16 result away from the exact erf GELU of torch.nn.GELU."""
    import numpy as np
    from scipy.special import erf

    k = np.array([-2.296416554e+00, -1.096917929e-01, 1.435476415e-03, -1.285982656e-05])   # coefficients * -log2(e)
    x = np.linspace(-9, 9, 72001)
    x2 = x * x
    y = x * (k[0] + x2 * (k[1] + x2 * (k[2] + x2 * k[3])))
    got = x / (1 + np.exp2(y))
    want = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-300))) - 7), 2.0 ** -20)   # bf16 ulp, floored at 1e-6 absolute
    assert (np.abs(got - want) / ulp).max() < 0.26
    assert np.abs(got - want).max() < 2.5e-4


def test_gelu_degree6_form_is_within_3e_7_of_erf_gelu():
    """The split-operand fc1 epilogue (csrc/gemm_a384_x3.hip) evaluates GELU as max(x, 0) - a 2^R(a), a = min(|x|, 7), R a degree-6
    polynomial fitted to log2 erfc(a / sqrt 2) - 1.  This pins the claim in its comment: within 2.8e-7 ABSOLUTE of the exact erf GELU of
    torch.nn.GELU on the whole line with fp32 Horner evaluation (the Abramowitz-Stegun form it replaced: 1.5e-7 on erf)."""
    import numpy as np
    from scipy.special import erfc

    x = np.concatenate([np.linspace(-12, 12, 480001), [-50.0, 50.0, -1e4, 1e4]]).astype(np.float32)
    want = 0.5 * x.astype(np.float64) * erfc(-x.astype(np.float64) / np.sqrt(2.0))
    c = np.asarray((-9.999930859e-01, -1.151201725e+00, -4.587709606e-01, -5.341212451e-02, 8.080729283e-03, -7.692237268e-04,
                    3.309327076e-05), np.float32)
    a = np.minimum(np.abs(x), np.float32(7.0))
    r = np.full_like(a, c[-1])
    for ck in c[-2::-1]:
        r = (r * a + ck).astype(np.float32)
    got = np.maximum(x, 0) - (a * np.exp2(r.astype(np.float64)).astype(np.float32)).astype(np.float32)
    assert np.abs(got.astype(np.float64) - want).max() < 3.0e-7
