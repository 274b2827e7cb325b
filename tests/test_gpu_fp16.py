"""The fp16-operand form of the speed path (precision "fp16" = WVN_PREC_F16; csrc/operand.h): the bf16 kernels compiled for
v_mfma_f32_32x32x16_f16 / v_cvt_pk_f16_f32 / v_dot2c_f32_f16.  Same tests as the bf16 form, with gates at about twice the
measured error (printed): 11 significand bits instead of 8 must show as ~8x smaller errors, or an operand is still bf16."""
import pytest
import torch

from oracle import interfaces as OI, vit as OV
from wild_visual_navigation_amd import _lib, ops
from wild_visual_navigation_amd._lib import check, lib, ptr, stream
from wild_visual_navigation_amd.backbone import VitBackbone

pytestmark = pytest.mark.gpu
C = 0.125 * 1.4426950408889634


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("M,N,K", [(256, 384, 384), (300, 1536, 384), (515, 384, 1536), (128, 90, 768), (200, 384, 192)])
@pytest.mark.parametrize("epi", ["lowp", "gelu", "relu", "f32", "resid"])
def test_gemm_f16_against_fp64(dev, M, N, K, epi):
    a = torch.randn(M, K, generator=g(1)).to(dev).half()
    w = (torch.randn(N, K, generator=g(2)) * 0.05).to(dev).half()
    bias = (torch.randn(N, generator=g(3)) * 0.1).to(dev)
    ref = a.double() @ w.double().T + bias.double()
    if epi == "lowp":
        got = ops.gemm_bf16(a, w, bias, _lib.EPI_BF16)
        assert got.dtype == torch.float16
    elif epi == "gelu":
        got, ref = ops.gemm_bf16(a, w, bias, _lib.EPI_GELU_BF16), torch.nn.functional.gelu(ref)
    elif epi == "relu":
        got, ref = ops.gemm_bf16(a, w, bias, _lib.EPI_RELU_BF16), ref.clamp_min(0)
    elif epi == "f32":
        got = ops.gemm_bf16(a, w, bias, _lib.EPI_F32)
    else:
        c0 = torch.randn(M, N, generator=g(4)).to(dev)
        got, ref = ops.gemm_bf16(a, w, bias, _lib.EPI_RESID_F32, out=c0.clone()), ref + c0.double()
    err = (got.double() - ref).abs().max().item()
    # operands are exact fp16 here: what is left is fp32 accumulation (+ one fp16 rounding of the result, 2^-11 relative, and the
    # speed path's GELU approximation, 0.25 bf16 ulp = 1e-3 relative at worst)
    tol = {"lowp": 4e-3, "gelu": 6e-3, "relu": 4e-3, "f32": 2e-4, "resid": 2e-4}[epi]
    assert err < tol, err


def _attn(dev, q_in, k, v, ntok, dtype, variant=-1):
    B, h = q_in.shape[:2]
    npad = (ntok + 127) // 128 * 128

    def pad(t, fill):
        out = torch.full((B, h, npad, 64), fill, dtype=t.dtype)
        out[:, :, :ntok] = t
        return out

    vt = pad(v, 1e3).transpose(-1, -2)[..., ops.vt_token_order(npad)].contiguous().to(dev)
    qd, kd = pad(q_in, 50.0).to(dev), pad(k, -1e3).to(dev)
    out = torch.empty(B * ntok, h * 64, dtype=dtype, device=dev)
    fn = lib().wvn_attention_f16 if dtype == torch.float16 else lib().wvn_attention_bf16
    lib().wvn_debug_attention_variant(variant)
    try:
        check(fn(ptr(qd), ptr(kd), ptr(vt), ptr(out), B, h, ntok, npad, 0.0, stream()))
        torch.cuda.synchronize()
    finally:
        lib().wvn_debug_attention_variant(-1)
    return out.float().cpu()


def _attn_ref(q_in, k, v):
    B, h, ntok, _ = q_in.shape
    s = (q_in.double() / C) @ k.double().transpose(-1, -2) * 0.125
    return (torch.softmax(s, dim=-1) @ v.double()).permute(0, 2, 1, 3).reshape(B * ntok, h * 64)


@pytest.mark.parametrize("case", ["plain", "climbing", "overflow", "tail_spike", "negative"])
@pytest.mark.parametrize("ntok", [197, 3137])
def test_attention_f16_lazy_and_exact_max(dev, ntok, case):
    """The cases of test_gpu_attention_lazy.py on fp16 operands: P must stay below fp16's 65504, so the lazy form's alarm rings
    at 2^12 (csrc/attention_bf16.hip); 'overflow' drives a score far beyond it inside one tile."""
    B, h = (1, 2) if ntok > 1000 else (2, 3)
    gen = torch.Generator().manual_seed(ntok * 7 + len(case))
    q, k, v = (torch.randn(B, h, ntok, 64, generator=gen) for _ in range(3))
    if case == "climbing":
        k = k * torch.linspace(0.2, 6.0, ntok)[None, None, :, None]
    elif case == "overflow":
        k[:, :, ntok // 2] = q[:, :, ntok // 3] * 30.0
    elif case == "tail_spike":
        k[:, :, ntok - 1] = k[:, :, ntok - 1] * 8.0
    elif case == "negative":
        k[:, :, :64] = -q[:, :, :1] * 3.0
    q_in, k, v = (q * C).half(), k.half(), v.half()
    ref = _attn_ref(q_in.float(), k.float(), v.float())
    lazy = _attn(dev, q_in, k, v, ntok, torch.float16, 1)
    exact = _attn(dev, q_in, k, v, ntok, torch.float16, 0)
    assert torch.isfinite(lazy).all() and torch.isfinite(exact).all()
    e_lazy, e_exact = (lazy.double() - ref).abs().max().item(), (exact.double() - ref).abs().max().item()
    print(f"attention f16 ntok={ntok} {case}: lazy {e_lazy:.2e} exact-max {e_exact:.2e}")
    assert e_lazy < 3e-3 and e_exact < 3e-3          # the bf16 form's gate on the same cases is 2e-2
    assert (lazy - exact).abs().max().item() < 3e-3


@pytest.mark.parametrize("S,depth,B", [(64, 2, 3), (224, 12, 2), (448, 12, 1)])
def test_vit_fp16_mode(dev, S, depth, B):
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=0, depth=depth)
    img = torch.rand(B, 3, S, S, generator=g(1))
    want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
    got = VitBackbone(sd, S, 8, 6, device=dev, precision="fp16", max_chunk=2).forward_tokens(img.to(dev)).cpu()
    ref16 = VitBackbone(sd, S, 8, 6, device=dev, precision="bf16", max_chunk=2).forward_tokens(img.to(dev)).cpu()
    e16, ebf = (got - want).abs().max().item(), (ref16 - want).abs().max().item()
    print(f"fp16 tokens S={S} depth={depth}: max|err| {e16:.3e} rel-L2 {rel_l2(got, want):.3e}   (bf16: {ebf:.3e} / {rel_l2(ref16, want):.3e})")
    assert e16 < 6e-3 and rel_l2(got, want) < 1.2e-3
    assert e16 < ebf / 3                              # 8x fewer rounding bits lost: anything near the bf16 error means a bf16 operand


def test_fp16_shipped_instantiations_at_448(dev):
    """fp16 path exactly as bench.py drives it, scaled down in depth only (see test_bf16_shipped_instantiations_at_448): fused
    LayerNorm + QKV, XCD-ordered lazy attention, projection + LayerNorm + MLP kernel; then the separate kernels; then frames one
    at a time (non-XCD attention, tiled fc2)."""
    B = 16
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=2, depth=2)
    img = torch.rand(B, 3, 448, 448, generator=g(11))
    want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
    auto = VitBackbone(sd, 448, 8, 6, device=dev, precision="fp16", max_chunk=B).forward_tokens(img.to(dev)).cpu()
    print(f"fp16 448^2 B=16 (fused block stages): rel-L2 = {rel_l2(auto, want):.3e}, max|err| = {(auto - want).abs().max().item():.3e}")
    assert rel_l2(auto, want) < 1.2e-3 and (auto - want).abs().max().item() < 1.5e-2
    bb = VitBackbone(sd, 448, 8, 6, device=dev, precision="fp16", max_chunk=B, fuse_mlp=False, fuse_qkv=False)
    got = bb.forward_tokens(img.to(dev)).cpu()
    print(f"fp16 448^2 B=16 (separate kernels): rel-L2 = {rel_l2(got, want):.3e}, max|err| = {(got - want).abs().max().item():.3e}")
    assert rel_l2(got, want) < 1.2e-3 and (got - want).abs().max().item() < 1.5e-2
    one = VitBackbone(sd, 448, 8, 6, device=dev, precision="fp16", max_chunk=1)
    for b in (0, 7, 15):
        assert torch.equal(one.forward_tokens(img[b:b + 1].to(dev)).cpu()[0], got[b]), b


def test_fp16_stego_extract_batch_and_per_pixel(dev, golden):
    """The extractor-level paths in fp16: STEGO head + k-means + pooling (extract_batch), uint8 ingest, per-pixel prediction."""
    from wild_visual_navigation_amd.feature_extractor import FeatureExtractor
    from wild_visual_navigation_amd.model import get_model

    S = 64
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=11, depth=2)
    head = OI.make_stego_head_state_dict(384, 90, seed=2)
    img = torch.rand(3, 3, S, S, generator=g(12))
    fe = FeatureExtractor(dev, segmentation_type="stego", feature_type="stego", input_size=S, pretrained_weights=sd, head_weights=head,
                          n_image_clusters=5, precision="fp16", flip_tta=False, cluster_resolution="patch")
    feat, seg, nseg = fe.extract_batch(img.to(dev))
    tok = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
    code = OI.stego_code_tokens(head, tok)
    assert (fe._extractor.feature_tokens.cpu() - code).abs().max().item() < 2e-2      # bf16 gate of the same test: 0.15
    assert feat.shape == (3, 5, 90) and seg.shape == (3, S, S)
    u8 = (img * 255).to(torch.uint8)
    a = fe._extractor._bb.forward_tokens(u8.to(dev))
    b = fe._extractor._bb.forward_tokens((u8.float() / 255).to(dev))
    assert torch.equal(a, b)                                                           # uint8 ingest, fp16 patches: bit-identical
    fd = FeatureExtractor(dev, segmentation_type="grid", feature_type="dino", input_size=S, pretrained_weights=sd, precision="fp16")
    from oracle import mlp as OM
    from wild_visual_navigation_amd.cfg import ExperimentParams

    params = ExperimentParams()
    params.model.simple_mlp_cfg.input_size = 384
    model = get_model(params.model).to(dev)
    model.eval()
    model.load_state_dict(OM.make_mlp_state_dict(384, seed=42), strict=False)
    trav, conf, _ = fd.predict_per_pixel(img[:1].to(dev), model)
    fb = FeatureExtractor(dev, segmentation_type="grid", feature_type="dino", input_size=S, pretrained_weights=sd, precision="bf16")
    trav_b, conf_b, _ = fb.predict_per_pixel(img[:1].to(dev), model)
    assert trav.shape == (1, S, S) and torch.isfinite(trav).all() and (trav - trav_b).abs().max().item() < 0.05
