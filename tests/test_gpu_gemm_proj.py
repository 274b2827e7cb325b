"""The W-resident projection + residual kernel (csrc/gemm_proj.hip), reached through wvn_gemm_bf16 (epilogue 4, N = K = 384, M >= 32768),
against fp32 math on the bf16-rounded inputs and against the A-stationary kernel it replaces at these sizes (wvn_debug_gemm_n384 /
smaller M go through the older kernels): ragged M, LayerScale, rows past M untouched, run-to-run identical."""
import pytest
import torch

from wild_visual_navigation_amd import _lib, ops

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("M", [32768, 32768 + 17, 3152 * 21])
def test_proj_resid_matches_reference(dev, M):
    a = torch.randn(M, 384, generator=g(1)).to(torch.bfloat16)
    w = (torch.randn(384, 384, generator=g(2)) * 0.05).to(torch.bfloat16)
    bias = torch.randn(384, generator=g(3))
    x = torch.randn(M, 384, generator=g(4))
    want = x.double() + a.double() @ w.double().T + bias.double()
    guard = torch.full((M + 64, 384), 7.0, device=dev)
    guard[:M] = x.to(dev)
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    ops.gemm_bf16(ad, wd, bd, _lib.EPI_RESID_F32, out=guard[:M])
    torch.cuda.synchronize()
    assert (guard[M:] == 7.0).all(), "rows past M were written"
    err = (guard[:M].cpu().double() - want).abs().max().item()
    assert err < 2e-4, err                      # fp32 accumulation of 384 bf16 products
    again = x.to(dev).clone()
    ops.gemm_bf16(ad, wd, bd, _lib.EPI_RESID_F32, out=again)
    assert torch.equal(again, guard[:M])
    # the row-panel kernel on the same operands: same products, another summation order
    other = x.to(dev).clone()
    _lib.check(_lib.lib().wvn_debug_gemm_n384(ad.data_ptr(), 384, wd.data_ptr(), 384, bd.data_ptr(), other.data_ptr(), 384, M, 384,
                                              torch.cuda.current_stream().cuda_stream))
    assert (other - again).abs().max().item() < 1e-4


def test_proj_resid_in_the_backbone_with_layerscale(dev):
    """DINOv2-style LayerScale on the projection (ls1) at a size that takes the new kernel: ViT-S/14 with 6 heads"""
    from wild_visual_navigation_amd.backbone import VitBackbone, synthetic_vit_state_dict
    sd = synthetic_vit_state_dict("vit_small", 14, pretrain_grid=37, seed=0, dinov2=True, depth=2)
    img = torch.rand(24, 3, 518, 518, generator=g(5)).to(dev)   # 24 x 1376 rows = 33,024
    a = VitBackbone(sd, 518, 14, 6, device=dev, precision="bf16", max_chunk=24).forward_tokens(img)
    b = VitBackbone(sd, 518, 14, 6, device=dev, precision="bf16", max_chunk=8).forward_tokens(img)     # 11,008 rows: the older kernels
    assert torch.isfinite(a).all()
    rel = ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel < 1e-2, rel
