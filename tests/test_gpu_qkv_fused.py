"""LayerNorm + QKV projection in one kernel (csrc/qkv_fused.hip), through the C-ABI, against fp32 math with the same bf16 rounding
points (normalised rows rounded to bf16, fp32 accumulation, q scaled before its rounding), in the layouts the attention kernel
reads: q / k [frames*6][npad][64], v^T [frames*6][64][npad] with the tokens of every aligned 16 permuted (bits 2 <-> 3).  Shapes
cover one partial row block, several row blocks per workgroup, and more row blocks than CUs with a thin last round (split by
column tiles over the workgroups); the repeat screens the DMA ring / counted-vmcnt pipeline for races."""
import pytest
import torch
import torch.nn.functional as F

from wild_visual_navigation_amd import ops

pytestmark = pytest.mark.gpu
QS = 0.125 * 1.4426950408889634


def g(seed):
    return torch.Generator().manual_seed(seed)


def reference(x, gam, bet, w, bias, frames, ntok_s, npad):
    xn = F.layer_norm(x, (384,), gam, bet, 1e-6).to(torch.bfloat16).float()
    y = xn @ w.float().T + bias
    y[:, :384] *= QS
    y = y.to(torch.bfloat16).float().view(frames, ntok_s, 3, 6, 64)
    q = torch.zeros(frames, 6, npad, 64)
    k = torch.zeros(frames, 6, npad, 64)
    v = torch.zeros(frames, 6, npad, 64)
    q[:, :, :ntok_s] = y[:, :, 0].permute(0, 2, 1, 3)
    k[:, :, :ntok_s] = y[:, :, 1].permute(0, 2, 1, 3)
    v[:, :, :ntok_s] = y[:, :, 2].permute(0, 2, 1, 3)
    vt = v.transpose(-1, -2)[..., ops.vt_token_order(npad)]
    return q.reshape(frames * 6, npad, 64), k.reshape(frames * 6, npad, 64), vt.reshape(frames * 6, 64, npad)


@pytest.mark.parametrize("frames,ntok_s", [(1, 16), (2, 208), (3, 3152), (5, 800), (21, 3152)])
def test_qkv_fused_matches_reference(dev, frames, ntok_s):
    M, npad = frames * ntok_s, (ntok_s + 127) // 128 * 128
    x = torch.randn(M, 384, generator=g(1)) * 2.0 + 0.3
    x[:, 7] *= 12.0                                  # an outlier channel, as ViT residual streams have
    gam = torch.rand(384, generator=g(2)) + 0.5
    bet = torch.randn(384, generator=g(3)) * 0.2
    w = (torch.randn(1152, 384, generator=g(4)) * 0.05).to(torch.bfloat16)
    bias = torch.randn(1152, generator=g(5)) * 0.3
    want = reference(x, gam, bet, w, bias, frames, ntok_s, npad)
    got = ops.qkv_fused(x.to(dev), (gam.to(dev), bet.to(dev), 1e-6), w.to(dev), bias.to(dev), frames, ntok_s, npad, QS)
    torch.cuda.synchronize()
    for name, a, b in zip("q k vt".split(), got, want):
        a = a.float().cpu()
        # one bf16 rounding of the result, plus the few normalised inputs that round the other way because the LayerNorm
        # statistics are summed in another order (1 bf16 ulp of an input times |w| ~ 0.05)
        err = (a - b).abs()
        assert (err <= 1.6e-2 * b.abs() + 2e-2).all(), (name, err.max().item())
        assert err.mean().item() <= 2e-3, name
        assert torch.isfinite(a).all()
    again = ops.qkv_fused(x.to(dev), (gam.to(dev), bet.to(dev), 1e-6), w.to(dev), bias.to(dev), frames, ntok_s, npad, QS)
    for a, b in zip(got, again):
        assert torch.equal(a, b), "run-to-run difference (pipeline race)"


def test_qkv_fused_argument_checks(dev):
    from wild_visual_navigation_amd import _lib
    x = torch.randn(24, 384, device=dev)
    gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
    w = torch.zeros(1152, 384, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.WvnError):
        ops.qkv_fused(x, (gam, bet, 1e-6), w, None, 1, 24, 128)     # ntok_s % 16 != 0
