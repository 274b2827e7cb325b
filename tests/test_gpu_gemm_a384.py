"""The A-stationary K = 384 GEMM (csrc/gemm_a384.hip: A rows in registers, W through a direct-to-LDS ring,
software-pipelined ping-pong epilogue), called through the C-ABI, against fp32 math on the bf16-rounded
inputs.  Shapes cover partial row blocks (M % 256 != 0, M < 32), one and many column tiles, and every
epilogue; the repeat test screens for pipeline races (ring reuse, counted vmcnt, wave-private staging):
a race shows up as run-to-run differences or as wrong tiles."""
import math

import pytest
import torch
import torch.nn.functional as F

from wild_visual_navigation_amd import _lib, ops

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("M", [8, 256, 300, 777])
@pytest.mark.parametrize("N", [64, 384, 1536])
def test_a384_epilogues(dev, M, N):
    K = 384
    a = torch.randn(M, K, generator=g(1)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g(2)) * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, generator=g(3))
    ref = a.float() @ w.float().T + bias
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    out = ops.gemm_bf16(ad, wd, bd, _lib.EPI_BF16).float().cpu()
    # one bf16 rounding of the result (2^-9 relative) + fp32 accumulation noise
    assert ((out - ref).abs() <= 4e-3 * ref.abs() + 1e-3).all(), "bf16-out epilogue"
    out = ops.gemm_bf16(ad, wd, bd, _lib.EPI_GELU_BF16).float().cpu()
    want = F.gelu(ref)
    # GELU in the bf16 path: within 0.25 bf16 ulp of the exact erf GELU (next test), then one bf16 rounding; the
    # absolute term covers the fp32 accumulation noise of ref through the GELU slope
    assert ((out - want).abs() <= 4e-3 * want.abs() + 1e-3).all(), "gelu epilogue"
    out = ops.gemm_bf16(ad, wd, bd, _lib.EPI_RELU_BF16).float().cpu()
    assert ((out - F.relu(ref)).abs() <= 4e-3 * ref.abs() + 1e-3).all(), "relu epilogue"
    c0 = torch.randn(M, N, generator=g(4))
    cd = c0.clone().to(dev)
    ops.gemm_bf16(ad, wd, bd, _lib.EPI_RESID_F32, out=cd)
    assert (cd.cpu() - (c0 + ref)).abs().max().item() < 2e-5 * ref.abs().max().item() * math.sqrt(K), "residual epilogue"


def test_a384_gelu_is_the_erf_gelu_to_one_bf16_ulp(dev):
    """The fc1 epilogue's GELU against torch's exact (erf) GELU on pre-activations that reach the kernel EXACTLY (a one-hot
    weight copies bf16 inputs into the accumulator): the stored bf16 value is the correctly rounded erf GELU or its neighbour,
    over the whole range including the negative tail where a tanh-form GELU is tens of ulps off."""
    M, N, K = 4096, 64, 384
    x = torch.cat([torch.linspace(-8, 8, M - 512), torch.randn(512, generator=g(9)) * 2]).to(torch.bfloat16)
    a = torch.zeros(M, K, dtype=torch.bfloat16)
    a[:, 7] = x
    w = torch.zeros(N, K, dtype=torch.bfloat16)
    w[:, 7] = 1.0
    out = ops.gemm_bf16(a.to(dev), w.to(dev), None, _lib.EPI_GELU_BF16).cpu()
    want = F.gelu(x.double())
    exact_bf16 = want.float().to(torch.bfloat16)
    ulp = torch.ldexp(torch.ones_like(want), torch.floor(torch.log2(want.abs().clamp_min(1e-300))).int() - 7).clamp_min(2.0 ** -20)
    err = (out[:, 0].double() - want).abs() / ulp
    assert err.max().item() <= 1.0, err.max().item()
    # a deviation of up to 0.25 ulp before the rounding moves at most ~1/4 of the values to the neighbouring bf16
    assert (out[:, 0] == exact_bf16).float().mean().item() > 0.7
    assert torch.equal(out, out[:, :1].expand(M, N))


def test_a384_transpose_detecting_strided(dev):
    """C[m][n] = m - n exactly (small integers are exact in bf16): catches row/column swaps, wrong column
    tiles, wrong k-slot pairing; A and C are strided views."""
    M, N, K = 520, 192, 384
    a = torch.zeros(M, K)
    a[:, 5] = torch.arange(M).float() % 200
    a[:, 300] = 1.0
    w = torch.zeros(N, K)
    w[:, 5] = 1.0
    w[:, 300] = -torch.arange(N).float()
    big = torch.zeros(M, 2 * K, dtype=torch.bfloat16, device=dev)
    big[:, K:] = a.to(torch.bfloat16).to(dev)
    want = (torch.arange(M).float() % 200)[:, None] - torch.arange(N).float()[None]
    out = torch.zeros(M, N + 8, dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(big[:, K:], w.to(torch.bfloat16).to(dev), None, _lib.EPI_BF16, out=out[:, :N])
    assert torch.equal(out[:, :N].float().cpu(), want) and float(out[:, N:].abs().sum()) == 0.0


def test_a384_repeatable_under_load(dev):
    """Many workgroups, many column tiles, 20 repeats: results must be bit-identical and correct."""
    M, N, K = 256 * 300, 1152, 384
    a = torch.randn(M, K, generator=g(5)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g(6)) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g(7)).to(dev)
    first = ops.gemm_bf16(a, w, bias, _lib.EPI_BF16)
    ref = (a[:4096].float() @ w.float().T + bias)
    assert ((first[:4096].float() - ref).abs() <= 4e-3 * ref.abs() + 1e-3).all()
    ref2 = (a[-4096:].float() @ w.float().T + bias)
    assert ((first[-4096:].float() - ref2).abs() <= 4e-3 * ref2.abs() + 1e-3).all()
    for _ in range(20):
        again = ops.gemm_bf16(a, w, bias, _lib.EPI_BF16)
        assert torch.equal(again, first)
