"""The row-panel N = 384 residual GEMM (csrc/gemm_n384.hip: 32 x 384 fp32 accumulators per wave, A and W streamed by DMA
through a swizzled 3-stage LDS ring), through the C-ABI, against fp32 math on the bf16-rounded inputs; plus a repeat
screen for ring / counted-vmcnt races (bit-identical results required)."""
import math

import pytest
import torch

from wild_visual_navigation_amd import _lib, ops

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def n384_forced(a, w, bias, out):
    """The row-panel kernel itself on every row block (wvn_gemm_bf16 only dispatches to it from ~0.75 x #CU row blocks on)."""
    h = _lib.lib()
    rc = h.wvn_debug_gemm_n384(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else 0,
                               out.data_ptr(), out.stride(0), a.shape[0], a.shape[1], _lib.stream())
    _lib.check(rc, "wvn_debug_gemm_n384")


@pytest.mark.parametrize("M", [8, 256, 300, 777])
@pytest.mark.parametrize("K", [1536, 768, 448])
def test_n384_residual_update(dev, M, K):
    a = torch.randn(M, K, generator=g(1)).to(torch.bfloat16)
    w = (torch.randn(384, K, generator=g(2)) * 0.05).to(torch.bfloat16)
    bias = torch.randn(384, generator=g(3))
    ref = a.float() @ w.float().T + bias
    c0 = torch.randn(M, 384, generator=g(4))
    cd, cf = c0.clone().to(dev), c0.clone().to(dev)
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    ops.gemm_bf16(ad, wd, bd, _lib.EPI_RESID_F32, out=cd)    # dispatcher: the tiled kernel at these sizes
    n384_forced(ad, wd, bd, cf)                              # the row-panel kernel
    err = (cf.cpu() - (c0 + ref)).abs().max().item()
    assert err < 2e-5 * ref.abs().max().item() * math.sqrt(K), err
    assert torch.equal(cd, cf)                               # same association in both kernels: same bits


def test_n384_transpose_detecting_strided(dev):
    """C[m][n] += (m % 100) - n exactly (small integers are exact in bf16); A is a strided view, C has ldc > 384."""
    M, K = 520, 1536
    a = torch.zeros(M, K)
    a[:, 7] = torch.arange(M).float() % 100
    a[:, 1300] = 1.0
    w = torch.zeros(384, K)
    w[:, 7] = 1.0
    w[:, 1300] = -(torch.arange(384).float() % 128) - 3 * (torch.arange(384) // 128).float()  # <= 256: exact in bf16, column-unique
    big = torch.zeros(M, 2 * K, dtype=torch.bfloat16, device=dev)
    big[:, K:] = a.to(torch.bfloat16).to(dev)
    out = torch.ones(M, 384 + 4, device=dev)
    n384_forced(big[:, K:], w.to(torch.bfloat16).to(dev), None, out[:, :384])
    want = 1.0 + (torch.arange(M).float() % 100)[:, None] + w[:, 1300][None]
    assert torch.equal(out[:, :384].cpu(), want) and torch.equal(out[:, 384:].cpu(), torch.ones(M, 4))


def test_n384_repeatable_under_load(dev):
    M, K = 256 * 400 + 40, 1536
    a = torch.randn(M, K, generator=g(5)).to(torch.bfloat16).to(dev)
    w = (torch.randn(384, K, generator=g(6)) * 0.03).to(torch.bfloat16).to(dev)
    bias = torch.randn(384, generator=g(7)).to(dev)
    c0 = torch.randn(M, 384, generator=g(8)).to(dev)
    first = c0.clone()
    ops.gemm_bf16(a, w, bias, _lib.EPI_RESID_F32, out=first)
    ref = a[-3000:].float() @ w.float().T + bias + c0[-3000:]
    assert (first[-3000:] - ref).abs().max().item() < 2e-5 * ref.abs().max().item() * math.sqrt(K)
    for _ in range(10):
        again = c0.clone()
        ops.gemm_bf16(a, w, bias, _lib.EPI_RESID_F32, out=again)
        assert torch.equal(again, first)


def test_n384_rows_do_not_depend_on_their_position(dev):
    """With more row blocks than CUs the thin last round is computed by the tiled kernel: a row must get the same bits
    there as in the row-panel kernel (same k order, same (acc + bias) + C association) -- batch-slot invariance."""
    K = 1536
    M = 256 * 260 + 24  # 261 row blocks on 256 CUs: 256 in the row-panel kernel, 5 (the last one ragged) in the tiled kernel
    a = torch.randn(M, K, generator=g(11)).to(torch.bfloat16).to(dev)
    w = (torch.randn(384, K, generator=g(12)) * 0.03).to(torch.bfloat16).to(dev)
    bias = torch.randn(384, generator=g(13)).to(dev)
    c0 = torch.randn(M, 384, generator=g(14)).to(dev)
    big = c0.clone()
    ops.gemm_bf16(a, w, bias, _lib.EPI_RESID_F32, out=big)
    tail = slice(256 * 256, M)  # rows the big launch left to the tiled kernel
    small = c0[tail].clone()     # the same rows as a small problem of their own: all in the row-panel kernel
    n384_forced(a[tail], w, bias, small)
    assert torch.equal(big[tail], small)
    head = c0[:300].clone()
    ops.gemm_bf16(a[:300], w, bias, _lib.EPI_RESID_F32, out=head)
    assert torch.equal(big[:300], head)
