"""The lazy-max form of the pre-scaled bf16 attention kernel (csrc/attention_bf16.hip, LAZY): no per-tile row max; the row sums
raise the alarm and only then the tile is redone exactly.  Same tolerance as the exact-max form against the fp64 reference, on
inputs that keep the alarm silent, that trip it on many tiles (scores climbing along the sequence), that overflow exp2 to inf
inside one tile, and on the masked tail tile; and the two forms must agree with each other."""
import pytest
import torch

from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd._lib import check, lib, ptr, stream

pytestmark = pytest.mark.gpu
C = 0.125 * 1.4426950408889634


def bf(t):
    return t.to(torch.bfloat16)


def run(dev, q_in, k, v, ntok, variant):
    B, h = q_in.shape[:2]
    npad = (ntok + 127) // 128 * 128

    def pad(t, fill):
        out = torch.full((B, h, npad, 64), fill, dtype=t.dtype)
        out[:, :, :ntok] = t
        return out

    vt = pad(v, 1e3).transpose(-1, -2)[..., ops.vt_token_order(npad)].contiguous().to(dev)
    qd, kd = pad(q_in, 50.0).to(dev), pad(k, -1e3).to(dev)
    out = torch.empty(B * ntok, h * 64, dtype=torch.bfloat16, device=dev)
    lib().wvn_debug_attention_variant(variant)
    try:
        check(lib().wvn_attention_bf16(ptr(qd), ptr(kd), ptr(vt), ptr(out), B, h, ntok, npad, 0.0, stream()))
        torch.cuda.synchronize()
    finally:
        lib().wvn_debug_attention_variant(-1)
    return out.float().cpu()


def reference(q_in, k, v):
    B, h, ntok, _ = q_in.shape
    s = (q_in.double() / C) @ k.double().transpose(-1, -2) * 0.125
    return (torch.softmax(s, dim=-1) @ v.double()).permute(0, 2, 1, 3).reshape(B * ntok, h * 64)


@pytest.mark.parametrize("case", ["plain", "climbing", "overflow", "tail_spike", "negative"])
@pytest.mark.parametrize("ntok", [65, 197, 785, 3137])
def test_lazy_matches_reference_and_exact_form(dev, ntok, case):
    B, h = (1, 2) if ntok > 1000 else (2, 3)
    gen = torch.Generator().manual_seed(ntok * 7 + len(case))
    q, k, v = (torch.randn(B, h, ntok, 64, generator=gen) for _ in range(3))
    if case == "climbing":      # key norms grow along the sequence: the running max moves on most tiles
        k = k * torch.linspace(0.2, 6.0, ntok)[None, None, :, None]
    elif case == "overflow":    # one key 30x larger in the middle: its score exceeds the running max by far more than 2^7 exp2-units
        k[:, :, ntok // 2] = q[:, :, ntok // 3] * 30.0
    elif case == "tail_spike":  # the largest score sits in the masked last tile
        k[:, :, ntok - 1] = k[:, :, ntok - 1] * 8.0
    elif case == "negative":    # all scores far below zero on the first tile, rising later
        k[:, :, :64] = -q[:, :, :1] * 3.0
    q_in, k, v = bf(q * C), bf(k), bf(v)
    ref = reference(q_in.float(), k.float(), v.float())
    lazy = run(dev, q_in, k, v, ntok, 1)
    exact = run(dev, q_in, k, v, ntok, 0)
    assert torch.isfinite(lazy).all()
    assert (lazy.double() - ref).abs().max().item() < 2e-2
    assert (exact.double() - ref).abs().max().item() < 2e-2
    # same math up to where the (equally valid) running max sits: bf16 rounding of P differs in the last place at most
    assert (lazy - exact).abs().max().item() < 2e-2


def test_lazy_uniform_v_gives_exact_ones(dev):
    """softmax weights sum to one within rounding whatever scale P is carried at"""
    ntok, B, h = 785, 2, 3
    gen = torch.Generator().manual_seed(5)
    q, k = (torch.randn(B, h, ntok, 64, generator=gen) for _ in range(2))
    k = k * torch.linspace(0.2, 5.0, ntok)[None, None, :, None]
    out = run(dev, bf(q * C), bf(k), torch.ones(B, h, ntok, 64, dtype=torch.bfloat16), ntok, 1)
    assert (out - 1.0).abs().max().item() < 1e-2
