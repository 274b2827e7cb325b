"""The XCD-aware workgroup decode of the bf16 attention kernel (taken when B*heads % 8 == 0) must give the
same answers as the plain order: compare against the fp64 reference, and against the fp32 exact kernel."""
import pytest
import torch

from wild_visual_navigation_amd import ops
from wild_visual_navigation_amd._lib import check, lib, ptr, stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,h,ntok", [(4, 2, 197), (8, 6, 130), (2, 12, 65), (16, 1, 330)])
def test_attention_bf16_xcd_mapping(dev, B, h, ntok):
    assert (B * h) % 8 == 0
    g = torch.Generator().manual_seed(B * 1000 + ntok)
    q, k, v = (torch.randn(B, h, ntok, 64, generator=g).to(torch.bfloat16) for _ in range(3))
    npad = (ntok + 127) // 128 * 128

    def pad(t):  # the contract (wvn_hip.h) is FINITE padding: large garbage must not leak into the result
        out = (torch.randn(B, h, npad, 64, generator=g) * 1e3).to(t.dtype)
        out[:, :, :ntok] = t
        return out

    qd, kd = pad(q).to(dev), pad(k).to(dev)
    vt = pad(v).transpose(-1, -2)[..., ops.vt_token_order(npad)].contiguous().to(dev)
    out = torch.empty(B * ntok, h * 64, dtype=torch.bfloat16, device=dev)
    check(lib().wvn_attention_bf16(ptr(qd), ptr(kd), ptr(vt), ptr(out), B, h, ntok, npad, 0.125, stream()))
    att = torch.softmax((q.double() @ k.double().transpose(-1, -2)) * 0.125, dim=-1) @ v.double()
    ref = att.permute(0, 2, 1, 3).reshape(B * ntok, h * 64)
    assert (out.float().cpu().double() - ref).abs().max().item() < 2e-2
    # every (frame, head) must have been computed exactly once and written to its own slot:
    # per-(b,h) error stays at round-off level (a mis-decoded block would be O(1) wrong)
    err = (out.float().cpu().double() - ref).reshape(B, ntok, h, 64).abs().amax(dim=(1, 3))
    assert err.max().item() < 2e-2
