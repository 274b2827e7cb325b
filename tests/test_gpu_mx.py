"""The MX correction terms of the <= 1e-3 mode (round 6): hi * hi on fp16 MFMAs + the two correction products on scaled e5m2 MFMAs of K = 64
(csrc/gemm_n384_x3.hip: gemm_n384_mx_pair_kernel; csrc/gemm_a384_x3.hip MX instantiations).  Every row is checked twice: against the
float64 statement of the SAME operand roundings (backbone.mx_matmul_reference: only the fp32 accumulation order separates the two -- a
layout or scale mistake shows as an O(1e-3) error) and against the plain float64 product (the precision the mode is for)."""
import pytest
import torch

from wild_visual_navigation_amd import _lib
from wild_visual_navigation_amd.backbone import mx_fragments, mx_matmul_reference, pack_n384_mx

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("M", [12608, 8192 + 48, 128 * 70])
@pytest.mark.parametrize("K", [384, 1536])
def test_row_panel_mx_every_row(dev, M, K):
    """x += A W^T + b (* ls) with the MX operand planes; partial last row blocks, guard rows behind M stay untouched."""
    lib = _lib.lib()
    a = torch.randn(M, K, generator=g(M + K))
    a[::7, ::5] *= 30.0                                   # a spread of magnitudes inside one row (the residues must keep their own exponents)
    w = torch.randn(384, K, generator=g(1)) * 0.03
    bias, ls = (torch.randn(384, generator=g(2)) * 0.1).to(dev), (0.5 + torch.rand(384, generator=g(3))).to(dev)
    ah, al8, ah8 = mx_fragments(a.to(dev))
    planes = torch.cat([ah.view(torch.uint8).reshape(-1), al8.reshape(-1), ah8.reshape(-1)])   # one allocation: hi | l8 | h8
    n_h = ah.numel() * 2
    wp = pack_n384_mx(w.to(dev))
    x0 = torch.randn(M + 64, 384, generator=g(4)).to(dev)
    c = x0.clone()
    _lib.check(lib.wvn_debug_gemm_n384_mx(planes.data_ptr(), planes.data_ptr() + n_h, planes.data_ptr() + n_h + al8.numel(), wp.data_ptr(), bias.data_ptr(),
                                          ls.data_ptr(), c.data_ptr(), 384, M, K, 0, _lib.stream()), "n384_mx")
    got = c[:M].double().cpu()
    same = x0[:M].double().cpu() + (mx_matmul_reference(a, w) + bias.double().cpu()) * ls.double().cpu()
    want = x0[:M].double().cpu() + (a.double() @ w.double().T + bias.double().cpu()) * ls.double().cpu()
    scale = (a.double().abs() @ w.double().abs().T).max().item()
    assert (got - same).abs().max().item() < 3e-6 * scale, "the kernel does not compute the MX statement"
    assert (got - want).abs().max().item() < 4e-5 * scale      # (measured 1.7e-5: 2^-12 / sqrt 3 x the e5m2 rounding of the correction operands)
    assert torch.equal(c[M:], x0[M:])
