"""The split-operand block kernels behind precisions "mixed" / "exact" from 8192 rows on (round 4): the A-stationary K = 384 kernel
(csrc/gemm_a384_x3.hip), the row-panel N = 384 kernel (csrc/gemm_n384_x3.hip) and the fragment-major hand-over of the hidden
activation between them.  Every row is checked (the first version of the row-panel epilogue corrupted rows 2 and 3 of every 32 -- the
gfx950 store hazard of tests/test_isa_hazards.py -- and a strided sample of rows never saw it)."""
import os

import pytest
import torch

from oracle import interfaces as OI, vit as OV
from wild_visual_navigation_amd import _lib
from wild_visual_navigation_amd.backbone import VitBackbone, pack_fc2_fragment_major, split_planes

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("M", [12608, 8192 + 48, 128 * 70])
@pytest.mark.parametrize("K", [384, 1536])
def test_row_panel_residual_update_every_row(dev, M, K):
    """x += A W^T + b (* ls) with A / W as hi + lo planes; partial last row blocks, guard rows behind M stay untouched."""
    lib = _lib.lib()
    a = torch.randn(M, K, generator=g(M + K)).to(dev)
    w = (torch.randn(384, K, generator=g(1)) * 0.03).to(dev)
    bias, ls = (torch.randn(384, generator=g(2)) * 0.1).to(dev), (0.5 + torch.rand(384, generator=g(3))).to(dev)
    ap, wp = split_planes(a), split_planes(w)
    x0 = torch.randn(M + 64, 384, generator=g(4)).to(dev)
    c = x0.clone()
    _lib.check(lib.wvn_debug_gemm_n384_x3(ap[0].data_ptr(), ap[1].data_ptr(), K, wp[0].data_ptr(), wp[1].data_ptr(), bias.data_ptr(), ls.data_ptr(),
                                          c.data_ptr(), 384, M, K, 0, _lib.stream()), "n384_x3")
    want = x0[:M].double() + (a.double() @ w.double().T + bias.double()) * ls.double()
    assert (c[:M].double() - want).abs().max().item() < 1e-4
    assert torch.equal(c[M:], x0[M:])


@pytest.mark.parametrize("M", [12608, 8192 + 80])
@pytest.mark.parametrize("N,epi", [(1536, 1), (384, 4), (768, 1)])
def test_a_stationary_k384_every_row(dev, M, N, epi):
    lib = _lib.lib()
    a = torch.randn(M, 384, generator=g(M + N)).to(dev)
    w = (torch.randn(N, 384, generator=g(1)) * 0.05).to(dev)
    bias = (torch.randn(N, generator=g(2)) * 0.1).to(dev)
    ap, wp = split_planes(a), split_planes(w)
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        c = torch.zeros(2, M + 32, N, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.wvn_debug_gemm_a384_x3(ap[0].data_ptr(), ap[1].data_ptr(), 384, wp[0].data_ptr(), wp[1].data_ptr(), bias.data_ptr(), c[0].data_ptr(),
                                              c[1].data_ptr(), N, M, N, epi, 0, _lib.stream()), "a384_x3")
        got = c[0, :M].double() + c[1, :M].double()
        assert (got - torch.nn.functional.gelu(ref)).abs().max().item() < 1e-4
        assert float(c[:, M:].abs().max()) == 0.0
    else:
        x0 = torch.randn(M + 32, N, generator=g(5)).to(dev)
        c = x0.clone()
        _lib.check(lib.wvn_debug_gemm_a384_x3(ap[0].data_ptr(), ap[1].data_ptr(), 384, wp[0].data_ptr(), wp[1].data_ptr(), bias.data_ptr(), c.data_ptr(), 0, N,
                                              M, N, epi, 0, _lib.stream()), "a384_x3")
        assert (c[:M].double() - (x0[:M].double() + ref)).abs().max().item() < 1e-4
        assert torch.equal(c[M:], x0[M:])


@pytest.fixture
def n384_pair(request):
    """The two forms of the fragment-major row-panel kernel: a wave pair per 32 rows (two waves per SIMD, the default) / one wave per SIMD."""
    _lib.lib().wvn_debug_n384_pair(request.param)
    yield request.param
    _lib.lib().wvn_debug_n384_pair(1)


@pytest.mark.parametrize("n384_pair", [1, 0], indirect=True, ids=["wave-pair", "one-wave-per-simd"])
@pytest.mark.parametrize("M", [12608, 128 * 65 + 16])
def test_fragment_major_mlp_every_row(dev, M, n384_pair):
    """fc1 writes the hidden activation as MFMA operand fragments (EPI_GELU_FRAG), fc2 consumes them (AFRAG) with the packed weight."""
    lib = _lib.lib()
    F = 1536
    a = torch.randn(M, 384, generator=g(M)).to(dev)
    w1, b1 = (torch.randn(F, 384, generator=g(1)) * 0.05).to(dev), (torch.randn(F, generator=g(2)) * 0.1).to(dev)
    w2, b2 = (torch.randn(384, F, generator=g(3)) * 0.03).to(dev), (torch.randn(384, generator=g(4)) * 0.1).to(dev)
    ap, w1p, w2p = split_planes(a), split_planes(w1), pack_fc2_fragment_major(w2)
    Mp = (M + 31) // 32 * 32
    hid = torch.zeros(2, Mp * F, dtype=torch.bfloat16, device=dev)
    x0 = torch.randn(M + 64, 384, generator=g(5)).to(dev)
    x = x0.clone()
    _lib.check(lib.wvn_debug_mlp_x3_frag(ap[0].data_ptr(), ap[1].data_ptr(), w1p[0].data_ptr(), w1p[1].data_ptr(), b1.data_ptr(), hid[0].data_ptr(), hid[1].data_ptr(),
                                         w2p.data_ptr(), b2.data_ptr(), x.data_ptr(), M, F, 0, 0, _lib.stream()), "mlp_x3_frag")
    want = x0[:M].double() + torch.nn.functional.gelu(a.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()
    assert (x[:M].double() - want).abs().max().item() < 2e-4
    assert torch.equal(x[M:], x0[M:])
    if n384_pair:   # the same products in the same order per output element: the two forms agree bit for bit
        lib.wvn_debug_n384_pair(0)
        x1 = x0.clone()
        _lib.check(lib.wvn_debug_mlp_x3_frag(ap[0].data_ptr(), ap[1].data_ptr(), w1p[0].data_ptr(), w1p[1].data_ptr(), b1.data_ptr(), hid[0].data_ptr(), hid[1].data_ptr(),
                                             w2p.data_ptr(), b2.data_ptr(), x1.data_ptr(), M, F, 0, 0, _lib.stream()), "mlp_x3_frag")
        assert torch.equal(x, x1)
        lib.wvn_debug_n384_pair(1)


@pytest.mark.parametrize("precision", ["mixed", "exact"])
def test_fast_block_kernels_inside_the_vit(dev, precision):
    """4 frames at 448^2 (12608 rows: a partial last row block) through 2 blocks with the fast kernels on and off (WVN_VIT_NO_A384_X3):
    within fp32 summation-order noise of each other and inside the mode's gate against the CPU oracle."""
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=0, depth=2)
    img = torch.rand(4, 3, 448, 448, generator=g(1))
    want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
    os.environ.pop("WVN_NO_A384_X3", None)
    a = VitBackbone(sd, 448, 8, 6, device=dev, precision=precision, max_chunk=4).forward_tokens(img.to(dev)).cpu()
    os.environ["WVN_NO_A384_X3"] = "1"
    try:
        b = VitBackbone(sd, 448, 8, 6, device=dev, precision=precision, max_chunk=4).forward_tokens(img.to(dev)).cpu()
    finally:
        os.environ.pop("WVN_NO_A384_X3", None)
    # ("mixed": a = the MX kernels (round 6), b = the tiled bf16 x 3 kernel: the correction terms' e5m2 operands separate them by ~1.5e-4)
    assert (a - b).abs().max().item() < (3e-4 if precision == "mixed" else 2e-4)
    assert (a - want).abs().max().item() < (5e-4 if precision == "mixed" else 1e-4)


@pytest.mark.parametrize("precision", ["mixed", "exact"])
def test_layernorm_across_kernel_boundaries(dev, precision):
    """The split-operand block kernels hand the LayerNorm across their boundaries: the row-panel kernels (projection, fc2) leave {mean, rstd}
    of the rows they update, the A-stationary kernels (fc1, the next block's QKV) normalise as they load.  4 frames at 448^2 (a partial
    last row block) through 3 blocks with the hand-over on and off (debug bit 512: the LayerNorm kernel and the plane round trip): the
    one-pass statistics and the different association of the affine step must stay inside fp32 noise of the two-pass kernel."""
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=0, depth=3)
    img = torch.rand(4, 3, 448, 448, generator=g(2))
    want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
    os.environ.pop("WVN_X3_DEBUG_BITS", None)
    # (the hand-over of the bf16 x 3 kernels is what this test is about: "mixed" defaults to the MX kernels since round 6, which exist only WITH the
    #  hand-over -- tests/test_gpu_mx.py covers their LayerNorm-on-load -- so both runs take the bf16 x 3 route here)
    os.environ["WVN_NO_MX"] = "1"
    try:
        a = VitBackbone(sd, 448, 8, 6, device=dev, precision=precision, max_chunk=4).forward_tokens(img.to(dev)).cpu()
        os.environ["WVN_X3_DEBUG_BITS"] = "512"
        b = VitBackbone(sd, 448, 8, 6, device=dev, precision=precision, max_chunk=4).forward_tokens(img.to(dev)).cpu()
    finally:
        os.environ.pop("WVN_X3_DEBUG_BITS", None)
        os.environ.pop("WVN_NO_MX", None)
    assert not torch.equal(a, b)                      # (the two routes really are different code)
    assert (a - b).abs().max().item() < 5e-5
    assert (a - want).abs().max().item() < (5e-4 if precision == "mixed" else 1e-4)
