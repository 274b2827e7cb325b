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


def _ln_stats(x, eps=1e-6):
    mean = x.double().mean(-1)
    var = x.double().var(-1, unbiased=False)
    return torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=-1).float().contiguous()


def _unfrag(h, l8, h8, M, K):
    """The inverse of backbone.mx_fragments: (h + l8 / 4096, h8) as [M][K] float64."""
    from wild_visual_navigation_amd.backbone import MX_RES_SCALE, _swap23
    R = h.shape[0]
    sw = _swap23(16)
    inv = torch.empty(16, dtype=torch.long)
    inv[sw] = torch.arange(16)
    hf = h.cpu().reshape(R, K // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(R, 32, K // 16, 16)[..., inv].reshape(R * 32, K)
    outs = []
    for b8 in (l8, h8):
        q = b8.cpu().view(torch.float8_e5m2).double().reshape(R, K // 64, 2, 2, 32, 2, 8)        # [R][c][x][hw][row][sp][j]
        q = q.permute(0, 4, 1, 2, 5, 3, 6).reshape(R, 32, K // 16, 16)[..., inv].reshape(R * 32, K)
        outs.append(q)
    return (hf.double() + outs[0] / MX_RES_SCALE)[:M], outs[1][:M]


@pytest.fixture(params=[2, 1], ids=["two-workgroups-per-cu", "one-wave-per-simd"])
def a384_form(request):
    """Both forms of the A-stationary MX kernel (csrc/gemm_a384_x3.hip: gemm_a384_mx2_kernel, the default, and gemm_a384_x3_kernel<..., MX>)."""
    lib = _lib.lib()
    lib.wvn_debug_n384_pair(32 + request.param)
    yield request.param
    lib.wvn_debug_n384_pair(32)


@pytest.mark.parametrize("M", [12608, 128 * 65 + 16, 128 * 530 + 40])
def test_mx_block_mlp_every_row(dev, M, a384_form):
    """LayerNorm-on-load fc1 + GELU writes the MX operand planes (fp16 fragments, l8), the MX row-panel kernel consumes them.  (The largest M
    gives the two-workgroups-per-CU form more than one row block per workgroup and row blocks shared between workgroups.)"""
    from wild_visual_navigation_amd.backbone import pack_a384_mx
    lib = _lib.lib()
    F = 1536
    x = torch.randn(M, 384, generator=g(M)) * 1.7 + 0.3
    gam, bet = 1.0 + 0.1 * torch.randn(384, generator=g(7)), 0.05 * torch.randn(384, generator=g(8))
    w1, b1 = torch.randn(F, 384, generator=g(1)) * 0.05, torch.randn(F, generator=g(2)) * 0.1
    w2, b2 = torch.randn(384, F, generator=g(3)) * 0.03, torch.randn(384, generator=g(4)) * 0.1
    st = _ln_stats(x)
    Mp = (M + 31) // 32 * 32
    hid = torch.zeros(Mp * F * 4, dtype=torch.uint8, device=dev)
    n_h, n_8 = Mp * F * 2, Mp * F
    x0 = torch.randn(M + 64, 384, generator=g(5)).to(dev)
    xo = x0.clone()
    d = lambda t: t.to(dev).contiguous()   # noqa: E731
    xd, sd_, gd, bd, w1p, b1d, w2p, b2d = d(x), d(st), d(gam), d(bet), pack_a384_mx(d(w1)), d(b1), pack_n384_mx(d(w2)), d(b2)
    _lib.check(lib.wvn_debug_mlp_mx(xd.data_ptr(), 384, sd_.data_ptr(), gd.data_ptr(), bd.data_ptr(), w1p.data_ptr(), b1d.data_ptr(), hid.data_ptr(),
                                    hid.data_ptr() + n_h, hid.data_ptr() + n_h + n_8, w2p.data_ptr(), b2d.data_ptr(), xo.data_ptr(), M, F, 0, 0, _lib.stream()), "mlp_mx")
    y = torch.nn.functional.layer_norm(x.double(), (384,), gam.double(), bet.double(), eps=1e-6)
    hidden = torch.nn.functional.gelu(y @ w1.double().T + b1.double())
    R = Mp // 32
    hv, h8v = _unfrag(hid[:n_h].view(torch.float16).reshape(R, F // 16, 64, 8), hid[n_h:n_h + n_8].reshape(R, F // 64, 2, 64, 16),
                      hid[n_h + n_8:].reshape(R, F // 64, 2, 64, 16), M, F)
    assert (hv - hidden).abs().max().item() < 2e-4                       # fc1 through the MX products, hidden = h + l8 / 4096
    assert float(h8v.abs().max()) == 0.0                                 # (no h8 plane is written: the consumer derives e5m2(h) in registers)
    want = x0[:M].double().cpu() + hidden @ w2.double().T + b2.double()
    assert (xo[:M].double().cpu() - want).abs().max().item() < 4e-4
    assert torch.equal(xo[M:], x0[M:])


@pytest.mark.parametrize("B", [4, 22])
def test_mx_qkv(dev, B, a384_form):
    """LayerNorm-on-load + q | k | v^T with the MX products: fp16 planes in the attention kernel's layouts (q pre-scaled, two planes)."""
    from wild_visual_navigation_amd.backbone import pack_a384_mx
    lib = _lib.lib()
    ntok, heads = 3137, 6
    ntok_s, npad = 3152, 3200
    M = B * ntok_s
    x = torch.randn(M, 384, generator=g(11)) * 1.3
    gam, bet = 1.0 + 0.1 * torch.randn(384, generator=g(7)), 0.05 * torch.randn(384, generator=g(8))
    w, b = torch.randn(1152, 384, generator=g(1)) * 0.06, torch.randn(1152, generator=g(2)) * 0.02
    qs = 0.125 * 1.4426950408889634
    d = lambda t: t.to(dev).contiguous()   # noqa: E731
    per = B * heads * npad * 64
    buf = torch.zeros(4 * per, dtype=torch.float16, device=dev)          # (one allocation: the kernel addresses q | k | v^T through one buffer descriptor)
    q, k, vt = buf[:2 * per].view(2, B, heads, npad, 64), buf[2 * per:3 * per].view(B, heads, npad, 64), buf[3 * per:].view(B, heads, 64, npad)
    xd, sd_, gd, bd, wp, bb = d(x), d(_ln_stats(x)), d(gam), d(bet), pack_a384_mx(d(w)), d(b)
    _lib.check(lib.wvn_debug_qkv_mx(xd.data_ptr(), 384, sd_.data_ptr(), gd.data_ptr(), bd.data_ptr(), wp.data_ptr(), bb.data_ptr(), q[0].data_ptr(), q[1].data_ptr(),
                                    k.data_ptr(), vt.data_ptr(), heads, npad, ntok_s, qs, M, 0, _lib.stream()), "qkv_mx")
    y = torch.nn.functional.layer_norm(x.double(), (384,), gam.double(), bet.double(), eps=1e-6)
    ref = (y @ w.double().T + b.double()).reshape(B, ntok_s, 3, heads, 64)
    qg = (q[0].double() + q[1].double()).cpu()[:, :, :ntok_s]
    assert (qg - ref[:, :, 0].permute(0, 2, 1, 3) * qs).abs().max().item() < 2e-4
    kg = k.double().cpu()[:, :, :ntok_s]
    kr = ref[:, :, 1].permute(0, 2, 1, 3)
    assert ((kg - kr).abs() <= 6e-4 * kr.abs() + 2e-4).all()             # one fp16 plane: its own rounding
    # v^T: tokens of every aligned group of 16 in the order 0-3, 8-11, 4-7, 12-15
    t = torch.arange(ntok_s)
    perm = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    vg = vt.double().cpu()[:, :, :, :ntok_s][..., perm]
    vr = ref[:, :, 2].permute(0, 2, 3, 1)
    assert ((vg - vr).abs() <= 6e-4 * vr.abs() + 2e-4).all()


def test_mx_block_kernels_inside_the_vit(dev):
    """4 frames at 448^2 (12608 rows) through 3 blocks in precision "mixed" with the MX kernels (default) and with the bf16 x 3 kernels
    (WVN_NO_MX): both inside the mode's gate against the CPU oracle, and close to each other."""
    import os

    from oracle import interfaces as OI, vit as OV
    from wild_visual_navigation_amd.backbone import VitBackbone
    sd = OV.make_vit_state_dict("vit_small", 8, pretrain_grid=28, seed=0, depth=3)
    img = torch.rand(4, 3, 448, 448, generator=g(1))
    want = OV.vit_tokens(sd, OI.normalize(img), 8, 6)[:, 1:]
    os.environ.pop("WVN_NO_MX", None)
    bb = VitBackbone(sd, 448, 8, 6, device=dev, precision="mixed", max_chunk=4)
    assert bb.mx and bb.model.layers[1].fc1_w_mx
    a = bb.forward_tokens(img.to(dev)).cpu()
    os.environ["WVN_NO_MX"] = "1"
    try:
        b = VitBackbone(sd, 448, 8, 6, device=dev, precision="mixed", max_chunk=4).forward_tokens(img.to(dev)).cpu()
    finally:
        os.environ.pop("WVN_NO_MX", None)
    assert not torch.equal(a, b)                         # (the two routes really are different code)
    print(f"tokens vs oracle: MX {(a - want).abs().max().item():.2e}, bf16 x 3 {(b - want).abs().max().item():.2e}; MX vs x3 {(a - b).abs().max().item():.2e}")
    assert (a - want).abs().max().item() < 5e-4
    assert (a - b).abs().max().item() < 4e-4


def test_mx_with_layerscale_and_patch14(dev):
    """DINOv2 ViT-S/14 (D = 384: the MX kernels apply; LayerScale on both branch outputs goes through the MX row-panel epilogue) at 518^2, 7 frames =
    9632 token rows of 1376 per frame (a row stride that is not a multiple of 128), 3 blocks, precision "mixed": inside the gate against the CPU oracle
    and close to the bf16 x 3 kernels."""
    import os

    from oracle import interfaces as OI, vit as OV
    from wild_visual_navigation_amd.backbone import VitBackbone
    sd = OV.make_dinov2_state_dict("vit_small", 14, pretrain_grid=37, seed=3, depth=3)
    img = torch.rand(7, 3, 518, 518, generator=g(9))
    want = OV.vit_tokens(sd, OI.normalize(img), 14, 6)[:, 1:]
    os.environ.pop("WVN_NO_MX", None)
    bb = VitBackbone(sd, 518, 14, 6, device=dev, precision="mixed", max_chunk=7)
    assert bb.mx and bb.model.layers[1].proj_w_mx and bb.model.layers[1].ls1
    a = bb.forward_tokens(img.to(dev)).cpu()
    os.environ["WVN_NO_MX"] = "1"
    try:
        b = VitBackbone(sd, 518, 14, 6, device=dev, precision="mixed", max_chunk=7).forward_tokens(img.to(dev)).cpu()
    finally:
        os.environ.pop("WVN_NO_MX", None)
    print(f"dinov2 ViT-S/14 518^2 x 7, mixed: MX {(a - want).abs().max().item():.2e}, bf16 x 3 {(b - want).abs().max().item():.2e}")
    assert not torch.equal(a, b)
    assert (a - want).abs().max().item() < 5e-4
