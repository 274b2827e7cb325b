"""The fused block MLP (csrc/mlp_fused.hip: fc1 -> GELU -> fc2 with the hidden activation kept in registers, W1 / W2 through one
direct-to-LDS ring), called through the C-ABI, against (a) the un-fused kernel pair it replaces -- same bf16 rounding points,
so the two agree to fp32 summation-order noise -- and (b) fp32 math on the bf16-rounded inputs.  Shapes cover partial row
blocks, fewer row blocks than CUs, several row blocks per workgroup (the ring streams across them) and LayerScale; the repeat
test screens for pipeline races (ring reuse, counted vmcnt): a race shows up as run-to-run differences."""
import pytest
import torch
import torch.nn.functional as F

from wild_visual_navigation_amd import _lib, ops

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def make(M, Fh, dev, ls=False):
    xn = torch.randn(M, 384, generator=g(1)).to(torch.bfloat16)
    w1 = (torch.randn(Fh, 384, generator=g(2)) * 0.06).to(torch.bfloat16)
    b1 = torch.randn(Fh, generator=g(3)) * 0.5
    w2 = (torch.randn(384, Fh, generator=g(4)) * 0.03).to(torch.bfloat16)
    b2 = torch.randn(384, generator=g(5)) * 0.5
    x = torch.randn(M, 384, generator=g(6))
    gam = (torch.rand(384, generator=g(7)) + 0.5) if ls else None
    t = [xn, w1, b1, w2, b2, x] + ([gam] if ls else [])
    return [a.to(dev) for a in t] + ([] if ls else [None])


def unfused(xn, w1, b1, w2, b2, x):
    hid = ops.gemm_bf16(xn, w1, b1, _lib.EPI_GELU_BF16)
    out = x.clone()
    ops.gemm_bf16(hid, w2, b2, _lib.EPI_RESID_F32, out=out)
    return out


@pytest.mark.parametrize("M", [1, 31, 128, 300, 1000, 128 * 300 + 17])
@pytest.mark.parametrize("Fh", [1536, 64, 704])
def test_fused_matches_unfused_pair(dev, M, Fh):
    if M > 1000 and Fh != 1536:
        pytest.skip("large M only at the shipped width")
    xn, w1, b1, w2, b2, x, _ = make(M, Fh, dev)
    want = unfused(xn, w1, b1, w2, b2, x)
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)].contiguous()
    got = ops.mlp_fused(xn, w1, b1, w2p, b2, x.clone())
    torch.cuda.synchronize()
    # identical bf16 rounding of the hidden activation except where fp32 summation order flips a rounding (rare); the output
    # sums 1536 products of magnitude ~0.03: allow a few bf16 flips of single hidden units
    err = (got - want).abs().max().item()
    assert err <= 2e-3, err
    assert (got - want).abs().mean().item() <= 2e-5
    # fp32 math on the same inputs
    ref = x.cpu() + F.gelu(xn.float().cpu() @ w1.float().cpu().T + b1.cpu()).to(torch.bfloat16).float() @ w2.float().cpu().T + b2.cpu()
    assert (got.cpu() - ref).abs().max().item() <= 2e-2
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("M", [5, 128, 777, 128 * 300 + 17])
def test_fused_with_layernorm(dev, M):
    """xn=None: the kernel normalises the rows of x itself (blocks.i.norm2 folded in) -- against the LayerNorm kernel + un-fused pair"""
    Fh = 1536
    _, w1, b1, w2, b2, x, _ = make(M, Fh, dev)
    x = x * 3.0 + 0.7                                   # rows with a mean and a spread
    gam = (torch.rand(384, generator=g(8)) + 0.5).to(dev)
    bet = (torch.randn(384, generator=g(9)) * 0.2).to(dev)
    xn = F.layer_norm(x, (384,), gam, bet, 1e-6).to(torch.bfloat16)
    want = unfused(xn, w1, b1, w2, b2, x)
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)].contiguous()
    got = ops.mlp_fused(None, w1, b1, w2p, b2, x.clone(), ln=(gam, bet, 1e-6))
    torch.cuda.synchronize()
    # the LayerNorm statistics are summed in another order than torch's: a few normalised values land on the other side of a bf16
    # rounding boundary (1 ulp = 2^-8 relative on values ~1, times |w1| ~0.06 through GELU and |w2| ~0.03): 1e-2 covers it
    assert (got - want).abs().max().item() <= 1e-2
    assert (got - want).abs().mean().item() <= 1e-4
    assert torch.isfinite(got).all()
    again = ops.mlp_fused(None, w1, b1, w2p, b2, x.clone(), ln=(gam, bet, 1e-6))
    assert torch.equal(got, again)


@pytest.mark.parametrize("M", [5, 128, 777, 128 * 300 + 17])
@pytest.mark.parametrize("scaled", [False, True])
def test_projection_in_the_prologue(dev, M, scaled):
    """attention projection + residual, then LayerNorm + MLP + residual, as ONE launch -- against the two kernels it replaces
    (projection GEMM with residual epilogue, then the fused MLP kernel): same rounding points, the LayerNorm sees a residual
    stream that differs by fp32 summation order only"""
    Fh = 1536
    _, w1, b1, w2, b2, x, _ = make(M, Fh, dev)
    x = x * 2.0 + 0.4
    attn = (torch.randn(M, 384, generator=g(11)) * 1.5).to(torch.bfloat16).to(dev)
    wp = (torch.randn(384, 384, generator=g(12)) * 0.05).to(torch.bfloat16).to(dev)
    bp = (torch.randn(384, generator=g(13)) * 0.3).to(dev)
    gam = (torch.rand(384, generator=g(8)) + 0.5).to(dev)
    bet = (torch.randn(384, generator=g(9)) * 0.2).to(dev)
    ls1 = (torch.rand(384, generator=g(14)) + 0.5).to(dev) if scaled else None
    ls2 = (torch.rand(384, generator=g(15)) + 0.5).to(dev) if scaled else None
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)]
    pack = torch.cat([wp.reshape(-1), w1.reshape(-1), w2p.reshape(-1)]).contiguous()   # (as the backbone packs a layer)
    n0, n1 = wp.numel(), w1.numel()
    wp_v, w1_v, w2p_v = pack[:n0].view(384, 384), pack[n0:n0 + n1].view(Fh, 384), pack[n0 + n1:].view(384, Fh)
    # reference: the two kernels
    want = x.clone()
    if scaled:   # the projection GEMM has no LayerScale operand at this level: fold it into weights / bias of the reference
        y = attn.float() @ wp.float().T + bp
        want = want + ls1 * y
    else:
        ops.gemm_bf16(attn, wp, bp, _lib.EPI_RESID_F32, out=want)
    ops.mlp_fused(None, w1_v, b1, w2p_v, b2, want, ls=ls2, ln=(gam, bet, 1e-6))
    got = ops.proj_mlp_fused(attn, wp_v, bp, (gam, bet, 1e-6), w1_v, b1, w2p_v, b2, x.clone(), ls1=ls1, ls2=ls2)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 1e-2
    assert (got - want).abs().mean().item() <= 1e-4
    again = ops.proj_mlp_fused(attn, wp_v, bp, (gam, bet, 1e-6), w1_v, b1, w2p_v, b2, x.clone(), ls1=ls1, ls2=ls2)
    assert torch.equal(got, again), "run-to-run difference (pipeline race)"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 100, 128 * 3 + 5, 128 * 300 + 17, 128 * 600])
def test_resident_form_matches_fused(dev, M, dtype):
    """wvn_proj_mlp_resident (the residual rows stay in the accumulators: x read once and written once, the LayerNorm reads the
    accumulators, fc1.weight with swapped column bits) against wvn_proj_mlp_fused on the same operands: same rounding points;
    the products are accumulated onto x + bias instead of being added to it at the end, and the projection bias enters as two
    operand-format terms (bf16: exact to 2^-17 relative)."""
    Fh = 1536
    _, w1, b1, w2, b2, x, _ = make(M, Fh, dev)
    x = x * 2.0 + 0.4
    attn = (torch.randn(M, 384, generator=g(11)) * 1.5).to(dtype).to(dev)
    wp = (torch.randn(384, 384, generator=g(12)) * 0.05).to(dtype).to(dev)
    bp = (torch.randn(384, generator=g(13)) * 0.3).to(dev)
    gam = (torch.rand(384, generator=g(8)) + 0.5).to(dev)
    bet = (torch.randn(384, generator=g(9)) * 0.2).to(dev)
    w1, w2 = w1.to(dtype), w2.to(dtype)
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)].contiguous()
    w1p = w1[:, ops.vt_token_order(384, device=dev)].contiguous()
    if dtype == torch.bfloat16:
        want = ops.proj_mlp_fused(attn, wp, bp, (gam, bet, 1e-6), w1, b1, w2p, b2, x.clone())
    else:   # fp16 operands: the separate fp16 kernels
        want = x.clone()
        ops.gemm_bf16(attn, wp, bp, _lib.EPI_RESID_F32, out=want)
        xn = F.layer_norm(want, (384,), gam, bet, 1e-6).to(dtype)
        hid = ops.gemm_bf16(xn, w1, b1, _lib.EPI_GELU_BF16)
        ops.gemm_bf16(hid, w2, b2, _lib.EPI_RESID_F32, out=want)
    got = ops.proj_mlp_resident(attn, wp, bp, (gam, bet, 1e-6), w1p, b1, w2p, b2, x.clone())
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= (1e-2 if dtype == torch.bfloat16 else 4e-3)
    assert (got - want).abs().mean().item() <= 1e-4
    # the residual stream itself is exact to fp32 rounding where the MLP contributes nothing
    z = ops.proj_mlp_resident(attn * 0, wp, bp * 0, (gam, bet, 1e-6), w1p * 0, b1 * 0, w2p * 0, b2 * 0, x.clone())
    assert torch.equal(z, x)
    zb = ops.proj_mlp_resident(attn * 0, wp, bp, (gam, bet, 1e-6), w1p * 0, b1 * 0, w2p * 0, b2, x.clone())
    assert (zb - (x + bp + b2)).abs().max().item() <= 2e-5
    again = ops.proj_mlp_resident(attn, wp, bp, (gam, bet, 1e-6), w1p, b1, w2p, b2, x.clone())
    assert torch.equal(got, again), "run-to-run difference (pipeline race)"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("frames,ntok_s", [(1, 144), (3, 3152), (2, 784)])
def test_resident_form_hands_next_layernorm_to_qkv(dev, frames, ntok_s, dtype):
    """wvn_proj_mlp_resident with xn_next: the rows it stores are the rows of the plain call (b2 joins as two operand-format
    terms instead of an fp32 add), the fragments it leaves are LayerNorm(next)(those rows) in the operand format, and
    wvn_qkv_prenorm on them equals wvn_qkv_fused on the stored rows up to the operand ulps of rows that round the other way."""
    M, Fh, npad = frames * ntok_s, 1536, (ntok_s + 127) // 128 * 128
    _, w1, b1, w2, b2, x, _ = make(M, Fh, dev)
    x = x * 2.0 + 0.4
    x[:, 7] *= 12.0
    attn = (torch.randn(M, 384, generator=g(11)) * 1.5).to(dtype).to(dev)
    wp = (torch.randn(384, 384, generator=g(12)) * 0.05).to(dtype).to(dev)
    bp = (torch.randn(384, generator=g(13)) * 0.3).to(dev)
    gam = (torch.rand(384, generator=g(8)) + 0.5).to(dev)
    bet = (torch.randn(384, generator=g(9)) * 0.2).to(dev)
    ngam = (torch.rand(384, generator=g(18)) + 0.5).to(dev)
    nbet = (torch.randn(384, generator=g(19)) * 0.2).to(dev)
    wq = (torch.randn(1152, 384, generator=g(20)) * 0.05).to(dtype).to(dev)
    bq = (torch.randn(1152, generator=g(21)) * 0.3).to(dev)
    w1, w2 = w1.to(dtype), w2.to(dtype)
    perm = ops.vt_token_order(384, device=dev)
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)].contiguous()
    w1p, wqp = w1[:, perm].contiguous(), wq[:, perm].contiguous()
    plain = ops.proj_mlp_resident(attn, wp, bp, (gam, bet, 1e-6), w1p, b1, w2p, b2, x.clone())
    got, frag = ops.proj_mlp_resident(attn, wp, bp, (gam, bet, 1e-6), w1p, b1, w2p, b2, x.clone(), next_ln=(ngam, nbet, 1e-6))
    torch.cuda.synchronize()
    assert (got - plain).abs().max().item() <= 3e-5 * max(1.0, b2.abs().max().item())
    xn = ops.unpack_row_fragments(frag, M).float()
    want_xn = F.layer_norm(got, (384,), ngam, nbet, 1e-6)
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert ((xn - want_xn).abs() <= ulp * want_xn.abs() + 1e-3).all()
    assert torch.equal(ops.unpack_row_fragments(frag, M), want_xn.to(dtype)) or ((xn - want_xn.to(dtype).float()) != 0).float().mean().item() < 0.02
    QS = 0.125 * 1.4426950408889634
    q1 = ops.qkv_prenorm(frag, wqp, bq, M, frames, ntok_s, npad, QS)
    if dtype == torch.bfloat16:
        q0 = ops.qkv_fused(got, (ngam, nbet, 1e-6), wq, bq, frames, ntok_s, npad, QS)
        for name, a, b in zip("q k vt".split(), q1, q0):
            err = (a.float() - b.float()).abs()
            assert (err <= 1.6e-2 * b.float().abs() + 2e-2).all(), (name, err.max().item())
            assert err.mean().item() <= 1e-3, name
    else:   # fp16 operands: fp32 math on the fragments' own values
        y = xn @ wq.float().T + bq
        qh = (y[:, :384] * QS).view(frames, ntok_s, 6, 64).permute(0, 2, 1, 3).reshape(frames * 6, ntok_s, 64)
        err = (q1[0][:, :ntok_s].float() - qh).abs()
        assert (err <= 2e-3 * qh.abs() + 4e-3).all(), err.max().item()
    again = ops.qkv_prenorm(frag, wqp, bq, M, frames, ntok_s, npad, QS)
    for a, b in zip(q1, again):
        assert torch.equal(a, b), "run-to-run difference (pipeline race)"


def test_fused_layerscale(dev):
    M, Fh = 515, 1536
    xn, w1, b1, w2, b2, x, gam = make(M, Fh, dev, ls=True)
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)].contiguous()
    got = ops.mlp_fused(xn, w1, b1, w2p, b2, x.clone(), ls=gam).cpu()
    hid = F.gelu(xn.float().cpu() @ w1.float().cpu().T + b1.cpu()).to(torch.bfloat16).float()
    ref = x.cpu() + gam.cpu() * (hid @ w2.float().cpu().T + b2.cpu())
    assert (got - ref).abs().max().item() <= 3e-2


def test_fused_rows_past_m_untouched_and_repeatable(dev):
    M, Fh = 128 * 520 + 77, 1536   # > 2 row blocks per workgroup on 256 CUs, ragged tail
    xn, w1, b1, w2, b2, x, _ = make(M, Fh, dev)
    w2p = w2[:, ops.vt_token_order(Fh, device=dev)].contiguous()
    guard = torch.full((M + 256, 384), 7.0, device=dev)
    guard[:M] = x
    outs = []
    for _ in range(4):
        buf = guard.clone()
        ops.mlp_fused(xn, w1, b1, w2p, b2, buf[:M])
        torch.cuda.synchronize()
        assert (buf[M:] == 7.0).all(), "rows past M were written"
        outs.append(buf[:M].clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "run-to-run difference (pipeline race)"
    want = unfused(xn, w1, b1, w2, b2, x)
    assert (outs[0] - want).abs().max().item() <= 2e-3


def test_fused_argument_checks(dev):
    xn, w1, b1, w2, b2, x, _ = make(64, 1536, dev)
    with pytest.raises(_lib.WvnError):
        ops.mlp_fused(xn, w1[:100], b1[:100], w2[:, :100].contiguous(), b2, x)   # F % 64 != 0
