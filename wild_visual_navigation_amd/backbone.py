"""DINO ViT backbone on MI355X: host-side owner of the device weights / workspace around
``wvn_vit_forward`` (include/wvn_hip.h).

Replaces the module returned by ``stego.backbones.backbone.get_backbone(cfg)`` that the reference
calls at wild_visual_navigation/feature_extractor/dino_interface.py:45,84.  Weights come in the
upstream DINO state-dict layout, so released DINO checkpoints load unchanged; without a checkpoint
(this build has no network) seeded synthetic weights of the same architecture are used.
"""
import ctypes as C
import math
import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import _lib

ARCH = {"vit_small": (384, 12, 6), "vit_base": (768, 12, 12)}
PRECISIONS = ("fp16", "bf16", "mixed", "exact", "fp32", "fp8")   # names VitBackbone(precision=...) accepts in this build


def synthetic_vit_state_dict(arch="vit_small", patch=8, pretrain_grid=28, seed=0, depth=None, dinov2=False) -> Dict[str, torch.Tensor]:
    """Seeded random weights with realistic scales in the upstream DINO layout (no network here to
    fetch dl.fbaipublicfiles.com/dino checkpoints).  LayerNorm affine and biases are non-trivial.
    ``dinov2``: add the LayerScale vectors ``blocks.i.ls{1,2}.gamma`` of the published DINOv2 block."""
    D, dd, _ = ARCH[arch]
    depth = dd if depth is None else depth
    g = torch.Generator().manual_seed(seed)

    def tn(*s, std=0.02):
        return torch.nn.init.trunc_normal_(torch.empty(*s), std=std, a=-2 * std, b=2 * std, generator=g)

    def rn(*s, std):
        return torch.randn(*s, generator=g) * std

    sd = {
        "patch_embed.proj.weight": tn(D, 3, patch, patch, std=0.05), "patch_embed.proj.bias": rn(D, std=0.02),
        "cls_token": tn(1, 1, D), "pos_embed": tn(1, 1 + pretrain_grid * pretrain_grid, D, std=0.1),
        "norm.weight": 1.0 + rn(D, std=0.1), "norm.bias": rn(D, std=0.1),
    }
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + rn(D, std=0.1)
        sd[p + "norm1.bias"] = rn(D, std=0.05)
        sd[p + "attn.qkv.weight"] = tn(3 * D, D, std=0.06)
        sd[p + "attn.qkv.bias"] = rn(3 * D, std=0.02)
        sd[p + "attn.proj.weight"] = tn(D, D, std=0.04)
        sd[p + "attn.proj.bias"] = rn(D, std=0.02)
        sd[p + "norm2.weight"] = 1.0 + rn(D, std=0.1)
        sd[p + "norm2.bias"] = rn(D, std=0.05)
        sd[p + "mlp.fc1.weight"] = tn(4 * D, D, std=0.04)
        sd[p + "mlp.fc1.bias"] = rn(4 * D, std=0.02)
        sd[p + "mlp.fc2.weight"] = tn(D, 4 * D, std=0.03)
        sd[p + "mlp.fc2.bias"] = rn(D, std=0.02)
    if dinov2:
        g2 = torch.Generator().manual_seed(seed + 77)
        for i in range(depth):
            sd[f"blocks.{i}.ls1.gamma"] = 0.2 + 0.8 * torch.rand(D, generator=g2)
            sd[f"blocks.{i}.ls2.gamma"] = 0.2 + 0.8 * torch.rand(D, generator=g2)
    return sd


def split_planes(t: torch.Tensor) -> torch.Tensor:
    """fp32 [...] -> bf16 [2, ...]: hi = bf16(t), lo = bf16(t - hi) (the exact-mode operand representation)."""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


def pack_fc2_fragment_major(w: torch.Tensor) -> torch.Tensor:
    """fc2.weight [384][F] fp32 -> the packed operand of the fragment-major split-operand MLP (include/wvn_hip.h, wvn_vit_layer.fc2_w_fused
    for WVN_PREC_X3 / WVN_PREC_MIX): bf16 [F / 16][2 planes][384][2 chunks][8]; k-step-major so that a k-step's 24 KB are contiguous, the
    column order inside a k-step = the order the fc1 accumulators hand the hidden activation over (bits 2 and 3 swapped), the two
    16-byte chunks of row n exchanged where (n >> 3) & 1 (the LDS bank swizzle, applied here because the DMA copies lane-linear)."""
    N, K = w.shape
    planes = split_planes(w.detach().float())                      # [2][N][K]
    i = torch.arange(16)
    sw = (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)
    p = planes.reshape(2, N, K // 16, 16)[..., sw.to(planes.device)].reshape(2, N, K // 16, 2, 8).clone()
    flip = ((torch.arange(N) >> 3) & 1).bool().to(planes.device)
    p[:, flip] = p[:, flip].flip(dims=[3])
    return p.permute(2, 0, 1, 3, 4).contiguous()


# ---- MX correction terms (round 6): an operand v as  h = fp16(v),  l8 = e5m2((v - h) * 2^12),  h8 = e5m2(v)  (csrc/gemm_n384_x3.hip) ----
MX_RES_SCALE = 4096.0


def _e5m2_bytes(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> e5m2 bytes (round to nearest even, finite overflow saturates at +-57344: what v_cvt_*_bf8_* does under MODE.FP16_OVFL)."""
    return t.float().clamp(-57344.0, 57344.0).to(torch.float8_e5m2).view(torch.uint8)


def mx_split(t: torch.Tensor):
    """fp32 [...] -> (h fp16, l8 uint8, h8 uint8): the three planes of the MX operand representation."""
    t = t.detach().float()
    h = t.clamp(-65504.0, 65504.0).to(torch.float16)
    return h, _e5m2_bytes((t - h.float()) * MX_RES_SCALE), _e5m2_bytes(t)


def _swap23(n: int, device=None) -> torch.Tensor:
    i = torch.arange(n, device=device)
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def pack_n384_mx(w: torch.Tensor) -> torch.Tensor:
    """W [384][K] fp32 -> the stage list of gemm_n384_mx_pair_kernel: uint8 [K / 64][4 stages][24 KB].  Per 64 k (c): stage 0 / 1 = W_h of
    k-steps (4c + 0 | 1) / (4c + 2 | 3): [x = k-step parity][384 n][2 chunks][8 fp16], element j of chunk h = W_h[n, 16 s + swap23(8 h + j)];
    stage 2 / 3 = W_l8 / W_h8: [x = half][384 n][2 chunks][16 bytes], byte (sp, j) of chunk h in half x = W8[n, 16 (4c + 2x + sp) + swap23(8 h + j)]
    -- the element order in which the producers' accumulators hold a row.  The two 16-byte chunks of row n are exchanged where (n >> 3) & 1
    (the LDS bank swizzle, baked in because the DMA copies lane-linear)."""
    N, K = w.shape
    assert N == 384 and K % 128 == 0
    h, l8, h8 = mx_split(w)
    sw = _swap23(16, w.device)
    flip = ((torch.arange(N, device=w.device) >> 3) & 1).bool()
    hp = h.reshape(N, K // 16, 16)[..., sw].reshape(N, K // 64, 2, 2, 2, 8).clone()      # [n][c][stage][x][h][j]
    hp[flip] = hp[flip].flip(dims=[4])
    hp = hp.permute(1, 2, 3, 0, 4, 5).contiguous().view(torch.uint8).reshape(K // 64, 2, 24576)
    planes = []
    for b8 in (l8, h8):
        q = b8.reshape(N, K // 16, 16)[..., sw].reshape(N, K // 64, 2, 2, 2, 8)           # [n][c][x][sp][h][j]
        q = q.permute(1, 2, 0, 4, 3, 5).reshape(K // 64, 2, N, 2, 16).clone()            # [c][x][n][h][(sp, j)]
        q[:, :, flip] = q[:, :, flip].flip(dims=[3])
        planes.append(q.reshape(K // 64, 1, 24576))
    return torch.cat([hp] + planes, dim=1).contiguous()


def pack_a384_mx(w: torch.Tensor) -> torch.Tensor:
    """W [N][384] fp32 -> the operand of the A-stationary MX kernels (csrc/gemm_a384_x3.hip): uint8 [2 images][2][N][768].
    Image 0 (the one-wave-per-SIMD kernel): plane 0 = fp16(W) row-major, plane 1 = per 128-k slice ks sixteen 16-byte chunks, chunk 2 s'' + h with
    s'' = 4 which + 2 mm + x (which: 0 = l8, 1 = h8; mm: 64-k step of the slice; x: half of the lane's 32 bytes), byte (sp, j) of it =
    W8[n, 128 ks + 64 mm + 16 (2 x + sp) + 8 h + j] -- natural k order, the order the LayerNorm-on-load prologue fills the A operands in.
    Image 1 (the two-workgroups-per-CU kernel, gemm_a384_mx2_kernel): [N / 32 tiles][3 slices of 128 k][32 chunk images][32 rows][16 B]; chunk images
    0 - 15 = fp16 W[n, 128 ks + 8 c .. + 8], 16 - 31 = the 16-byte chunks 8 mm + 4 which + 2 x + h of the same bytes as above."""
    N, K = w.shape
    assert K == 384 and N % 64 == 0
    h, l8, h8 = mx_split(w)
    q = torch.stack([l8, h8], dim=1).reshape(N, 2, 3, 2, 2, 2, 2, 8)            # [n][which][ks][mm][x][sp][h][j]
    q0 = q.permute(0, 2, 1, 3, 4, 6, 5, 7).contiguous().reshape(N, 768)         # [n][ks][which][mm][x][h][sp][j]
    img0 = torch.stack([h.contiguous().view(torch.uint8).reshape(N, 768), q0])
    hb = h.contiguous().view(torch.uint8).reshape(N // 32, 32, 3, 16, 16)       # [tile][row][ks][image 2 s + h][16 B]
    q1 = q.permute(0, 2, 3, 1, 4, 6, 5, 7).contiguous().reshape(N // 32, 32, 3, 16, 16)   # [n][ks][mm][which][x][h][sp][j] -> [tile][row][ks][image 8 mm + 4 which + 2 x + h][16 B]
    img1 = torch.cat([hb, q1], dim=3).permute(0, 2, 3, 1, 4).contiguous()       # [tile][ks][image][row][16 B]
    return torch.stack([img0, img1.reshape(2, N, 768)]).contiguous()


def pack_a768_fp8(wq: torch.Tensor) -> torch.Tensor:
    """e4m3 weight bytes [N][768] (uint8 or float8_e4m3fn) -> the operand of the A-stationary fp8 kernel (csrc/gemm_a768_fp8.hip): uint8
    [N / 32 tiles][12 k-steps][2 halves x][64 lanes = (hi, row)][16 B]; byte j of lane (hi, row) in (s, x) = W[32 tile + row, 64 s + 32 hi + 16 x + j]."""
    N, K = wq.shape
    assert K == 768 and N % 32 == 0
    b = wq.contiguous().view(torch.uint8).reshape(N // 32, 32, 12, 2, 2, 16)     # [tile][row][s][hi][x][16]
    return b.permute(0, 2, 4, 3, 1, 5).contiguous()                              # [tile][s][x][hi][row][16]


def mx_fragments(a: torch.Tensor):
    """A [M][K] fp32 -> the fragment-major MX planes the row-panel kernel reads (what the producers' epilogues write):
    (h fp16 [R][K / 16][64][8], l8 uint8 [R][K / 64][2][64][16], h8 likewise -- h8 is NOT read by the kernels any more: they derive e5m2(h) in
    registers; kept for the tests' bookkeeping), R = ceil(M / 32); lane = 32 hw + row, element j of k-step s =
    A[32 R + row, 16 s + swap23(8 hw + j)], byte (sp, j) of half x of 64-k step c = A8[.., 16 (4c + 2x + sp) + swap23(8 hw + j)].  Rows past M: 0."""
    M, K = a.shape
    R = (M + 31) // 32
    ap = torch.zeros(R * 32, K, device=a.device)
    ap[:M] = a.float()
    h, l8, h8 = mx_split(ap)
    sw = _swap23(16, a.device)
    hf = h.reshape(R, 32, K // 16, 16)[..., sw].reshape(R, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()    # [R][s][hw][row][j]
    outs = [hf.reshape(R, K // 16, 64, 8)]
    for b8 in (l8, h8):
        q = b8.reshape(R, 32, K // 16, 16)[..., sw].reshape(R, 32, K // 64, 2, 2, 2, 8)    # [R][row][c][x][sp][hw][j]
        outs.append(q.permute(0, 2, 3, 5, 1, 4, 6).contiguous().reshape(R, K // 64, 2, 64, 16))
    return tuple(outs)


def mx_matmul_reference(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """float64 statement of what an MX kernel computes for a [M][K] x w [N][K]^T (the operand roundings exactly, the sums in float64)."""
    ah, al8, _ = mx_split(a)
    ah8 = _e5m2_bytes(ah.float())                      # the activation's h8 operand is derived from its fp16 image in registers (gemm_n384_x3.hip: derive_h8)
    wh, wl8, wh8 = mx_split(w)
    f = lambda b: b.view(torch.float8_e5m2).double()   # noqa: E731
    return ah.double() @ wh.double().T + (f(ah8) @ f(wl8).T + f(al8) @ f(wh8).T) / MX_RES_SCALE


def resample_pos_embed(pos_embed: torch.Tensor, grid: int, rule: str = "dino") -> torch.Tensor:
    """DINO's position-table resampling (rule "dino": bicubic, scale (grid+0.1)/g; DINOv2 keeps the same rule with its default
    interpolate_offset = 0.1, antialias off; rule "size": bicubic to size=(grid, grid) -- HuggingFace's interpolate_pos_encoding and the
    DINO code under torch < 1.6: the sampling step differs by 0.18 %), done ONCE when the model is
    built -- it depends only on the input size, so it is weight preparation, not per-frame work."""
    n_pre = pos_embed.shape[1] - 1
    g = int(round(math.sqrt(n_pre)))
    if g == grid:
        return pos_embed.clone()
    D = pos_embed.shape[-1]
    tab = pos_embed[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2).float()
    if rule == "dino":
        sf = (grid + 0.1) / g
        tab = F.interpolate(tab, scale_factor=(sf, sf), mode="bicubic")
    elif rule == "size":
        tab = F.interpolate(tab, size=(grid, grid), mode="bicubic", align_corners=False)
    else:
        raise _lib.WvnError(f"pos_embed_rule {rule!r}: 'dino' or 'size'")
    if tab.shape[-1] != grid or tab.shape[-2] != grid:
        raise _lib.WvnError(f"position table resampling produced {tuple(tab.shape)}, expected grid {grid}")
    tab = tab.permute(0, 2, 3, 1).reshape(1, grid * grid, D)
    return torch.cat([pos_embed[:, :1].float(), tab], dim=1)


def _debug_bits() -> int:
    """WVN_X3_DEBUG_BITS: the documented A/B bits of the split-operand block kernels only (64 | 128 | 256 | 512, include/wvn_hip.h); a
    malformed value is an error with the variable's name in it, other bits are refused (the raw value used to be OR'ed into
    the production flags, where bits 1 / 2 / 4 silently toggled unrelated fusions)."""
    raw = os.environ.get("WVN_X3_DEBUG_BITS", "").strip()
    if not raw:
        return 0
    try:
        v = int(raw, 0)
    except ValueError:
        raise _lib.WvnError(f"WVN_X3_DEBUG_BITS={raw!r} is not an integer") from None
    if v & ~(64 | 128 | 256 | 512):
        raise _lib.WvnError(f"WVN_X3_DEBUG_BITS={raw!r}: only the bits 64, 128, 256 and 512 are debug switches")
    return v


class VitBackbone:
    """Device-resident DINO / DINOv2 ViT.  ``precision``: "bf16" (MFMA fast path), "fp16" (the same kernels with fp16 operands:
    same speed, 8x less operand rounding -- csrc/operand.h), "mixed" (the <= 1e-3 parity mode sized by the error budget: the linears as
    in "exact", the two attention products on the fp16 kernel -- include/wvn_hip.h WVN_PREC_MIX), "exact" (hi + lo split bf16 operands, three
    MFMAs per product: fp32-class results on the matrix pipe, the <= 1e-3 parity mode) or "fp32" (the same gate on fp32 FMA
    kernels; slow, kept as the independent cross-check of "exact") or "fp8" (BASELINE configs[4]: the four linears of every block on
    e4m3 MFMA at twice the bf16 rate, per-token / per-channel scales; everything else as "bf16")."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], img_size: int, patch: int, heads: int,
                 device="cuda", precision: str = "bf16", max_chunk: int = 16, fuse_mlp: Optional[bool] = None,
                 fuse_qkv: Optional[bool] = None, fuse_proj: bool = True, qsplit_blocks: Optional[int] = None, pos_embed_rule: str = "dino"):
        """qsplit_blocks (precision "mixed"): the number of LEADING blocks whose attention takes q as two fp16 planes (None: the
        library's default, include/wvn_hip.h WVN_VIT_QSPLIT_DEFAULT; the environment variable WVN_QSPLIT_BLOCKS overrides None)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.WvnError("VitBackbone needs a GPU device: the HIP path has no CPU fallback")
        self.lib = _lib.lib()
        self.precision = {"bf16": _lib.PREC_BF16, "fp16": _lib.PREC_F16, "fp32": _lib.PREC_F32, "exact": _lib.PREC_X3,
                          "fp8": _lib.PREC_FP8, "mixed": _lib.PREC_MIX}[precision]
        self._lowp16 = {_lib.PREC_BF16: torch.bfloat16, _lib.PREC_FP8: torch.bfloat16, _lib.PREC_F16: torch.float16}.get(self.precision)
        self.precision_name = precision
        self.img_size, self.patch, self.heads = img_size, patch, heads
        self.grid = img_size // patch
        self.dim = state_dict["cls_token"].shape[-1]
        self.depth = 1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("blocks."))
        self.mlp_dim = state_dict["blocks.0.mlp.fc1.weight"].shape[0]
        self.max_chunk = max_chunk
        # block MLP as one kernel with the hidden activation kept on chip (csrc/mlp_fused.hip): bf16, D = 384
        lowp16 = self.precision in (_lib.PREC_BF16, _lib.PREC_F16)
        can_fuse = lowp16 and self.dim == 384 and self.mlp_dim % 64 == 0 and self.mlp_dim <= 2176
        if fuse_mlp and not can_fuse:
            raise _lib.WvnError("fuse_mlp needs precision 'bf16' or 'fp16', dim 384 and mlp_dim % 64 == 0")
        self.fuse_mlp = can_fuse if fuse_mlp is None else bool(fuse_mlp)
        # LayerNorm 1 + QKV projection as one kernel (csrc/qkv_fused.hip): bf16, D = 384 with 6 heads
        can_fuse_qkv = lowp16 and self.dim == 384 and heads == 6
        if fuse_qkv and not can_fuse_qkv:
            raise _lib.WvnError("fuse_qkv needs precision 'bf16' or 'fp16', dim 384 and 6 heads")
        self.fuse_qkv = can_fuse_qkv if fuse_qkv is None else bool(fuse_qkv)
        self._fuse_args = (fuse_mlp, fuse_qkv, fuse_proj)
        # precision "mixed": the MX form of the block linears (fp16 hi * hi + scaled e5m2 correction products); WVN_NO_MX=1: the bf16 x 3 kernels (A/B, tests)
        self.mx = self.precision == _lib.PREC_MIX and not os.environ.get("WVN_NO_MX")
        self._sd = state_dict  # kept (host / original tensors) so that .to(device) can re-home the model
        self._keep = []  # device tensors referenced by raw pointers in the C struct

        def mat(t, pad_cols=0):
            t = t.detach().float()
            if pad_cols:
                t = F.pad(t, (0, pad_cols))
            t = t.to(self.device)
            if self._lowp16 is not None:
                t = t.to(self._lowp16).contiguous()
            elif self.precision in (_lib.PREC_X3, _lib.PREC_MIX):  # two stacked bf16 planes: hi = bf16(w), lo = bf16(w - hi)
                t = split_planes(t)
            else:
                t = t.contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def mat8(t):
            """fp8 (OCP e4m3) weight with one scale per output channel: w ~ q * s, s = amax(row) / 448."""
            t = t.detach().float().cpu()
            s = t.abs().amax(dim=1).clamp_min(1e-30) / 448.0
            qw = (t / s[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(self.device).contiguous()
            s = s.to(self.device).contiguous()
            self._keep += [qw, s]
            if t.shape[1] == 768 and t.shape[0] % 32 == 0:   # the A-stationary kernel's packed image rides behind in _pack8 (csrc/gemm_a768_fp8.hip)
                self._pack8 = pack_a768_fp8(qw)
                self._keep.append(self._pack8)
            else:
                self._pack8 = None
            return qw.data_ptr(), s.data_ptr()

        def vec(t):
            t = t.detach().to(self.device, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        sd = state_dict
        m = _lib.VitModel()
        m.img_size, m.patch, m.dim, m.depth, m.heads, m.mlp_dim = img_size, patch, self.dim, self.depth, heads, self.mlp_dim
        m.precision = self.precision
        # None: the library decides by size (the fused kernels pay from about half a chip of row blocks on); True: always
        m.flags = ((_lib.VIT_MLP_FUSED if self.fuse_mlp else 0) | (_lib.VIT_QKV_FUSED if self.fuse_qkv else 0)
                   | (_lib.VIT_FUSE_ANY_SIZE if (fuse_mlp is True or fuse_qkv is True) else 0)
                   | (0 if fuse_proj else _lib.VIT_NO_PROJ_IN_MLP) | (_lib.VIT_NO_LN_HANDOVER if os.environ.get("WVN_NO_HANDOVER") else 0)
                   | (_lib.VIT_NO_A384_X3 if os.environ.get("WVN_NO_A384_X3") else 0) | _debug_bits())
        if qsplit_blocks is None and os.environ.get("WVN_QSPLIT_BLOCKS", "") != "":
            qsplit_blocks = int(os.environ["WVN_QSPLIT_BLOCKS"])
        if qsplit_blocks is not None:
            if not 0 <= int(qsplit_blocks) <= 62:
                raise _lib.WvnError(f"qsplit_blocks must be in 0..62, not {qsplit_blocks}")
            m.flags |= _lib.vit_qsplit_blocks(qsplit_blocks)
        kp = 3 * patch * patch  # the MFMA GEMMs read patch rows padded to a multiple of 64 columns (588 -> 640 for patch 14)
        m.patch_w = mat(sd["patch_embed.proj.weight"].reshape(self.dim, -1),
                        0 if self.precision == _lib.PREC_F32 else (-kp) % 64)
        m.patch_b = vec(sd["patch_embed.proj.bias"])
        self.dinov2 = any(k.endswith("ls1.gamma") for k in sd)
        self.pos_embed_rule = pos_embed_rule   # "dino": scale_factor (grid + 0.1) / g as published | "size": HuggingFace's / torch < 1.6's reading
        pos = resample_pos_embed(sd["pos_embed"].float().cpu(), self.grid, pos_embed_rule)[0]  # [1+G*G, D]
        m.cls_pos = vec(sd["cls_token"].reshape(-1).float().cpu() + pos[0])
        m.pos = vec(pos)
        m.norm_g, m.norm_b = vec(sd["norm.weight"]), vec(sd["norm.bias"])
        for i in range(self.depth):
            p = f"blocks.{i}."
            L = m.layers[i]
            if self.precision == _lib.PREC_FP8:
                # (the *_w_mx fields carry the packed images of the K = 768 linears in this precision: include/wvn_hip.h)
                L.qkv_w, L.qkv_s = mat8(sd[p + "attn.qkv.weight"])
                L.qkv_w_mx = self._pack8.data_ptr() if self._pack8 is not None else 0
                L.proj_w, L.proj_s = mat8(sd[p + "attn.proj.weight"])
                L.proj_w_mx = self._pack8.data_ptr() if self._pack8 is not None else 0
                L.fc1_w, L.fc1_s = mat8(sd[p + "mlp.fc1.weight"])
                L.fc1_w_mx = self._pack8.data_ptr() if self._pack8 is not None else 0
                L.fc2_w, L.fc2_s = mat8(sd[p + "mlp.fc2.weight"])
            else:
                L.qkv_w = mat(sd[p + "attn.qkv.weight"])
                if self.fuse_qkv and self.fuse_mlp and (p + "ls1.gamma") not in sd and self.dim == 384 and not os.environ.get("WVN_NO_RESIDENT"):
                    # the copy the QKV kernel multiplies with when the previous block's kernel hands the normalised rows over as
                    # operand fragments (include/wvn_hip.h: qkv_w_fused)
                    kk = torch.arange(self.dim)
                    L.qkv_w_fused = mat(sd[p + "attn.qkv.weight"][:, (kk & ~12) | ((kk & 4) << 1) | ((kk & 8) >> 1)])
                if not self.fuse_mlp:
                    L.proj_w = mat(sd[p + "attn.proj.weight"])
                w2 = sd[p + "mlp.fc2.weight"]
                L.fc2_w = mat(w2)
                if self.precision in (_lib.PREC_X3, _lib.PREC_MIX) and self.dim == 384 and self.mlp_dim % 96 == 0:
                    t = pack_fc2_fragment_major(w2.to(self.device))
                    self._keep.append(t)
                    L.fc2_w_fused = t.data_ptr()
                    if self.precision == _lib.PREC_MIX and heads * 64 == self.dim:   # the attention kernel hands its output over as fragments
                        t = pack_fc2_fragment_major(sd[p + "attn.proj.weight"].to(self.device))
                        self._keep.append(t)
                        L.proj_w_frag = t.data_ptr()
                        if self.mlp_dim % 128 == 0 and self.mx:   # the MX operand form of the four block linears (round 6)
                            for name, key, pack in (("qkv_w_mx", "attn.qkv.weight", pack_a384_mx), ("proj_w_mx", "attn.proj.weight", pack_n384_mx),
                                                    ("fc1_w_mx", "mlp.fc1.weight", pack_a384_mx), ("fc2_w_mx", "mlp.fc2.weight", pack_n384_mx)):
                                t = pack(sd[p + key].detach().float().to(self.device))
                                self._keep.append(t)
                                setattr(L, name, t.data_ptr())
                if self.fuse_mlp:
                    # proj.weight, fc1.weight and the fused kernel's own copy of fc2.weight (hidden index in the order the fc1
                    # accumulators hand it over, wvn_hip.h), one allocation per layer
                    # ... and, where the block has no LayerScale, fc1.weight with its column index in the order the projection
                    # accumulators hand the residual rows over (the kernel then keeps the rows in registers for the whole block)
                    k = torch.arange(self.mlp_dim)
                    swap23 = lambda i: (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)   # noqa: E731
                    w1 = sd[p + "mlp.fc1.weight"]
                    parts = [sd[p + "attn.proj.weight"], w1, w2[:, swap23(k)]]
                    resident = (p + "ls1.gamma") not in sd and self.dim == 384 and not os.environ.get("WVN_NO_RESIDENT")   # (env: A/B runs)
                    if resident:
                        parts.append(w1[:, swap23(torch.arange(self.dim))])
                    pack = torch.cat([t.detach().float().reshape(-1) for t in parts]).to(self.device).to(self._lowp16).contiguous()
                    self._keep.append(pack)
                    n0, n1, n2 = parts[0].numel(), parts[1].numel(), parts[2].numel()
                    L.proj_w, L.fc1_w, L.fc2_w_fused = pack.data_ptr(), pack.data_ptr() + 2 * n0, pack.data_ptr() + 2 * (n0 + n1)
                    if resident:
                        L.fc1_w_fused = pack.data_ptr() + 2 * (n0 + n1 + n2)
                else:
                    L.fc1_w = mat(sd[p + "mlp.fc1.weight"])
            L.qkv_b, L.proj_b = vec(sd[p + "attn.qkv.bias"]), vec(sd[p + "attn.proj.bias"])
            L.fc1_b, L.fc2_b = vec(sd[p + "mlp.fc1.bias"]), vec(sd[p + "mlp.fc2.bias"])
            L.ln1_g, L.ln1_b = vec(sd[p + "norm1.weight"]), vec(sd[p + "norm1.bias"])
            L.ln2_g, L.ln2_b = vec(sd[p + "norm2.weight"]), vec(sd[p + "norm2.bias"])
            if p + "ls1.gamma" in sd:  # DINOv2 LayerScale
                L.ls1, L.ls2 = vec(sd[p + "ls1.gamma"]), vec(sd[p + "ls2.gamma"])
        self.model = m
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = 0

    def to(self, device) -> "VitBackbone":
        """A copy of this backbone on another GPU (DinoInterface.change_device, dino_interface.py:61-68); self if unchanged."""
        device = torch.device(device)
        if device == self.device:
            return self
        return VitBackbone(self._sd, self.img_size, self.patch, self.heads, device=device, precision=self.precision_name,
                           max_chunk=self.max_chunk, fuse_mlp=self._fuse_args[0], fuse_qkv=self._fuse_args[1], fuse_proj=self._fuse_args[2],
                           pos_embed_rule=self.pos_embed_rule)

    # ---- workspace (needs no initialisation: wvn_vit_forward resets the padding rows it relies on at every call) ----
    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch < batch:
            n = self.lib.wvn_vit_workspace_bytes(C.byref(self.model), batch)
            self._ws = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    @property
    def lowp_dtype(self):
        """dtype of ``lowp_out`` rows (None: the exact mode hands out fp32 tokens only)."""
        return {_lib.PREC_BF16: torch.bfloat16, _lib.PREC_FP8: torch.bfloat16, _lib.PREC_F16: torch.float16,
                _lib.PREC_F32: torch.float32}.get(self.precision)

    def forward_tokens(self, img: torch.Tensor, out: Optional[torch.Tensor] = None,
                       lowp_out: Optional[torch.Tensor] = None, flip: bool = False) -> torch.Tensor:
        """img [B,3,H,W]: fp32 in [0,1] or raw uint8 frames (x/255 is then fused into the patch gather, bit-identical to passing
        ``img.float() / 255``), at the network size or at ANY camera size -- the NEAREST resize + centre crop of
        dino_interface.py:52-59 is then fused into the same gather through index tables (``transforms.ingest_tables``;
        ``wvn_vit_forward_frames``).  ``flip``: run on the horizontally mirrored network input (STEGO's second pass; reversed
        column table, no copy).  -> final-LN patch tokens [B, G*G, D] fp32 (fresh tensor unless ``out`` is given).
        ``lowp_out`` (optional, [B*G*G, ld] in the model precision) receives the same values for a following MFMA GEMM
        (STEGO head).  Frames are pushed through in chunks of ``max_chunk`` so a chunk's activations stay resident in the
        256 MB Infinity Cache."""
        _lib.require_cuda(img, "img")
        B, Cc, Hs, Ws = img.shape
        if Cc != 3:
            raise _lib.WvnError(f"expected [B,3,H,W], got {tuple(img.shape)}")
        if lowp_out is not None and self.precision in (_lib.PREC_X3, _lib.PREC_MIX):
            raise _lib.WvnError("precisions 'exact' / 'mixed' return fp32 tokens only (split them with ops.split_planes)")
        if img.dtype != torch.uint8:
            img = img.float()
        img = img.contiguous()
        u8 = img.dtype == torch.uint8
        direct = Hs == self.img_size and Ws == self.img_size and not flip
        tab = None
        if not direct:
            from .feature_extractor.transforms import ingest_tables
            tab = ingest_tables(Hs, Ws, self.img_size, self.device, flip=flip)
        if direct and u8 and not (self._lowp16 is not None and self.patch == 8):
            img = img.float() / 255   # wvn_vit_forward_u8 covers the 16-bit-operand precisions at patch 8; others take fp32 here
            u8 = False
        P = self.grid * self.grid
        if out is None:
            out = torch.empty(B, P, self.dim, dtype=torch.float32, device=self.device)
        chunk = min(self.max_chunk, B)
        ws = self._workspace(chunk)
        st = _lib.stream()
        esz = 2 if self._lowp16 is not None else 4
        frame_bytes = 3 * Hs * Ws * img.element_size()
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            lp, ld = 0, 0
            if lowp_out is not None:
                ld = lowp_out.stride(0)
                lp = lowp_out.data_ptr() + b0 * P * ld * esz
            src = img.data_ptr() + b0 * frame_bytes
            if tab is not None:
                rc = self.lib.wvn_vit_forward_frames(C.byref(self.model), src, int(u8), Hs, Ws, tab.rows.data_ptr(), tab.cols.data_ptr(),
                                                     nb, out[b0:].data_ptr(), lp, ld, ws.data_ptr(), ws.numel(), st)
                _lib.check(rc, "wvn_vit_forward_frames")
            elif u8:
                _lib.check(self.lib.wvn_vit_forward_u8(C.byref(self.model), src, nb, out[b0:].data_ptr(), lp, ld, ws.data_ptr(), ws.numel(), st),
                           "wvn_vit_forward_u8")
            else:
                _lib.check(self.lib.wvn_vit_forward(C.byref(self.model), src, nb, out[b0:].data_ptr(), lp, ld, ws.data_ptr(), ws.numel(), st),
                           "wvn_vit_forward")
        return out

    def forward_tokens_pair(self, img: torch.Tensor, lowp_out: Optional[torch.Tensor] = None):
        """The frames AND their horizontal mirror images through the network in one launch sequence per chunk
        (``wvn_vit_forward_frames_pair``: twice the rows per launch; bit-identical to ``forward_tokens(img)`` and
        ``forward_tokens(img, flip=True)``).  Returns (tokens [2B, G*G, D] fp32, chunk): rows are laid out chunk by chunk,
        [chunk's frames | chunk's mirrors]; ``lowp_out`` ([2B*G*G, ld]) likewise."""
        _lib.require_cuda(img, "img")
        B, Cc, Hs, Ws = img.shape
        if Cc != 3:
            raise _lib.WvnError(f"expected [B,3,H,W], got {tuple(img.shape)}")
        if lowp_out is not None and self.precision in (_lib.PREC_X3, _lib.PREC_MIX):
            raise _lib.WvnError("precisions 'exact' / 'mixed' return fp32 tokens only (split them with ops.split_planes)")
        if img.dtype != torch.uint8:
            img = img.float()
        img = img.contiguous()
        from .feature_extractor.transforms import ingest_tables
        tab = ingest_tables(Hs, Ws, self.img_size, self.device, flip=False)
        tabm = ingest_tables(Hs, Ws, self.img_size, self.device, flip=True)
        P = self.grid * self.grid
        out = torch.empty(2 * B, P, self.dim, dtype=torch.float32, device=self.device)
        chunk = min(self.max_chunk, B)
        ws = self._workspace(2 * chunk)
        st = _lib.stream()
        esz = 2 if self._lowp16 is not None else 4
        frame_bytes = 3 * Hs * Ws * img.element_size()
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            lp, ld = 0, 0
            if lowp_out is not None:
                ld = lowp_out.stride(0)
                lp = lowp_out.data_ptr() + 2 * b0 * P * ld * esz
            rc = self.lib.wvn_vit_forward_frames_pair(C.byref(self.model), img.data_ptr() + b0 * frame_bytes, int(img.dtype == torch.uint8),
                                                      Hs, Ws, tab.rows.data_ptr(), tab.cols.data_ptr(), tabm.cols.data_ptr(), nb,
                                                      out[2 * b0:].data_ptr(), lp, ld, ws.data_ptr(), ws.numel(), st)
            _lib.check(rc, "wvn_vit_forward_frames_pair")
        return out, chunk

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        """What get_backbone's module returns: per-patch features [B, D, G, G] (a permuted view of the
        token tensor; no copy)."""
        tok = self.forward_tokens(img)
        B = tok.shape[0]
        return tok.reshape(B, self.grid, self.grid, self.dim).permute(0, 3, 1, 2)

    __call__ = forward
