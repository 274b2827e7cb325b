"""Oracle: DINO ViT backbone (CPU, PyTorch fp32).  TEST INFRASTRUCTURE ONLY.

The reference obtains this network from the external package ``stego``
(``stego.backbones.backbone.get_backbone``; call sites
wild_visual_navigation/feature_extractor/dino_interface.py:12,45,84).  That package is not in
/root/reference, so this file restates the *published* DINO ``VisionTransformer``
(facebookresearch/dino, vision_transformer.py) as used by STEGO's featurizer:
patch-embed conv(k=P,s=P) -> [cls]+tokens + bicubic-interpolated pos-embed -> 12 pre-LN blocks
(LN eps 1e-6, qkv Linear with bias, softmax(QK^T/sqrt(dh))V, proj, +res, LN, fc1, exact GELU,
fc2, +res) -> final LN -> drop cls -> [B, D, G, G]; with ``blocks.i.ls{1,2}.gamma`` present the block is the published
DINOv2 one (LayerScale on both branch outputs, facebookresearch/dinov2).   PARITY UNPINNED (no reference golden
vectors exist for it); guarded by the HF ViTModel weight-copy cross-check in tests/.

State-dict layout = upstream DINO names, so real checkpoints load unchanged:
  patch_embed.proj.{weight[D,3,P,P],bias[D]}, cls_token[1,1,D], pos_embed[1,1+g*g,D],
  blocks.{i}.norm1.{weight,bias}, blocks.{i}.attn.qkv.{weight[3D,D],bias[3D]},
  blocks.{i}.attn.proj.{weight,bias}, blocks.{i}.norm2.{weight,bias},
  blocks.{i}.mlp.fc1.{weight[4D,D],bias}, blocks.{i}.mlp.fc2.{weight[D,4D],bias}, norm.{weight,bias}
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

ARCH = {
    # name: (embed dim, depth, heads)
    "vit_small": (384, 12, 6),
    "vit_base": (768, 12, 12),
}


def make_vit_state_dict(
    arch: str = "vit_small", patch: int = 8, pretrain_grid: int = 28, seed: int = 0, depth: Optional[int] = None
) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights in the upstream layout (trunc-normal sigma .02).

    LayerNorm affine and all biases are randomised (not 1/0) so that a kernel that forgets a
    bias / gamma / beta fails parity instead of passing by accident.
    """
    D, dflt_depth, _ = ARCH[arch]
    depth = dflt_depth if depth is None else depth
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, std=0.02):
        return torch.nn.init.trunc_normal_(torch.empty(*shape), std=std, a=-2 * std, b=2 * std, generator=g)

    def rn(*shape, std):
        return torch.randn(*shape, generator=g) * std

    sd = {
        "patch_embed.proj.weight": tn(D, 3, patch, patch, std=0.05),
        "patch_embed.proj.bias": rn(D, std=0.02),
        "cls_token": tn(1, 1, D),
        "pos_embed": tn(1, 1 + pretrain_grid * pretrain_grid, D, std=0.1),
        "norm.weight": 1.0 + rn(D, std=0.1),
        "norm.bias": rn(D, std=0.1),
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
    return sd


def make_dinov2_state_dict(arch: str = "vit_base", patch: int = 14, pretrain_grid: int = 37, seed: int = 0,
                           depth: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights in the published DINOv2 layout (facebookresearch/dinov2, vision_transformer.py / hub
    ``dinov2_vit{s,b}14``): the DINO layout plus ``blocks.i.ls{1,2}.gamma`` (LayerScale on the attention / MLP branch outputs).
    ``mask_token`` (training only) is omitted.  The architecture difference to DINO that matters at inference is LayerScale and
    the patch size; the position table is resampled by the same bicubic (grid + 0.1) / g rule."""
    sd = make_vit_state_dict(arch, patch, pretrain_grid, seed, depth)
    D = ARCH[arch][0]
    g = torch.Generator().manual_seed(seed + 77)
    for i in range(vit_depth(sd)):
        # released DINOv2 gammas are O(0.01 .. 1); random positive values so that a kernel that forgets them fails parity
        sd[f"blocks.{i}.ls1.gamma"] = 0.2 + 0.8 * torch.rand(D, generator=g)
        sd[f"blocks.{i}.ls2.gamma"] = 0.2 + 0.8 * torch.rand(D, generator=g)
    return sd


def make_vit_state_dict_heavy_tailed(
    arch: str = "vit_small", patch: int = 8, pretrain_grid: int = 28, seed: int = 0, depth: Optional[int] = None,
    outlier_channels=(17, 203), outlier_gain: float = 150.0, dof: float = 3.0, common_offset: float = 0.0,
) -> Dict[str, torch.Tensor]:
    """Synthetic weights with the two features of released DINO checkpoints the Gaussian ones above lack:
    HEAVY-TAILED linear weights (every linear weight multiplied element-wise by a Student-t(dof) / its std factor: a few entries
    per row are 5-10 sigma) and MASSIVE ACTIVATIONS -- a handful of residual channels that an early block's fc2 drives to
    ~outlier_gain times the typical magnitude for every token (its bias) and that stay there (the released models carry such
    channels from the first blocks to the last; their LayerNorm gains are small, as there).  ``common_offset`` adds the same
    constant to EVERY channel of that bias: rows with mean^2 / var in the hundreds, the case a one-pass variance (E[y^2] - mean^2,
    ADVICE r4) loses digits on.  Same layout as make_vit_state_dict."""
    sd = make_vit_state_dict(arch, patch, pretrain_grid, seed, depth)
    t = torch.distributions.StudentT(dof)
    torch.manual_seed(seed + 4243)
    for k in list(sd):
        if k.endswith(("qkv.weight", "proj.weight", "fc1.weight", "fc2.weight")) and "patch_embed" not in k:
            f = t.sample(sd[k].shape).abs().clamp(max=12.0) / 1.1      # E|t_3| ~ 1.1: the typical entry keeps its size
            sd[k] = sd[k] * f
    blk = "blocks.1." if "blocks.1.mlp.fc2.bias" in sd else "blocks.0."
    sd[blk + "mlp.fc2.bias"] += common_offset
    for c in outlier_channels:
        sd[blk + "mlp.fc2.bias"][c] = outlier_gain * (1.0 if c % 2 else -1.0)
        sd[blk + "mlp.fc2.weight"][c] *= 8.0
        for name in [k for k in sd if k.endswith(("norm1.weight", "norm2.weight")) or k == "norm.weight"]:
            sd[name][c] = 0.05 * sd[name][c]          # small LayerNorm gains on the massive channels
    return sd


def vit_depth(sd: Dict[str, torch.Tensor]) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))


def interpolate_pos_embed(pos_embed: torch.Tensor, grid: int, rule: str = "dino") -> torch.Tensor:
    """Resample the [1, 1+g*g, D] position table to a grid x grid token map.

    ``rule="dino"`` (default): the published DINO / DINOv2 behaviour (facebookresearch/dino vision_transformer.py, interpolate_pos_encoding --
    the code STEGO's backbone package copies): identity when the grid matches; otherwise bicubic on the patch part with
    ``scale_factor=(grid + 0.1) / g`` (the 0.1 keeps the output size from rounding down; since torch 1.6 the given factor -- 2.0036 at
    28 -> 56 -- is ALSO the sampling step), the class slot passed through.  ``rule="size"``: bicubic to ``size=(grid, grid)``, i.e. a sampling
    step of exactly g / grid -- what HuggingFace's ``ViTModel(..., interpolate_pos_encoding=True)`` computes today and what the DINO code
    computed under torch < 1.6 (``recompute_scale_factor``).  The two differ by up to 15 % of the table's magnitude on a random table
    (tests/test_oracle_vit.py pins the number); which one a checkpoint's authors ran cannot be decided from /root/reference, so the rule is a
    switch here and in the product (``VitBackbone(pos_embed_rule=...)``).  Done once at model-build time (weights prep), not per frame.
    """
    n_pre = pos_embed.shape[1] - 1
    g = int(round(math.sqrt(n_pre)))
    assert g * g == n_pre, "pos_embed must hold a square grid"
    if g == grid:
        return pos_embed.clone()
    D = pos_embed.shape[-1]
    table = pos_embed[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2)
    if rule == "dino":
        sf = (grid + 0.1) / g
        table = F.interpolate(table, scale_factor=(sf, sf), mode="bicubic")
    elif rule == "size":
        table = F.interpolate(table, size=(grid, grid), mode="bicubic", align_corners=False)
    else:
        raise ValueError(f"pos-embed rule {rule!r}: 'dino' or 'size'")
    assert table.shape[-1] == grid and table.shape[-2] == grid, table.shape
    table = table.permute(0, 2, 3, 1).reshape(1, grid * grid, D)
    return torch.cat([pos_embed[:, :1], table], dim=1)


def vit_tokens(
    sd: Dict[str, torch.Tensor],
    img: torch.Tensor,
    patch: int,
    heads: int,
    taps: Optional[List[torch.Tensor]] = None,
    pos_embed_rule: str = "dino",
) -> torch.Tensor:
    """Normalised image [B,3,S,S] -> final-LayerNorm'ed tokens [B, 1+G*G, D] (fp32).

    ``taps`` (optional list) receives the residual stream after patch-embed(+pos) and after
    every block, for layer-by-layer parity tests.
    """
    B, _, S, S2 = img.shape
    assert S == S2 and S % patch == 0
    G = S // patch
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    D = x.shape[1]
    x = x.flatten(2).transpose(1, 2)  # [B, G*G, D], row-major over (gy, gx)
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + interpolate_pos_embed(sd["pos_embed"], G, pos_embed_rule)
    if taps is not None:
        taps.append(x.clone())
    dh = D // heads
    scale = dh**-0.5
    for i in range(vit_depth(sd)):
        p = f"blocks.{i}."
        y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, -1, 3, heads, dh).permute(2, 0, 3, 1, 4)  # [3, B, h, N, dh]
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
        y = (att @ v).transpose(1, 2).reshape(B, -1, D)
        y = F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        x = x + (y * sd[p + "ls1.gamma"] if p + "ls1.gamma" in sd else y)  # DINOv2: LayerScale on the branch output
        y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
        y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + (y * sd[p + "ls2.gamma"] if p + "ls2.gamma" in sd else y)
        if taps is not None:
            taps.append(x.clone())
    return F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], eps=1e-6)


def vit_features(sd, img, patch: int, heads: int) -> torch.Tensor:
    """[B,3,S,S] normalised -> per-patch features [B, D, G, G] (what get_backbone's module returns,
    SURVEY.md 8a3: tokens[:,1:] reshaped [B,G,G,D] -> permute [B,D,G,G])."""
    tok = vit_tokens(sd, img, patch, heads)
    B, N, D = tok.shape
    G = img.shape[-1] // patch
    return tok[:, 1:].reshape(B, G, G, D).permute(0, 3, 1, 2).contiguous()
