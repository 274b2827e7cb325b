#!/usr/bin/env python3
"""Per-kernel-family error budget of the 16-bit speed path (VERDICT r3 "Next round" item 1).

CPU emulation (torch fp32): the oracle's ViT (`oracle/vit.py`) with the OPERANDS of one kernel family at a time rounded the way the
HIP kernels round them (fp16 / bf16 significands, fp32 accumulation), everything else exact.  Reports, per configuration, the
max-abs / rms error of the final-LayerNorm'ed tokens at 448 x 448 x 12 blocks against the all-fp32 oracle, i.e. what
`bench.py`'s `max_abs_tokens` measures on the GPU.  The emulation is validated against the GPU's own numbers: "all fp16" must land
near the 4.5e-3 / 7.4e-4 the fp16 kernels measure (`profiles/r03g_bench_default.json`), "all split" near the exact mode's 5e-5.

Families (= the operand pairs of one MFMA product):
  patch  patch-embedding GEMM                       (pixels, conv weight)
  qkv    LayerNorm 1 -> QKV linear                  (normalised rows, qkv.weight)
  qk     S = Q K^T                                  (q pre-scaled by scale * log2 e as the QKV epilogue stores it, k)
  pv     O = P V                                    (probabilities relative to the row maximum, v)
  proj   attention projection                       (attention rows, proj.weight)
  fc1    LayerNorm 2 -> fc1                         (normalised rows, fc1.weight)
  fc2    GELU -> fc2                                (hidden activations, fc2.weight)

Operand modes per family:
  f32    exact
  h      one 16-bit value per operand ("fp16" | "bf16")
  a      activation operand split hi + lo (2 MFMAs), weight single 16-bit
  w      weight operand split, activation single
  x3     both split, hi*hi + hi*lo + lo*hi (3 MFMAs)  -- the shipped exact mode uses bf16 planes
  c8/c6/c4  hi*hi in fp16 + the two correction products (a_lo * w, a * w_lo) with BOTH operands in an MX format (e4m3 / e2m3 /
         e2m1 elements, one power-of-two scale per 32 elements along K) -- `v_mfma_scale_f32_32x32x64_f8f6f4` runs those at 2x / 4x /
         4x the fp16 rate, so the product costs 2 / 1.5 / 1.5 fp16 MFMAs instead of 3.

Usage:  python scripts/error_budget.py [--frames 2] [--size 448] [--out profiles/r04a_error_budget.md] [--weights synthetic|peaked]
"""
import argparse
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import vit as ovit  # noqa: E402

FAMILIES = ["patch", "qkv", "qk", "pv", "proj", "fc1", "fc2"]
MFMA_COST = {"f32": None, "h": 1.0, "a": 2.0, "w": 2.0, "x3": 3.0, "c8": 2.0, "c6": 1.5, "c4": 1.5, "q8r": 0.5, "q8m": 0.5, "q8x3": 1.5, "q8a": 1.0, "q8w": 1.0}
# share of the backbone's multiply-adds per family at 448^2 (315.1 GFLOP per frame)
FLOP_SHARE = {"patch": 0.0015, "qkv": 0.1057, "qk": 0.2878, "pv": 0.2878, "proj": 0.0352, "fc1": 0.141, "fc2": 0.141}


def r16(x, fmt):
    return x.to(torch.float16 if fmt == "fp16" else torch.bfloat16).float()


def mx_quant(x, ebits, mbits, block=32):
    """Round to an MX element format (sign, ebits, mbits; no inf / nan) with one power-of-two scale per `block` elements of the
    last dimension (the K dimension of the product), scale = 2^(floor(log2 amax) - emax_elem): the OCP MX rule."""
    shp = x.shape
    K = shp[-1]
    pad = (-K) % block
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(-1, block)
    amax = xb.abs().amax(dim=1, keepdim=True).clamp_min(1e-38)
    if (ebits, mbits) == (4, 3):
        emax, vmax = 8, 448.0
    elif (ebits, mbits) == (2, 3):
        emax, vmax = 2, 7.5
    elif (ebits, mbits) == (2, 1):
        emax, vmax = 2, 6.0
    else:
        raise ValueError
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = xb / scale
    # element rounding: exponent of |y| clamped to the format's normal range, mantissa to mbits (round to nearest even)
    bias = (1 << (ebits - 1)) - 1
    emin = 1 - bias
    e = torch.floor(torch.log2(y.abs().clamp_min(1e-38))).clamp_min(emin)
    q = torch.exp2(e - mbits)
    y = (torch.round(y / q) * q).clamp(-vmax, vmax)
    out = (y * scale).reshape(*shp[:-1], K + pad)
    return out[..., :K] if pad else out


MX = {"c8": (4, 3), "c6": (2, 3), "c4": (2, 1)}


def mx_quant_elem(y):
    """e4m3 element rounding of values already scaled into [-448, 448] (no block scale)."""
    e = torch.floor(torch.log2(y.abs().clamp_min(1e-38))).clamp_min(-6.0)
    q = torch.exp2(e - 3)
    return (torch.round(y / q) * q).clamp(-448.0, 448.0)


def product(a, w, mode, fmt, kdim_last_w=True):
    """a [..., K] x w [N, K]^T with the operand handling of `mode`; fp32 accumulation (torch's, order differs from the MFMA's:
    ~1e-7 relative, far below what is measured here)."""
    if mode == "f32":
        return a @ w.transpose(-1, -2)
    ah, wh = r16(a, fmt), r16(w, fmt)
    if mode == "h":
        return ah @ wh.transpose(-1, -2)
    al, wl = r16(a - ah, fmt), r16(w - wh, fmt)
    if mode == "a":
        return ah @ wh.transpose(-1, -2) + al @ wh.transpose(-1, -2)
    if mode == "w":
        return ah @ wh.transpose(-1, -2) + ah @ wl.transpose(-1, -2)
    if mode == "x3":
        return ah @ wh.transpose(-1, -2) + (al @ wh.transpose(-1, -2) + ah @ wl.transpose(-1, -2))
    if mode in ("q8r", "q8m", "q8x3", "q8a", "q8w"):
        # fp8 e4m3 operands (the --fp8 legs, BASELINE configs[4]).  q8r: per-ROW scales (activation: per token, weight: per output channel:
        # what csrc/fp8.hip does); q8m: one power-of-two scale per 32 elements along K (OCP MX, v_mfma_scale_f32_32x32x64_f8f6f4);
        # q8x3 / q8a: hi + lo e4m3 planes of both operands (three fp8 MFMAs) / of the activation only (two)
        def q_row(x):
            sc = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
            return mx_quant_elem(x / sc) * sc
        qa, qw = (q_row, q_row) if mode != "q8m" else ((lambda x: mx_quant(x, 4, 3)),) * 2
        a8, w8 = qa(a), qw(w)
        if mode in ("q8r", "q8m"):
            return a8 @ w8.transpose(-1, -2)
        a8l = qa(a - a8) if mode != "q8w" else None
        if mode == "q8a":
            return (a8 + a8l) @ w8.transpose(-1, -2)
        w8l = qw(w - w8)
        if mode == "q8w":
            return a8 @ (w8 + w8l).transpose(-1, -2)
        return a8 @ w8.transpose(-1, -2) + (a8l @ w8.transpose(-1, -2) + a8 @ w8l.transpose(-1, -2))
    if mode in MX:
        eb, mb = MX[mode]
        a_lo, w_lo = a - ah, w - wh  # the fp32 residues the kernel has in registers / the weight prep has offline
        corr = mx_quant(a_lo, eb, mb) @ mx_quant(w, eb, mb).transpose(-1, -2) + mx_quant(a, eb, mb) @ mx_quant(w_lo, eb, mb).transpose(-1, -2)
        return ah @ wh.transpose(-1, -2) + corr
    raise ValueError(mode)


def vit_tokens_emulated(sd, img, patch, heads, modes, fmt, center_k=False, qk_mode_by_block=None, tail_blocks=0, tail_modes=None):
    """tail_blocks / tail_modes: the LAST `tail_blocks` blocks run with `tail_modes` (dict family -> mode) instead of `modes`."""
    """oracle.vit.vit_tokens with per-family operand modes (dict family -> mode)."""
    B, _, S, _ = img.shape
    G = S // patch
    D = sd["patch_embed.proj.weight"].shape[0]
    cols = F.unfold(img, kernel_size=patch, stride=patch).transpose(1, 2)  # [B, G*G, 3*P*P], (c, py, px) like conv weight
    wpe = sd["patch_embed.proj.weight"].reshape(D, -1)
    x = product(cols, wpe, modes["patch"], fmt) + sd["patch_embed.proj.bias"]
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + ovit.interpolate_pos_embed(sd["pos_embed"], G)
    dh = D // heads
    qscale = dh**-0.5 * math.log2(math.e)
    depth_ = ovit.vit_depth(sd)
    modes_all = modes
    for i in range(depth_):
        modes = tail_modes if (tail_modes is not None and i >= depth_ - tail_blocks) else modes_all
        p = f"blocks.{i}."
        y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
        qkv = product(y, sd[p + "attn.qkv.weight"], modes["qkv"], fmt) + sd[p + "attn.qkv.bias"]
        qkv = qkv.reshape(B, -1, 3, heads, dh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * qscale, qkv[1], qkv[2]
        if center_k:   # softmax is invariant to a per-query shift of the scores: q . (k - mean_j k) differs from q . k by a row constant
            k = k - k.mean(dim=2, keepdim=True)
        outs = []
        for b in range(B):  # per frame: the [h, N, N] score block is 236 MB at 448^2
            s = product(q[b], k[b], modes["qk"] if qk_mode_by_block is None else qk_mode_by_block[i], fmt)  # log2 domain
            pr = torch.exp2(s - s.amax(dim=-1, keepdim=True))
            m = modes["pv"]
            if m == "f32":
                o = pr @ v[b]
                den = pr.sum(-1, keepdim=True)
            else:
                # the kernel's row sums add the ROUNDED probabilities (v_dot2c on the packed pairs): numerator and denominator see
                # the same values
                o = product(pr, v[b].transpose(-1, -2), m, fmt)
                den = (r16(pr, fmt) if m in ("h", "w") else pr).sum(-1, keepdim=True)
            outs.append((o / den).transpose(0, 1).reshape(-1, D))
        y = torch.stack(outs)
        y = product(y, sd[p + "attn.proj.weight"], modes["proj"], fmt) + sd[p + "attn.proj.bias"]
        x = x + y
        y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
        hdn = F.gelu(product(y, sd[p + "mlp.fc1.weight"], modes["fc1"], fmt) + sd[p + "mlp.fc1.bias"])
        y = product(hdn, sd[p + "mlp.fc2.weight"], modes["fc2"], fmt) + sd[p + "mlp.fc2.bias"]
        x = x + y
    return F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], eps=1e-6)


def peaked(sd, gain=1.4):
    """A second synthetic weight set whose attention is NOT nearly uniform (trunc-normal sigma .02 - .06 weights give softmax rows
    close to 1 / N: the attention branch then hardly reaches the tokens and its rounding looks free).  q / k weights scaled by
    `gain`: score spread grows by gain^2 (1.4: logit standard deviation 1.4 -> 2.7; at 6 the network is chaotic -- hard attention
    flips under the fp32 summation-order noise alone and even the all-f32 emulation lands 2.0 away from the oracle)."""
    sd = {k: v.clone() for k, v in sd.items()}
    D = sd["norm.weight"].shape[0]
    for i in range(ovit.vit_depth(sd)):
        sd[f"blocks.{i}.attn.qkv.weight"][: 2 * D] *= gain
    return sd


def cost(modes):
    c = 0.0
    for f in FAMILIES:
        c += FLOP_SHARE[f] * (MFMA_COST[modes[f]] or 8.0)
    return c


def real_frame(args, sd):
    """Which operand of the attention products carries the error on a REAL image (assets/graph/img.png of the reference): linears split
    (x3) throughout, Q K^T / P V varied.  Found with this table: the query's rounding is the coherent error (the same direction against
    every key of its row), the key's averages out over the row -- the reason attention_bf16.hip takes q as two planes in WVN_PREC_MIX."""
    from oracle import interfaces as OI
    u8 = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "graph_img_448.pt"))["frame_u8"]
    img = OI.normalize((u8.float() / 255)[None])
    rows = []
    with torch.no_grad():
        ref = ovit.vit_tokens(sd, img, 8, 6)
        base = {f: "x3" for f in FAMILIES}
        for name, qk, pv, kw in (("linears x3, attention single (WVN_PREC_MIX before QSPLIT)", "h", "h", {}),
                                 ("only Q K^T single", "h", "x3", {}), ("only P V single", "x3", "h", {}),
                                 ("Q K^T: q split (2 MFMAs), k single; P V single  (= WVN_PREC_MIX)", "a", "h", {}),
                                 ("Q K^T: k split (2 MFMAs), q single; P V single", "w", "h", {}),
                                 ("attention single, K centred per head (softmax-invariant shift)", "h", "h", {"center_k": True})):
            m = dict(base)
            m["qk"], m["pv"] = qk, pv
            e = vit_tokens_emulated(sd, img, 8, 6, m, args.fmt, **kw) - ref
            rows.append((name, e.abs().max().item(), e.pow(2).mean().sqrt().item()))
            print(f"{name:80s} max {rows[-1][1]:.2e} rms {rows[-1][2]:.2e}", flush=True)
        if args.fp16_tail:   # (round 5) the LAST n blocks entirely on the single-fp16 speed kernels, the rest = the mixed mode with q split in the first six
            depth = ovit.vit_depth(sd)
            allh = {f: "h" for f in FAMILIES}
            for n in (int(v) for v in args.fp16_tail.split(";")):
                by_block = ["a" if i < 6 else "h" for i in range(depth)]
                m = dict(base)
                m["qk"], m["pv"] = "h", "h"
                e = vit_tokens_emulated(sd, img, 8, 6, m, args.fmt, qk_mode_by_block=by_block, tail_blocks=n, tail_modes=allh) - ref
                name = f"mixed (q split in the first 6), the LAST {n} blocks all single fp16"
                rows.append((name, e.abs().max().item(), e.pow(2).mean().sqrt().item()))
                print(f"{name:80s} max {rows[-1][1]:.2e} rms {rows[-1][2]:.2e}", flush=True)
        if args.qsplit_blocks:   # which blocks need the two-plane q (round 5): q split in a subset of the blocks, single elsewhere
            depth = ovit.vit_depth(sd)
            for spec in args.qsplit_blocks.split(";"):
                lo, hi = (int(v) for v in spec.split("-"))
                by_block = ["a" if lo <= i < hi else "h" for i in range(depth)]
                m = dict(base)
                m["qk"], m["pv"] = "h", "h"
                e = vit_tokens_emulated(sd, img, 8, 6, m, args.fmt, qk_mode_by_block=by_block) - ref
                name = f"q split in blocks [{lo}, {hi}) only; P V single"
                rows.append((name, e.abs().max().item(), e.pow(2).mean().sqrt().item()))
                print(f"{name:80s} max {rows[-1][1]:.2e} rms {rows[-1][2]:.2e}", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write("# Attention operands on the reference's real 448^2 frame (assets/graph/img.png), 12 blocks, linears split throughout\n\n"
                    "CPU emulation (`scripts/error_budget.py --real-frame`); the GPU measured 1.009e-3 for the first row and 8.9e-5 for the fourth "
                    "(`tests/test_gpu_x3.py::test_reference_448_frame_through_the_full_backbone`).  In `product(a, w)` of Q K^T the activation operand "
                    "is q, the weight operand k.\n\n| configuration | max abs | rms |\n|---|---|---|\n")
            for name, mx, rms in rows:
                f.write(f"| {name} | {mx:.2e} | {rms:.2e} |\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--fmt", default="fp16")
    ap.add_argument("--weights", default="synthetic", choices=["synthetic", "peaked"])
    ap.add_argument("--out", default="")
    ap.add_argument("--configs", default="")
    ap.add_argument("--gain", type=float, default=1.4)
    ap.add_argument("--fp8", action="store_true", help="the fp8 table: e4m3 linears under per-row / MX block scales / hi + lo planes")
    ap.add_argument("--real-frame", action="store_true", help="the reference's one real 448^2 frame (tests/golden/graph_img_448.pt) and the "
                    "attention-operand variants of the mixed mode instead of the synthetic frames and the family table")
    ap.add_argument("--fp16-tail", default="", help="with --real-frame: ';'-separated counts n: the last n blocks on single fp16 operands")
    ap.add_argument("--qsplit-blocks", default="", help="with --real-frame: ';'-separated block ranges lo-hi in which q is split")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    sd = ovit.make_vit_state_dict("vit_small", 8, 28, seed=0, depth=args.depth)
    if args.weights == "peaked":
        sd = peaked(sd, args.gain)
    if args.real_frame:
        return real_frame(args, sd)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(args.frames, 3, args.size, args.size, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img = (img - mean) / std
    with torch.no_grad():
        t0 = time.time()
        ref = ovit.vit_tokens(sd, img, 8, 6)
        print(f"fp32 oracle: {time.time() - t0:.1f} s; token rms {ref.pow(2).mean().sqrt():.3f}, max {ref.abs().max():.2f}", flush=True)
        # attention entropy as a witness of how peaked the rows are is not needed here; the table is the result
        allh = {f: "h" for f in FAMILIES}
        configs = [("emulation self-check: all f32", {f: "f32" for f in FAMILIES}), ("all single " + args.fmt, dict(allh))]
        for f in FAMILIES:  # one family exact, the rest single 16-bit
            m = dict(allh)
            m[f] = "f32"
            configs.append((f"all {args.fmt}, {f} exact", m))
        for f in FAMILIES:  # one family single 16-bit, the rest exact: the family's own contribution
            m = {ff: "f32" for ff in FAMILIES}
            m[f] = "h"
            configs.append((f"only {f} in {args.fmt}", m))
        for mode in ("a", "w", "x3", "c8", "c6", "c4"):
            configs.append((f"all {mode}", {f: mode for f in FAMILIES}))
        # mixes: the linears compensated, attention products single
        for mode in ("x3", "c8", "c4"):
            m = {f: mode for f in FAMILIES}
            m["qk"] = "h"
            m["pv"] = "h"
            configs.append((f"linears {mode}, attention single", m))
            m = dict(m)
            m["qk"] = mode
            configs.append((f"linears + qk {mode}, pv single", m))
        if args.fp8:   # the fp8 legs (configs[4]): linears in e4m3, attention products in the 16-bit format
            configs = [("all single " + args.fmt, dict(allh))]
            for mode, what in (("q8r", "e4m3, per-row scales (shipped)"), ("q8m", "e4m3, MX block-32 scales"), ("q8a", "e4m3, activation hi + lo (2 MFMAs)"), ("q8w", "e4m3, weight hi + lo (2 MFMAs)"),
                               ("q8x3", "e4m3, both operands hi + lo (3 MFMAs)")):
                m = {f: mode for f in FAMILIES}
                m["qk"] = "h"
                m["pv"] = "h"
                configs.append((f"linears {what}, attention {args.fmt}", m))
        if args.configs:
            keep = set(args.configs.split(";"))
            configs = [c for c in configs if c[0] in keep]
        rows = []
        for name, modes in configs:
            t0 = time.time()
            tok = vit_tokens_emulated(sd, img, 8, 6, modes, args.fmt)
            err = (tok - ref)
            mx, rms = err.abs().max().item(), err.pow(2).mean().sqrt().item()
            rel = rms / ref.pow(2).mean().sqrt().item()
            rows.append((name, mx, rms, rel, cost(modes)))
            print(f"{name:44s} max {mx:.2e}  rms {rms:.2e}  rel-L2 {rel:.2e}  mfma cost {cost(modes):.2f}  ({time.time() - t0:.0f} s)", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write((f"# fp8 linears: what the scales can and cannot buy ({args.weights} weights, ViT-S/8, {args.frames} frame(s) at {args.size}^2, "
                     f"{args.depth} blocks; attention products in {args.fmt})\n\n" if args.fp8 else
                     f"# Error budget of the 16-bit path by kernel family ({args.weights} weights, {args.frames} frames at {args.size}^2, "
                     f"{args.depth} blocks, operand format {args.fmt})\n\n"))
            f.write("CPU emulation (`scripts/error_budget.py`): operands of the named products rounded as the kernels round them, fp32 "
                    "accumulation; error of the final-LayerNorm'ed tokens against the fp32 oracle.  `mfma cost` = matrix-pipe work "
                    "relative to the all-single-16-bit path (FLOP shares of DESIGN section 4).\n\n")
            f.write("| configuration | max abs | rms | rel-L2 | mfma cost |\n|---|---|---|---|---|\n")
            for name, mx, rms, rel, c in rows:
                f.write(f"| {name} | {mx:.2e} | {rms:.2e} | {rel:.2e} | {c:.2f} |\n")


if __name__ == "__main__":
    main()
