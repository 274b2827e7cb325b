#!/usr/bin/env python3
"""Round-6 error budget: WHICH products of the <= 1e-3 mode need how much (CPU emulation, torch fp32; extends scripts/error_budget.py).

Finer than the round-4 table in two directions:
  * the QKV linear is three linears (q | k | v rows of qkv.weight) with a mode each, per block;
  * correction-product formats that need NO block scale: "e5" = hi*hi in fp16 + (a_lo8 * w_hi8 + a_hi8 * w_lo8) with every 8-bit operand
    e5m2 (the top byte of an fp16 value, rounded) and the residues pre-multiplied by the constant 2^12 -- what
    v_mfma_scale_f32_32x32x64_f8f6f4 computes with format bf8 and ONE constant scale operand; "e5t" = the same with the hi8 operands
    TRUNCATED (a byte permute of the fp16 plane instead of a conversion); "c8" / "c6" = the block-scaled e4m3 / e2m3 forms of round 4.

Data sets: synthetic weights + uniform-random frame (bench.py's), the reference's real 448^2 frame, the two heavy-tailed weight sets of
tests/test_gpu_robustness.py at 224^2.

Usage:  python scripts/error_budget_r6.py [--sets synth,real,massive,offset] [--configs name;name...] [--out profiles/r06_error_budget.md]
"""
import argparse
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
from oracle import interfaces as OI, vit as ovit  # noqa: E402
import error_budget as EB  # noqa: E402


def e5m2(x, trunc=False):
    """fp32 -> e5m2 value (as fp32): the top byte of the fp16 encoding, round-to-nearest-even or truncated towards zero."""
    h = x.to(torch.float16)
    bits = h.view(torch.int16).to(torch.int32) & 0xFFFF
    if trunc:
        b = bits & 0xFF00
    else:
        b = bits + 0x7F + ((bits >> 8) & 1)
        b = b & 0xFF00
        b = torch.where((b & 0x7C00) == 0x7C00, (bits & 0x8000) | 0x7B00, b)   # saturate instead of rounding into inf
    b = b & 0xFFFF
    return (((b ^ 0x8000) - 0x8000).to(torch.int16)).view(torch.float16).float()


def product(a, w, mode, fmt="fp16"):
    if mode in ("e5", "e5t"):
        ah, wh = EB.r16(a, "fp16"), EB.r16(w, "fp16")
        sc = 4096.0
        al8, wl8 = e5m2((a - ah) * sc), e5m2((w - wh) * sc)
        ah8, wh8 = e5m2(ah, trunc=mode == "e5t"), e5m2(wh)
        corr = (al8 @ wh8.transpose(-1, -2) + ah8 @ wl8.transpose(-1, -2)) / sc
        return ah @ wh.transpose(-1, -2) + corr
    return EB.product(a, w, mode, fmt)


def per_block(m, depth):
    return m if isinstance(m, (list, tuple)) else [m] * depth


def vit_tokens_emulated(sd, img, patch, heads, modes):
    """modes: dict over {patch, q, k, v, qk, pv, proj, fc1, fc2}; every value a mode string or a per-block list."""
    B, _, S, _ = img.shape
    G = S // patch
    D = sd["patch_embed.proj.weight"].shape[0]
    depth = ovit.vit_depth(sd)
    M = {k: per_block(v, depth) for k, v in modes.items()}
    cols = F.unfold(img, kernel_size=patch, stride=patch).transpose(1, 2)
    wpe = sd["patch_embed.proj.weight"].reshape(D, -1)
    x = product(cols, wpe, M["patch"][0]) + sd["patch_embed.proj.bias"]
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + ovit.interpolate_pos_embed(sd["pos_embed"], G)
    dh = D // heads
    qscale = dh**-0.5 * math.log2(math.e)
    for i in range(depth):
        p = f"blocks.{i}."
        y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
        W, bqkv = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        parts = [product(y, W[j * D:(j + 1) * D], M[n][i]) + bqkv[j * D:(j + 1) * D] for j, n in enumerate(("q", "k", "v"))]
        q, k, v = (t.reshape(B, -1, heads, dh).permute(0, 2, 1, 3) for t in parts)
        q = q * qscale
        outs = []
        for b in range(B):
            s = EB.product(q[b], k[b], M["qk"][i], "fp16")
            pr = torch.exp2(s - s.amax(dim=-1, keepdim=True))
            m = M["pv"][i]
            if m == "f32":
                o = pr @ v[b]
                den = pr.sum(-1, keepdim=True)
            else:
                o = EB.product(pr, v[b].transpose(-1, -2), m, "fp16")
                den = (EB.r16(pr, "fp16") if m in ("h", "w") else pr).sum(-1, keepdim=True)
            outs.append((o / den).transpose(0, 1).reshape(-1, D))
        y = torch.stack(outs)
        y = product(y, sd[p + "attn.proj.weight"], M["proj"][i]) + sd[p + "attn.proj.bias"]
        x = x + y
        y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
        hdn = F.gelu(product(y, sd[p + "mlp.fc1.weight"], M["fc1"][i]) + sd[p + "mlp.fc1.bias"])
        y = product(hdn, sd[p + "mlp.fc2.weight"], M["fc2"][i]) + sd[p + "mlp.fc2.bias"]
        x = x + y
    return F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], eps=1e-6)


def base(lin, q6="a"):
    """The shipped mixed mode with the linears in `lin`: q split in the first six blocks, P V single."""
    m = {f: lin for f in ("patch", "q", "k", "v", "proj", "fc1", "fc2")}
    m["qk"] = [q6] * 6 + ["h"] * 6
    m["pv"] = "h"
    return m


def configs():
    c = []
    c.append(("mixed as shipped (linears x3, q split in blocks 0-5)", base("x3")))
    for lin in ("c8", "c6", "e5", "e5t"):
        c.append((f"linears {lin}", base(lin)))
    # the QKV linear: which of its three parts can run on single fp16 operands?
    for name, kw in (("k | v linears single fp16", {"k": "h", "v": "h"}), ("k linear single", {"k": "h"}), ("v linear single", {"v": "h"}),
                     ("q linear single in blocks 6-11", {"q": ["x3"] * 6 + ["h"] * 6}),
                     ("k | v single, q single in blocks 6-11", {"k": "h", "v": "h", "q": ["x3"] * 6 + ["h"] * 6}),
                     ("q | k | v single everywhere", {"q": "h", "k": "h", "v": "h"})):
        m = base("x3")
        m.update(kw)
        c.append((name + " (rest x3)", m))
    for lin in ("c8", "e5"):
        m = base(lin)
        m.update({"k": "h", "v": "h"})
        c.append((f"linears {lin}, k | v single", m))
        m = dict(m)
        m["patch"] = "x3"
        c.append((f"linears {lin}, k | v single, patch x3", m))
    return c


def data_sets(which):
    out = []
    torch.manual_seed(0)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    if "synth" in which:
        sd = ovit.make_vit_state_dict("vit_small", 8, 28, seed=0)
        img = (torch.rand(1, 3, 448, 448, generator=torch.Generator().manual_seed(1)) - mean) / std
        out.append(("synth", sd, img))
    if "real" in which:
        sd = ovit.make_vit_state_dict("vit_small", 8, 28, seed=0)
        u8 = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "graph_img_448.pt"))["frame_u8"]
        out.append(("real", sd, OI.normalize((u8.float() / 255)[None])))
    for kind in ("massive", "offset"):
        if kind in which:
            sd = ovit.make_vit_state_dict_heavy_tailed("vit_small", 8, 28, seed=3, common_offset=40.0 if kind == "offset" else 0.0,
                                                       outlier_gain=150.0 if kind == "massive" else 60.0)
            img = torch.rand(12, 3, 224, 224, generator=torch.Generator().manual_seed(1))[:3]
            out.append((kind, sd, OI.normalize(img)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="synth,real,massive,offset")
    ap.add_argument("--configs", default="")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    cfgs = configs()
    if args.configs:
        keep = args.configs.split(";")
        cfgs = [c for c in cfgs if any(k in c[0] for k in keep)]
    sets = data_sets(args.sets.split(","))
    table = {name: [] for name, _ in cfgs}
    with torch.no_grad():
        for sname, sd, img in sets:
            ref = ovit.vit_tokens(sd, img, 8, 6)
            for name, modes in cfgs:
                t0 = time.time()
                e = vit_tokens_emulated(sd, img, 8, 6, modes) - ref
                table[name].append((e.abs().max().item(), e.pow(2).mean().sqrt().item()))
                print(f"[{sname:8s}] {name:60s} max {table[name][-1][0]:.2e} rms {table[name][-1][1]:.2e} ({time.time() - t0:.0f} s)", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write("# Round-6 error budget of the <= 1e-3 mode (CPU emulation, `scripts/error_budget_r6.py`): max abs / rms token error against the fp32 oracle\n\n")
            f.write("| configuration | " + " | ".join(s[0] for s in sets) + " |\n|---|" + "---|" * len(sets) + "\n")
            for name, _ in cfgs:
                f.write(f"| {name} | " + " | ".join(f"{mx:.2e} / {rms:.2e}" for mx, rms in table[name]) + " |\n")


if __name__ == "__main__":
    main()
