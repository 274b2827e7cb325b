#!/usr/bin/env python
"""Build, check and time the experimental attention kernels of this directory against the shipped one (GPU box):

    python scripts/experiments/run_attention_exp.py                  # both configurations of attention_pp
    python scripts/experiments/run_attention_exp.py --only occ2 --frames 64 --launches 30

Each configuration is compiled with hipcc into its own shared object (nothing here is part of libwvn_hip.so), called through
ctypes with the same buffers as wvn_attention_bf16, compared with an fp64 reference and with the shipped lazy kernel on the cases
of tests/test_gpu_attention_lazy.py, and timed with HIP events at the bench shape (frames x 6 heads x 3137 tokens)."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wild_visual_navigation_amd import ops  # noqa: E402
from wild_visual_navigation_amd._lib import check, lib, ptr, stream  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(ROOT, "_build", "exp")
CONFIGS = {
    # one wave per SIMD: cross-phase fragment prefetch, three fragments ahead, Q parked in the accumulation registers
    "occ1": ("attention_pp", ["-DPP_OCC=1", "-DPP_XPRE=1", "-DPP_LA=3"]),
    # two waves per SIMD (two workgroups per CU): 255 registers
    "occ2": ("attention_pp", ["-DPP_OCC=2", "-DPP_XPRE=0", "-DPP_LA=2"]),
}
SCALE = 0.125 * 1.4426950408889634


def build(tag, prebuilt=False):
    name, defs = CONFIGS[tag]
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(OUT, f"{name}_{tag}.hip")
    if prebuilt and os.path.exists(os.path.join(OUT, f"lib{name}_{tag}.so")):
        return load(os.path.join(OUT, f"lib{name}_{tag}.so"))
    with open(os.path.join(HERE, name + ".hip.txt")) as f, open(src, "w") as g:
        g.write(f.read())
    so = os.path.join(OUT, f"lib{name}_{tag}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mcode-object-version=5",
           "-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", os.path.join(ROOT, "wild_visual_navigation_amd", "csrc"),
           "-I", os.path.join(ROOT, "include")] + defs + [src, "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-4000:])
    return load(so)


def load(so):
    h = C.CDLL(so)
    fn = h.wvn_exp_attention_pp
    fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
    fn.restype = C.c_int
    return fn


def layouts(dev, q_in, k, v, ntok):
    B, h = q_in.shape[:2]
    npad = (ntok + 127) // 128 * 128

    def pad(t, fill):
        out = torch.full((B, h, npad, 64), fill, dtype=t.dtype)
        out[:, :, :ntok] = t
        return out

    vt = pad(v, 1e3).transpose(-1, -2)[..., ops.vt_token_order(npad)].contiguous().to(dev)
    return pad(q_in, 50.0).to(dev), pad(k, -1e3).to(dev), vt, npad


def shipped(qd, kd, vt, B, h, ntok, npad, variant=1, out=None):
    if out is None:
        out = torch.empty(B * ntok, h * 64, dtype=torch.bfloat16, device=qd.device)
    lib().wvn_debug_attention_variant(variant)
    try:
        check(lib().wvn_attention_bf16(ptr(qd), ptr(kd), ptr(vt), ptr(out), B, h, ntok, npad, 0.0, stream()))
    finally:
        lib().wvn_debug_attention_variant(-1)
    return out


def experimental(fn, qd, kd, vt, B, h, ntok, npad, out=None):
    if out is None:
        out = torch.zeros(B * ntok, h * 64, dtype=torch.bfloat16, device=qd.device)
    rc = fn(ptr(qd), ptr(kd), ptr(vt), ptr(out), B, h, ntok, ntok, npad, stream())
    if rc != 0:
        raise SystemExit(f"experimental kernel returned {rc}")
    return out


def reference(q_in, k, v):
    B, h, ntok, _ = q_in.shape
    s = (q_in.double() / SCALE) @ k.double().transpose(-1, -2) * 0.125
    return (torch.softmax(s, dim=-1) @ v.double()).permute(0, 2, 1, 3).reshape(B * ntok, h * 64)


def check_cases(fn, dev):
    worst = 0.0
    for ntok in (197, 785, 3137):
        for case in ("plain", "climbing", "overflow", "tail_spike", "negative"):
            B, h = (1, 2) if ntok > 1000 else (2, 3)
            gen = torch.Generator().manual_seed(ntok * 7 + len(case))
            q, k, v = (torch.randn(B, h, ntok, 64, generator=gen) for _ in range(3))
            if case == "climbing":
                k = k * torch.linspace(0.2, 6.0, ntok)[None, None, :, None]
            elif case == "overflow":
                k[:, :, ntok // 2] = q[:, :, ntok // 3] * 30.0
            elif case == "tail_spike":
                k[:, :, ntok - 1] = k[:, :, ntok - 1] * 8.0
            elif case == "negative":
                k[:, :, :64] = -q[:, :, :1] * 3.0
            q_in, k, v = (q * SCALE).to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
            ref = reference(q_in.float(), k.float(), v.float())
            qd, kd, vt, npad = layouts(dev, q_in, k, v, ntok)
            got = experimental(fn, qd, kd, vt, B, h, ntok, npad).float().cpu()
            ship = shipped(qd, kd, vt, B, h, ntok, npad).float().cpu()
            e_ref, e_ship = (got.double() - ref).abs().max().item(), (got - ship).abs().max().item()
            ok = torch.isfinite(got).all().item() and e_ref < 2.5e-2
            worst = max(worst, e_ref)
            print(f"  ntok {ntok:5d} {case:10s}: vs fp64 {e_ref:.2e}  vs shipped {e_ship:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
            if not ok:
                return False
    print(f"  worst error vs fp64 reference {worst:.2e}")
    return True


def timeit(call, n):
    call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=list(CONFIGS))
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--prebuilt", action="store_true", help="use _build/exp/lib*.so if present (built on the CPU box) instead of compiling")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, h, ntok = args.frames, 6, 3137
    gen = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, h, ntok, 64, generator=gen) for _ in range(3))
    qd, kd, vt, npad = layouts(dev, (q * SCALE).to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), ntok)
    flops = 4.0 * ntok * ntok * 64 * B * h
    buf = torch.empty(B * ntok, h * 64, dtype=torch.bfloat16, device=dev)   # (timed launches write here: no allocation in the loop)
    ms = timeit(lambda: shipped(qd, kd, vt, B, h, ntok, npad, out=buf), args.launches)
    print(f"shipped lazy kernel: {ms:.4f} ms per launch, {flops / ms / 1e9:.0f} TFLOP/s")
    ref_out = shipped(qd, kd, vt, B, h, ntok, npad).float()
    for tag in ([args.only] if args.only else list(CONFIGS)):
        print(f"[{tag}] {CONFIGS[tag]}", flush=True)
        fn = build(tag, args.prebuilt)
        if not args.skip_check and not check_cases(fn, dev):
            continue
        got = experimental(fn, qd, kd, vt, B, h, ntok, npad).float()
        print(f"  bench shape: max |experimental - shipped| = {(got - ref_out).abs().max().item():.2e}")
        ms = timeit(lambda: experimental(fn, qd, kd, vt, B, h, ntok, npad, out=buf), args.launches)
        print(f"  {ms:.4f} ms per launch, {flops / ms / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
