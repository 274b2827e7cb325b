#!/usr/bin/env python
"""Micro-benchmark of the dominant kernels at the BASELINE configs[2] chunk shape (16 frames of 448x448:
M = 16*3144 token rows, 96 (frame, head) pairs x 3137 tokens), for A/B-ing kernel variants selected by
environment variables (WVN_GEMM_PD, WVN_ATTN_VARIANT).  Random bf16 data (never zeros: DVFS), torch
events on the launch stream, median of `--reps` after warm-up.  Prints one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wild_visual_navigation_amd import _lib, ops  # noqa: E402
from wild_visual_navigation_amd._lib import check, lib, ptr, stream  # noqa: E402


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--reps", type=int, default=15)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, ntok, ntok_s, npad, D, h = args.frames, 3137, 3144, 3200, 384, 6
    M = B * ntok_s
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(torch.bfloat16).to(dev)

    res = {"env": {k: os.environ.get(k) for k in ("WVN_GEMM_PD", "WVN_ATTN_VARIANT")}, "frames": B}
    a384, a1536 = rnd(M, 384), rnd(M, 1536)
    for name, a, N, K, epi in (("qkv_like", a384, 1152, 384, _lib.EPI_BF16), ("fc1_gelu", a384, 1536, 384, _lib.EPI_GELU_BF16),
                               ("proj_resid", a384, 384, 384, _lib.EPI_RESID_F32), ("fc2_resid", a1536, 384, 1536, _lib.EPI_RESID_F32)):
        w = rnd(N, K, scale=0.05)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.zeros(M, N, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.bfloat16, device=dev)
        med, best = timeit(lambda: ops.gemm_bf16(a, w, bias, epi, out=out), args.reps)
        res[name] = {"ms": round(med, 4), "tflops": round(2.0 * M * N * K / med / 1e9, 1), "best_tflops": round(2.0 * M * N * K / best / 1e9, 1)}
    q, k = rnd(B, h, npad, 64), rnd(B, h, npad, 64)
    vt = rnd(B, h, 64, npad)
    out = torch.empty(B * ntok, h * 64, dtype=torch.bfloat16, device=dev)
    fn = lambda: check(lib().wvn_attention_bf16(ptr(q), ptr(k), ptr(vt), ptr(out), B, h, ntok, npad, 0.125, stream()))  # noqa: E731
    med, best = timeit(fn, args.reps)
    fl = 4.0 * ntok * ntok * 64 * h * B
    res["attention"] = {"ms": round(med, 4), "tflops": round(fl / med / 1e9, 1), "best_tflops": round(fl / best / 1e9, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
