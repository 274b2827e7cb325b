#!/usr/bin/env python
"""The A-stationary fp8 kernel (csrc/gemm_a768_fp8.hip) against the tiled one (csrc/gemm_fp8.hip) on the K = 768 linears of ViT-Base at the rows of configs[4]'s
per-GPU share (16 frames x 1370 tokens, padded to 1376): HIP-event time per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from wild_visual_navigation_amd import _lib, ops  # noqa: E402
from wild_visual_navigation_amd.backbone import pack_a768_fp8  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ntok_s, npad, heads = 1376, 1408, 12
M = B * ntok_s
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
a = torch.randn(M, 768, generator=g).to(dev)
aq, sa = ops.quantize_rows_fp8(a)


def timed(call, label, N):
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{label}: {ms * 1e3:.1f} us = {2.0 * M * N * 768 / ms / 1e9:.0f} TFLOP/s", flush=True)


for N, epi, name in ((768, _lib.EPI_RESID_F32, "projection (fp32 +=)"), (2304, _lib.EPI_QKV, "q | k | v^T"), (3072, _lib.EPI_GELU_BF16, "fc1 + GELU")):
    w = (torch.randn(N, 768, generator=g) * 0.05).to(dev)
    wq, sw = ops.quantize_rows_fp8(w)
    wp = pack_a768_fp8(wq)
    bias = torch.zeros(N, device=dev)
    per = B * heads * npad * 64
    buf = torch.zeros(3 * per, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(M, N, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.bfloat16, device=dev)
    st = _lib.stream()
    timed(lambda: _lib.check(lib.wvn_gemm_a768_fp8(aq.data_ptr(), 768, wp.data_ptr(), sa.data_ptr(), sw.data_ptr(), bias.data_ptr(), 0, out.data_ptr(), N, M, N, epi,
                                                   buf.data_ptr(), buf.data_ptr() + 2 * per, buf.data_ptr() + 4 * per, heads, npad, ntok_s, 0.18, st), "a768"), f"A-stationary {name}", N)
    if epi != _lib.EPI_QKV:
        timed(lambda: ops.gemm_fp8(aq, sa, wq, sw, bias, epi, out=out), f"tiled        {name}", N)
