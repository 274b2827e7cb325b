#!/usr/bin/env python
"""Time the A-stationary split-operand K = 384 GEMM on its own (GPU box) and print where its slice periods go (instrumented build):
fc1 (N = 1536, erf GELU, plane output) and the projection (N = 384, fp32 residual update) at the rows of 128 frames; checks the
result against torch fp32 on a sample of rows."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from wild_visual_navigation_amd import _lib  # noqa: E402
from wild_visual_navigation_amd.backbone import split_planes  # noqa: E402
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 3152
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
a = torch.randn(M, 384, generator=g).to(dev)
ap = split_planes(a)
for name, N, epi in (("fc1 (GELU planes)", 1536, 1), ("proj (fp32 residual)", 384, 4), ("qk-sized (GELU planes, N = 768)", 768, 1)):
    w = (torch.randn(N, 384, generator=g) * 0.05).to(dev)
    wp = split_planes(w)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    if epi == 1:
        c = torch.empty(2, M, N, dtype=torch.bfloat16, device=dev)
        args = lambda dbg: (ap[0].data_ptr(), ap[1].data_ptr(), 384, wp[0].data_ptr(), wp[1].data_ptr(), bias.data_ptr(), c[0].data_ptr(), c[1].data_ptr(), N, M, N, epi, dbg, _lib.stream())
    else:
        c = torch.zeros(M, N, dtype=torch.float32, device=dev)
        args = lambda dbg: (ap[0].data_ptr(), ap[1].data_ptr(), 384, wp[0].data_ptr(), wp[1].data_ptr(), bias.data_ptr(), c.data_ptr(), 0, N, M, N, epi, dbg, _lib.stream())
    _lib.check(lib.wvn_debug_gemm_a384_x3(*args(0)), "a384_x3")
    rows = torch.cat([torch.arange(0, min(M, 4096), device=dev), torch.arange(0, M, max(1, M // 512), device=dev)])   # every row of the first blocks + a sample
    want = a[rows].double() @ w.double().T + bias.double()
    if epi == 1:
        want = torch.nn.functional.gelu(want)
        got = (c[0][rows].double() + c[1][rows].double())
    else:
        got = c[rows].double()
    err = (got - want).abs().max().item()
    for _ in range(3):
        lib.wvn_debug_gemm_a384_x3(*args(0))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.wvn_debug_gemm_a384_x3(*args(0))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tf = 2.0 * M * N * 384 * 3 / ms / 1e9
    dbg = torch.zeros(256 * 4 * 4, dtype=torch.int64, device=dev)
    lib.wvn_debug_gemm_a384_x3(*args(dbg.data_ptr()))
    torch.cuda.synchronize()
    d = dbg.reshape(256, 4, 4).double().mean(dim=(0, 1))
    slices = (M / 128) * (N / 64) * 3 / 256
    print(f"{name}: {ms * 1e3:.0f} us, {tf:.0f} TFLOP/s issued, max err {err:.2e}; per slice period ({slices:.0f} per wave): wait+barrier {d[0] / slices:.0f}, "
          f"DMA issue {d[1] / slices:.0f}, steps {d[2] / slices:.0f} (MFMA floor 1536), total {d[3] / slices:.0f} cycles", flush=True)

# ---- the row-panel N = 384 kernel (gemm_n384_x3.hip): fc2 (K = 1536) and the projection (K = 384) ----
for name, K in (("fc2 row panel (K = 1536)", 1536), ("proj row panel (K = 384)", 384)):
    a2 = torch.randn(M, K, generator=g).to(dev) if K != 384 else a
    a2p = split_planes(a2)
    w = (torch.randn(384, K, generator=g) * 0.03).to(dev)
    wp = split_planes(w)
    bias = (torch.randn(384, generator=g) * 0.1).to(dev)
    c = torch.zeros(M, 384, dtype=torch.float32, device=dev)
    args = lambda dbg: (a2p[0].data_ptr(), a2p[1].data_ptr(), K, wp[0].data_ptr(), wp[1].data_ptr(), bias.data_ptr(), 0, c.data_ptr(), 384, M, K, dbg, _lib.stream())
    _lib.check(lib.wvn_debug_gemm_n384_x3(*args(0)), "n384_x3")
    rows = torch.cat([torch.arange(0, min(M, 4096), device=dev), torch.arange(0, M, max(1, M // 512), device=dev)])
    want = a2[rows].double() @ w.double().T + bias.double()
    err = (c[rows].double() - want).abs().max().item()
    for _ in range(3):
        lib.wvn_debug_gemm_n384_x3(*args(0))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.wvn_debug_gemm_n384_x3(*args(0))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tf = 2.0 * M * 384 * K * 3 / ms / 1e9
    dbg = torch.zeros(256 * 4 * 4, dtype=torch.int64, device=dev)
    lib.wvn_debug_gemm_n384_x3(*args(dbg.data_ptr()))
    torch.cuda.synchronize()
    d = dbg.reshape(256, 4, 4).double().mean(dim=(0, 1))
    slices = (M / 128) * (K / 32) / 256
    print(f"{name}: {ms * 1e3:.0f} us, {tf:.0f} TFLOP/s issued, max err {err:.2e}; per slice ({slices:.0f} per wave): wait+barrier {d[0] / slices:.0f}, "
          f"k-steps {d[1] / slices:.0f} (MFMA floor 2304), epilogue {d[2] / slices:.0f}, total {d[3] / slices:.0f} cycles", flush=True)
    del a2p, c

# ---- the fragment-major split-operand MLP: fc1 (EPI_GELU_FRAG) -> fc2 (AFRAG) ----
from wild_visual_navigation_amd.backbone import pack_fc2_fragment_major  # noqa: E402
F = 1536
w1 = (torch.randn(F, 384, generator=g) * 0.05).to(dev); b1 = (torch.randn(F, generator=g) * 0.1).to(dev)
w2 = (torch.randn(384, F, generator=g) * 0.03).to(dev); b2 = (torch.randn(384, generator=g) * 0.1).to(dev)
w1p, w2p = split_planes(w1), pack_fc2_fragment_major(w2)
Mp = (M + 31) // 32 * 32
hid = torch.empty(2, Mp * F, dtype=torch.bfloat16, device=dev)
x = torch.zeros(M, 384, dtype=torch.float32, device=dev)
margs = lambda d1, d2: (ap[0].data_ptr(), ap[1].data_ptr(), w1p[0].data_ptr(), w1p[1].data_ptr(), b1.data_ptr(), hid[0].data_ptr(), hid[1].data_ptr(),
                        w2p.data_ptr(), b2.data_ptr(), x.data_ptr(), M, F, d1, d2, _lib.stream())
_lib.check(lib.wvn_debug_mlp_x3_frag(*margs(0, 0)), "mlp_x3_frag")
rows = torch.cat([torch.arange(0, min(M, 4096), device=dev), torch.arange(0, M, max(1, M // 512), device=dev)])
want = torch.nn.functional.gelu(a[rows].double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()
print(f"fragment-major MLP: max err {(x[rows].double() - want).abs().max().item():.2e} (|want| max {want.abs().max().item():.2f})", flush=True)
for _ in range(2):
    lib.wvn_debug_mlp_x3_frag(*margs(0, 0))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.wvn_debug_mlp_x3_frag(*margs(0, 0))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
d1 = torch.zeros(256 * 16, dtype=torch.int64, device=dev); d2 = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
lib.wvn_debug_mlp_x3_frag(*margs(d1.data_ptr(), d2.data_ptr()))
torch.cuda.synchronize()
t1 = d1.reshape(256, 4, 4).double().mean(dim=(0, 1)); t2 = d2.reshape(256, 4, 4).double().mean(dim=(0, 1))
sl1 = (M / 128) * (F / 64) * 3 / 256; ks2 = (M / 128) * (F / 16) / 256
print(f"fragment-major MLP (fc1 + fc2): {ms * 1e3:.0f} us = {2.0 * M * 384 * F * 2 * 3 / ms / 1e9:.0f} TFLOP/s issued; fc1 per slice period: wait {t1[0] / sl1:.0f}, steps {t1[2] / sl1:.0f} (floor 1536), "
      f"total {t1[3] / sl1:.0f}; fc2 per k-step: wait+barrier {t2[0] / ks2:.0f}, steps {t2[1] / ks2:.0f} (floor 1152), epilogue {t2[2] / ks2:.0f}, total {t2[3] / ks2:.0f}", flush=True)
