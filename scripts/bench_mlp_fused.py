"""Micro-benchmark of the fused block MLP against the un-fused pair at the shipped shape (64 frames of 3152 rows)."""
import sys
import torch
from wild_visual_navigation_amd import _lib, ops

dev = torch.device("cuda:0")
M, F = 64 * 3152, 1536
g = torch.Generator().manual_seed(0)
xn = torch.randn(M, 384, generator=g).to(torch.bfloat16).to(dev)
w1 = (torch.randn(F, 384, generator=g) * 0.06).to(torch.bfloat16).to(dev)
w2 = (torch.randn(384, F, generator=g) * 0.03).to(torch.bfloat16).to(dev)
b1 = torch.randn(F, generator=g).to(dev)
b2 = torch.randn(384, generator=g).to(dev)
x = torch.randn(M, 384, generator=g).to(dev)
hid = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
w2p = w2[:, ops.vt_token_order(F, device=dev)].contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def unfused():
    ops.gemm_bf16(xn, w1, b1, _lib.EPI_GELU_BF16, out=hid)
    ops.gemm_bf16(hid, w2, b2, _lib.EPI_RESID_F32, out=x)


print("unfused pair: %.1f us" % timeit(unfused))
fl = 2.0 * M * 384 * F * 2
t = timeit(lambda: ops.mlp_fused(xn, w1, b1, w2p, b2, x))
print("fused (xn given): %.1f us  (%.0f TFLOP/s)" % (t, fl / t / 1e6))
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
x.normal_()
t = timeit(lambda: ops.mlp_fused(None, w1, b1, w2p * 0, b2 * 0, x, ln=(gam, bet, 1e-6)))
print("fused (LayerNorm inside): %.1f us  (%.0f TFLOP/s)" % (t, fl / t / 1e6))
import ctypes
attn = torch.randn(M, 384, generator=g).to(torch.bfloat16).to(dev)
wp = (torch.randn(384, 384, generator=g) * 0.05).to(torch.bfloat16).to(dev)
bp = torch.randn(384, generator=g).to(dev)
w1p = w1[:, ops.vt_token_order(384, device=dev)].contiguous()
t = timeit(lambda: ops.proj_mlp_resident(attn, wp * 0, bp * 0, (gam, bet, 1e-6), w1p, b1, w2p * 0, b2 * 0, x))
print("projection + MLP, resident rows: %.1f us  (%.0f TFLOP/s)" % (t, (fl + 2.0 * M * 384 * 384) / t / 1e6))
names = ["prologue (rows, projection, LayerNorm)", "fc1 slices", "GELU + pack", "fc2 slices", "epilogue", "projection slices"]
for label, fn in (("LayerNorm + MLP", lambda: ops.mlp_fused(None, w1, b1, w2p * 0, b2 * 0, x, ln=(gam, bet, 1e-6))),
                  ("resident", lambda: ops.proj_mlp_resident(attn, wp * 0, bp * 0, (gam, bet, 1e-6), w1p, b1, w2p * 0, b2 * 0, x))):
    dbg = torch.zeros(256 * 4 * 12, dtype=torch.int64, device=dev)
    _lib.lib().wvn_debug_mlp_fused_timing(ctypes.c_void_p(dbg.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    _lib.lib().wvn_debug_mlp_fused_timing(ctypes.c_void_p(0))
    d = dbg.cpu().view(256, 4, 12).double()
    tot = d[..., 7].mean()
    print("%s: instrumented launch %.1f us; per wave cycles: total %.0f (max %.0f); " % (label, a.elapsed_time(b) * 1e3, tot, d[..., 7].max()) +
          " | ".join("%s %.0f (%.1f%%)" % (n, d[..., i].mean(), 100 * d[..., i].mean() / tot) for i, n in enumerate(names)) +
          " || inside the slices: " + " | ".join("%s %.0f (%.1f%%)" % (n, d[..., i].mean(), 100 * d[..., i].mean() / tot) for i, n in ((8, "wait for the next slice"), (9, "barrier"), (10, "DMA issue"))))
print("MFMA floor per wave: %.0f cycles (+ %.0f projection)" % (M / 128 / 256 * 24 * 96 * 32, M / 128 / 256 * 288 * 32))
