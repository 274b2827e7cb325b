"""Micro-benchmark + in-kernel phase timing of the fused LayerNorm + QKV kernel at the shipped shape (64 frames of 3152 rows)."""
import ctypes
import torch
from wild_visual_navigation_amd import _lib, ops

dev = torch.device("cuda:0")
frames, ntok_s, npad = 64, 3152, 3200
M = frames * ntok_s
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 384, generator=g).to(dev)
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
w = (torch.randn(1152, 384, generator=g) * 0.05).to(torch.bfloat16).to(dev)
bias = torch.randn(1152, generator=g).to(dev)
n = frames * 6 * npad * 64
buf = torch.zeros(3 * n, dtype=torch.bfloat16, device=dev)
q, k, vt = buf[:n], buf[n:2 * n], buf[2 * n:]
L = _lib.lib()


def run():
    _lib.check(L.wvn_qkv_fused(x.data_ptr(), 384, gam.data_ptr(), bet.data_ptr(), 1e-6, w.data_ptr(), bias.data_ptr(), q.data_ptr(), k.data_ptr(),
                               vt.data_ptr(), 6, npad, ntok_s, 0.18, M, torch.cuda.current_stream().cuda_stream))


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record()
torch.cuda.synchronize()
t = a.elapsed_time(b) / 20 * 1e3
print("qkv_fused: %.1f us (%.0f TFLOP/s)" % (t, 2.0 * M * 384 * 1152 / t / 1e6))
dbg = torch.zeros(256 * 4 * 4, dtype=torch.int64, device=dev)
L.wvn_debug_qkv_fused_timing(ctypes.c_void_p(dbg.data_ptr()))
a.record(); run(); b.record(); torch.cuda.synchronize()
L.wvn_debug_qkv_fused_timing(ctypes.c_void_p(0))
d = dbg.cpu().view(256, 4, 4).double()
tot = d[..., 3]
print("instrumented launch %.1f us; per wave cycles: total mean %.0f max %.0f | LayerNorm %.0f (%.1f%%) | slices %.0f (%.1f%%) | epilogues %.0f (%.1f%%)" % (
    a.elapsed_time(b) * 1e3, tot.mean(), tot.max(), d[..., 0].mean(), 100 * d[..., 0].mean() / tot.mean(), d[..., 1].mean(), 100 * d[..., 1].mean() / tot.mean(),
    d[..., 2].mean(), 100 * d[..., 2].mean() / tot.mean()))
print("MFMA floor per wave: %.0f cycles" % (3.08 * 18 * 96 * 32))

# the PRE form: rows already normalised, as operand fragments (what the resident projection + MLP kernel leaves)
frag = (torch.randn((M + 31) // 32 * 24 * 512, generator=g) * 1.0).to(torch.bfloat16).to(dev)
wperm = w[:, ops.vt_token_order(384, device=dev)].contiguous()


def run_pre():
    _lib.check(L.wvn_qkv_prenorm(frag.data_ptr(), wperm.data_ptr(), bias.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), 6, npad, ntok_s,
                                 0.18, M, torch.cuda.current_stream().cuda_stream))


for _ in range(3):
    run_pre()
torch.cuda.synchronize()
a.record()
for _ in range(20):
    run_pre()
b.record()
torch.cuda.synchronize()
t = a.elapsed_time(b) / 20 * 1e3
print("qkv_prenorm: %.1f us (%.0f TFLOP/s)" % (t, 2.0 * M * 384 * 1152 / t / 1e6))
dbg.zero_()
L.wvn_debug_qkv_fused_timing(ctypes.c_void_p(dbg.data_ptr()))
a.record(); run_pre(); b.record(); torch.cuda.synchronize()
L.wvn_debug_qkv_fused_timing(ctypes.c_void_p(0))
d = dbg.cpu().view(256, 4, 4).double()
tot = d[..., 3]
print("instrumented launch %.1f us; per wave cycles: total mean %.0f max %.0f | fragment loads %.0f (%.1f%%) | slices %.0f (%.1f%%) | epilogues %.0f (%.1f%%)" % (
    a.elapsed_time(b) * 1e3, tot.mean(), tot.max(), d[..., 0].mean(), 100 * d[..., 0].mean() / tot.mean(), d[..., 1].mean(), 100 * d[..., 1].mean() / tot.mean(),
    d[..., 2].mean(), 100 * d[..., 2].mean() / tot.mean()))
