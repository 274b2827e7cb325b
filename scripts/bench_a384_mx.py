#!/usr/bin/env python
"""The A-stationary kernel's MX instantiations (csrc/gemm_a384_x3.hip) on their own at the rows of 128 frames: LayerNorm-on-load fc1 + GELU -> MX planes and
LayerNorm-on-load q | k | v^T, with the in-kernel cycle counters of the instrumented builds."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from wild_visual_navigation_amd import _lib  # noqa: E402
from wild_visual_navigation_amd.backbone import pack_a384_mx  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ntok_s, npad, heads = 3152, 3200, 6
M = B * ntok_s
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, 384, generator=g) * 1.5).to(dev)
st = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)], -1).contiguous()
gam, bet = (1.0 + 0.1 * torch.randn(384, generator=g)).to(dev), (0.05 * torch.randn(384, generator=g)).to(dev)
Mp = (M + 31) // 32 * 32

def timed(call, label, N):
    call(0)
    for _ in range(3): call(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call(0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    dbg = torch.zeros(256 * 4 * 4, dtype=torch.int64, device=dev)
    call(dbg.data_ptr()); torch.cuda.synchronize()
    d = dbg.reshape(256, 4, 4).double().mean(dim=(0, 1))
    slices = (M / 128) * (N / 64) * 3 / 256
    print(f"{label}: {ms * 1e3:.0f} us = {2.0 * M * N * 384 / ms / 1e9:.0f} algorithmic TFLOP/s; per slice period ({slices:.0f} per wave): wait+barrier {d[0] / slices:.0f}, "
          f"DMA issue {d[1] / slices:.0f}, steps {d[2] / slices:.0f} (MFMA floor 1024), total {d[3] / slices:.0f} cycles", flush=True)

F = 1536
w1 = (torch.randn(F, 384, generator=g) * 0.05).to(dev); b1 = (torch.randn(F, generator=g) * 0.1).to(dev)
w1p = pack_a384_mx(w1)
hid = torch.zeros(Mp * F * 4, dtype=torch.uint8, device=dev)
n_h, n_8 = Mp * F * 2, Mp * F
timed(lambda dbg: _lib.check(lib.wvn_debug_mlp_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), w1p.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.data_ptr() + n_h,
                                                  hid.data_ptr() + n_h + n_8, 0, 0, 0, M, F, dbg, 0, _lib.stream()), "fc1 mx"), "fc1 MX (LayerNorm on load, GELU, MX planes out)", F)
wq = (torch.randn(1152, 384, generator=g) * 0.06).to(dev); bq = (torch.randn(1152, generator=g) * 0.02).to(dev)
wqp = pack_a384_mx(wq)
q = torch.zeros(2, B, heads, npad, 64, dtype=torch.float16, device=dev)
k = torch.zeros(B, heads, npad, 64, dtype=torch.float16, device=dev)
vt = torch.zeros(B, heads, 64, npad, dtype=torch.float16, device=dev)
for two in (1, 0):
    timed(lambda dbg: _lib.check(lib.wvn_debug_qkv_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), wqp.data_ptr(), bq.data_ptr(), q[0].data_ptr(), q[1].data_ptr() if two else 0,
                                                      k.data_ptr(), vt.data_ptr(), heads, npad, ntok_s, 0.18, M, dbg, _lib.stream()), "qkv mx"), f"q | k | v^T MX ({'two' if two else 'one'}-plane q)", 1152)
