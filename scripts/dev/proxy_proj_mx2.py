#!/usr/bin/env python
"""Timing proxy: the two-workgroups-per-CU A-stationary MX kernel at N = 384 (q | k | v^T with two heads) -- what a projection on that kernel would cost -- against the
row-panel kernel's projection (scripts/bench_n384_mx.py: 0.41 ms)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from wild_visual_navigation_amd import _lib
from wild_visual_navigation_amd.backbone import pack_a384_mx
dev = torch.device("cuda:0")
B, ntok_s, npad = 128, 3152, 3200
M = B * ntok_s
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, 384, generator=g) * 1.5).to(dev)
st = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)], -1).contiguous()
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
for heads in (2, 6):
    N = 3 * heads * 64
    wp = pack_a384_mx((torch.randn(N, 384, generator=g) * 0.06).to(dev)); bq = torch.zeros(N, device=dev)
    per = B * heads * npad * 64
    buf = torch.zeros(4 * per, dtype=torch.float16, device=dev)
    call = lambda: _lib.check(lib.wvn_debug_qkv_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), wp.data_ptr(), bq.data_ptr(), buf.data_ptr(), 0,
                                                   buf.data_ptr() + 4 * per, buf.data_ptr() + 6 * per, heads, npad, ntok_s, 0.18, M, 0, _lib.stream()), "qkv")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    print(f"N = {N}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us", flush=True)
