#!/usr/bin/env python
"""A/B of the two forms of the A-stationary MX kernel (csrc/gemm_a384_x3.hip: one wave per SIMD against two workgroups per CU) at the rows of 128 frames:
time per launch (HIP events) and the difference between their outputs."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
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
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{label}: {ms * 1e3:.0f} us = {2.0 * M * N * 384 / ms / 1e9:.0f} algorithmic TFLOP/s", flush=True)


F = 1536
w1 = (torch.randn(F, 384, generator=g) * 0.05).to(dev); b1 = (torch.randn(F, generator=g) * 0.1).to(dev)
w1p = pack_a384_mx(w1)
n_h, n_8 = Mp * F * 2, Mp * F
outs = []
for form in (1, 2, 1, 2):
    lib.wvn_debug_n384_pair(32 + form)
    hid = torch.zeros(n_h + n_8, dtype=torch.uint8, device=dev)
    timed(lambda: _lib.check(lib.wvn_debug_mlp_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), w1p.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.data_ptr() + n_h,
                                                 0, 0, 0, 0, M, F, 0, 0, _lib.stream()), "fc1 mx"), f"fc1 MX form {form}", F)
    outs.append(hid)
h1, h2 = outs[0][:n_h].view(torch.float16), outs[1][:n_h].view(torch.float16)
print("fc1: fp16 plane max |form 1 - form 2| =", (h1.float() - h2.float()).abs().max().item(), " max |h| =", h1.float().abs().max().item(),
      " l8 bytes differing:", (outs[0][n_h:] != outs[1][n_h:]).float().mean().item(), flush=True)
del outs, hid
wq = (torch.randn(1152, 384, generator=g) * 0.06).to(dev); bq = (torch.randn(1152, generator=g) * 0.02).to(dev)
wqp = pack_a384_mx(wq)
per = B * heads * npad * 64
res = []
for form in (1, 2, 1, 2):
    lib.wvn_debug_n384_pair(32 + form)
    buf = torch.zeros(4 * per, dtype=torch.float16, device=dev)
    q0, q1, k, vt = (buf[i * per:(i + 1) * per] for i in range(4))
    for two in (1, 0):
        timed(lambda: _lib.check(lib.wvn_debug_qkv_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), wqp.data_ptr(), bq.data_ptr(), q0.data_ptr(), q1.data_ptr() if two else 0,
                                                     k.data_ptr(), vt.data_ptr(), heads, npad, ntok_s, 0.18, M, 0, _lib.stream()), "qkv mx"), f"q | k | v^T MX form {form} ({'two' if two else 'one'}-plane q)", 1152)
    res.append(buf)
d = (res[0].float() - res[1].float()).abs()
print("qkv: max |form 1 - form 2| =", d.max().item(), " max |value| =", res[0].float().abs().max().item(), flush=True)
lib.wvn_debug_n384_pair(32)

# in-kernel counters of form 2 (two workgroups per CU: 512 x 4 waves)
lib.wvn_debug_n384_pair(34)
hid = torch.zeros(n_h + n_8, dtype=torch.uint8, device=dev)
dbg = torch.zeros(512 * 4 * 4, dtype=torch.int64, device=dev)
_lib.check(lib.wvn_debug_mlp_mx(x.data_ptr(), 384, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), w1p.data_ptr(), b1.data_ptr(), hid.data_ptr(), hid.data_ptr() + n_h,
                                0, 0, 0, 0, M, F, dbg.data_ptr(), 0, _lib.stream()), "fc1 mx")
torch.cuda.synchronize()
d = dbg.reshape(512, 4, 4).double()
tiles = (M / 128) * (F / 64) / 512
m = d.mean(dim=(0, 1))
print(f"form 2 fc1 counters per wave and TILE ({tiles:.1f} tiles per wave): wait + barrier {m[0] / tiles:.0f}, prologues {m[1] / tiles:.0f}, regions {m[2] / tiles:.0f} (MFMA 3072), "
      f"epilogue + rest {(m[3] - m[0] - m[1] - m[2]) / tiles:.0f}, total {m[3] / tiles:.0f}; total min / max over waves {d[..., 3].min().item():.0f} / {d[..., 3].max().item():.0f}", flush=True)
lib.wvn_debug_n384_pair(32)
