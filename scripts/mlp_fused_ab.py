"""One line per library build: the fused block MLP (LayerNorm inside; with and without the projection phase) at the shipped shape.
    for l in ...; do WVN_LIB_PATH=$PWD/wild_visual_navigation_amd/lib/$l python scripts/mlp_fused_ab.py $l; done"""
import sys
import torch
from wild_visual_navigation_amd import _lib, ops

dev = torch.device("cuda:0")
M, F = 64 * 3136, 1536
dt = torch.float16 if "f16" in sys.argv[2:] else torch.bfloat16
g = torch.Generator().manual_seed(0)
w1 = (torch.randn(F, 384, generator=g) * 0.06).to(dt).to(dev)
w2 = (torch.randn(384, F, generator=g) * 0.03).to(dt).to(dev)
wp = (torch.randn(384, 384, generator=g) * 0.05).to(dt).to(dev)
attn = torch.randn(M, 384, generator=g).to(dt).to(dev)
b1, b2, bp = torch.randn(F, generator=g).to(dev), torch.randn(384, generator=g).to(dev), torch.randn(384, generator=g).to(dev)
x = torch.randn(M, 384, generator=g).to(dev)
w2p = (w2 * 0)[:, ops.vt_token_order(F, device=dev)].contiguous()
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)


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


t1 = timeit(lambda: ops.mlp_fused(None, w1, b1, w2p, b2 * 0, x, ln=(gam, bet, 1e-6)))
t2 = timeit(lambda: ops.proj_mlp_fused(attn, wp * 0, bp * 0, (gam, bet, 1e-6), w1, b1, w2p, b2 * 0, x))
t3 = float("nan")
if hasattr(_lib.lib(), "wvn_proj_mlp_resident"):
    w1p = w1[:, ops.vt_token_order(384, device=dev)].contiguous()
    t3 = timeit(lambda: ops.proj_mlp_resident(attn, wp * 0, bp * 0, (gam, bet, 1e-6), w1p, b1, w2p, b2 * 0, x))
fl = 2.0 * M * 384 * F * 2
print("%-22s mlp %.1f us (%.0f TFLOP/s) | proj+mlp %.1f us (%.0f TFLOP/s) | resident %.1f us (%.0f TFLOP/s)"
      % (sys.argv[1], t1, fl / t1 / 1e6, t2, (fl + 2.0 * M * 384 * 384) / t2 / 1e6, t3, (fl + 2.0 * M * 384 * 384) / t3 / 1e6), flush=True)
