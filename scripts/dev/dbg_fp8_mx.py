import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from wild_visual_navigation_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.lib()
g = lambda s: torch.Generator().manual_seed(s)
def run(M, N, K, lo, hi):
    a8 = (torch.randn(M, K, generator=g(31)) * 40).clamp(-448, 448).to(torch.float8_e4m3fn).to(dev)
    a_sc = torch.randint(lo, hi, (M, K // 32), generator=g(32), dtype=torch.uint8).to(dev)
    w = torch.randn(N, K, generator=g(33)) * 0.05
    wq, sw = ops.quantize_rows_fp8(w.to(dev))
    bias = (torch.randn(N, generator=g(34)) * 0.1).to(dev)
    av = a8.double() * torch.pow(2.0, a_sc.double() - 127.0).repeat_interleave(32, dim=1)
    wv = wq.double() * sw.double()[:, None]
    ref = av @ wv.T + bias.double()
    mag = av.abs() @ wv.abs().T + 1.0
    out = torch.zeros(M, N, device=dev)
    _lib.check(lib.wvn_gemm_fp8_mx(a8.data_ptr(), K, a_sc.data_ptr(), wq.data_ptr(), K, sw.data_ptr(), bias.data_ptr(), 0, out.data_ptr(), N, M, N, K, _lib.EPI_F32, _lib.stream()), "mx")
    e = (out.double() - ref).abs() / mag
    i = int(e.argmax()); r, c = i // N, i % N
    print(M, N, K, (lo, hi), "max rel", e.max().item(), "at", (r, c), "abs", (out.double() - ref)[r, c].item(), "ref", ref[r, c].item(), "mag", mag[r, c].item(), flush=True)
run(130, 256, 128, 118, 132)
run(130, 256, 128, 127, 128)
run(130, 256, 256, 118, 132)
run(128, 128, 128, 118, 132)
run(128, 128, 128, 120, 128)
run(515, 768, 3072, 118, 132)
