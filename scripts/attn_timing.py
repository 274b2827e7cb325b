#!/usr/bin/env python
"""Phase timing of the bf16 attention kernel (s_memtime inside the kernel, TIMING build): per wave shader
cycles in {wait + barrier + DMA issue, QK^T, softmax, PV} summed over the tile loop; printed per tile."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wild_visual_navigation_amd import _lib  # noqa: E402
from wild_visual_navigation_amd._lib import check, ptr, stream  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    h = _lib.lib()
    B, heads, ntok, npad = 16, 6, 3137, 3200
    g = torch.Generator().manual_seed(0)
    q, k = (torch.randn(B, heads, npad, 64, generator=g).to(torch.bfloat16).to(dev) for _ in range(2))
    vt = torch.randn(B, heads, 64, npad, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty(B * ntok, heads * 64, dtype=torch.bfloat16, device=dev)
    nwg = B * heads * ((ntok + 127) // 128)
    dbg = torch.zeros(nwg * 4 * 5, dtype=torch.int64, device=dev)
    h.wvn_debug_attention_timing.argtypes = [C.c_void_p]
    h.wvn_debug_attention_timing(ptr(dbg))
    for _ in range(2):
        check(h.wvn_attention_bf16(ptr(q), ptr(k), ptr(vt), ptr(out), B, heads, ntok, npad, 0.125, stream()))
    torch.cuda.synchronize()
    h.wvn_debug_attention_timing(None)
    d = dbg.cpu().reshape(nwg, 4, 5).double().mean(dim=(0, 1))
    nt = (ntok + 63) // 64
    print(json.dumps({"per_tile": {"wait": round(d[0].item() / nt), "qk": round(d[1].item() / nt), "softmax": round(d[2].item() / nt),
                                   "pv": round(d[3].item() / nt), "total": round(d[4].item() / nt)}, "total": round(d[4].item())}))


if __name__ == "__main__":
    main()
