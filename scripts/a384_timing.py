#!/usr/bin/env python
"""Phase timing of the A-stationary K=384 GEMM (s_memtime inside the kernel, TIMING build): per wave the
shader cycles spent in {wait + barrier, MFMA block, epilogue part} summed over all slice periods, and the
wave's total.  Prints means per ping-pong role (waves 0-3: MFMA first; waves 4-7: epilogue first)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wild_visual_navigation_amd import _lib  # noqa: E402
from wild_visual_navigation_amd._lib import check, ptr, stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    h = _lib.lib()
    fn = h.wvn_debug_gemm_bf16_timed
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_int, C.c_void_p, C.c_void_p]
    M = args.frames * 3144
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, 384, generator=g).to(torch.bfloat16).to(dev)
    res = {}
    for name, N, epi in (("bf16_1152", 1152, _lib.EPI_BF16), ("gelu_1536", 1536, _lib.EPI_GELU_BF16),
                         ("resid_384", 384, _lib.EPI_RESID_F32)):
        w = (torch.randn(N, 384, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.zeros(M, N, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.bfloat16, device=dev)
        units = ((M + 255) // 256) * (N // 64)
        nwg = min(units, 256)  # persistent launch: one workgroup per CU
        dbg = torch.zeros(nwg * 8 * 4, dtype=torch.int64, device=dev)
        for _ in range(3):
            check(fn(ptr(a), 384, ptr(w), 384, ptr(bias), ptr(out), N, M, N, 384, epi, ptr(dbg), stream()))
        torch.cuda.synchronize()
        d = dbg.cpu().reshape(nwg, 8, 4).double()
        periods = units * 3 / nwg  # slice periods per workgroup (balanced to +-3)
        for role, sl in (("mfma_first", slice(0, 4)), ("epi_first", slice(4, 8))):
            m = d[:, sl].mean(dim=(0, 1))
            res[f"{name}/{role}"] = {"wait": round(m[0].item() / periods), "mfma": round(m[1].item() / periods),
                                     "epi": round(m[2].item() / periods), "total_per_period": round(m[3].item() / periods),
                                     "total": round(m[3].item())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
