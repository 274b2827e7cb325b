#!/usr/bin/env python
"""How long does a SMALL kernel on a third stream wait behind the persistent backbone kernels?  (VERDICT r4 item 6 / missing #4.)

RCCL elides its reduction kernel in a one-rank process group (`bench.py --force-collectives` under rocprofv3: no ncclDevKernel, the
two all-reduces cost 18 us of c10d bookkeeping each), so the queueing risk of the N > 1 runs -- an all-reduce kernel of a few
workgroups dispatched while every CU holds a persistent backbone workgroup with the whole register file -- is probed directly:
while the fp16 backbone runs back to back on stream A, a 478 KB element-wise kernel (the size of the gradient all-reduce) is
issued on stream C every ~0.5 ms, bracketed by events; C is a NORMAL-priority stream (c10d's default for RCCL) or a HIGH-priority
one (what `distributed.init_from_env` now asks c10d for).  Prints the percentiles of (wait + run) per probe.
    python scripts/probe_queueing.py [seconds of backbone work per leg]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import vit as OV  # noqa: E402  (synthetic weights only)
from wild_visual_navigation_amd.backbone import VitBackbone  # noqa: E402

dev = torch.device("cuda:0")
sd = OV.make_vit_state_dict("vit_small", 8, 28, seed=0)
bb = VitBackbone(sd, 448, 8, 6, device=dev, precision="fp16", max_chunk=64)
img = torch.rand(64, 3, 448, 448, device=dev)
buf = torch.zeros(119491, device=dev)


def pct(xs):
    xs = sorted(xs)
    return {k: round(xs[min(len(xs) - 1, int(q * len(xs)))], 4) for k, q in (("p10", 0.1), ("median", 0.5), ("p90", 0.9), ("p99", 0.99), ("max", 1.0))}


def leg(priority, busy, n_iter=12):
    A = torch.cuda.Stream(device=dev)
    Cs = torch.cuda.Stream(device=dev, priority=priority)
    torch.cuda.synchronize()
    evs = []
    if busy:
        with torch.cuda.stream(A):
            for _ in range(n_iter):           # ~12 x 23 ms of persistent kernels, enqueued ahead
                bb.forward_tokens(img)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(Cs):
            a.record()
            buf.add_(1.0)
            b.record()
        evs.append((a, b))
        time.sleep(0.0005)
    torch.cuda.synchronize()
    return pct([a.elapsed_time(b) for a, b in evs]), len(evs)


for _ in range(2):
    bb.forward_tokens(img)
torch.cuda.synchronize()
print("| third stream | GPU | probes | p10 ms | median | p90 | p99 | max |\n|---|---|---|---|---|---|---|---|")
for name, prio in (("normal priority (c10d default)", 0), ("high priority", -1)):
    for busy in (False, True):
        p, n = leg(prio, busy)
        print(f"| {name} | {'persistent backbone kernels running' if busy else 'idle'} | {n} | {p['p10']} | {p['median']} | {p['p90']} | {p['p99']} | {p['max']} |", flush=True)
