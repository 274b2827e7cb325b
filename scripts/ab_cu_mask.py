#!/usr/bin/env python3
"""Same-process A/B of the two-stream step schedule: shared CUs with stream priorities (round 3) against CU-partitioned streams
(hipExtStreamCreateWithCUMask, VERDICT r3 "Next round" item 4).  One pipeline, one set of inputs; every configuration runs
`--warmup` + `--steps` steps of the headline workload.  Prints one line per configuration and a table at the end.

    python scripts/ab_cu_mask.py [--steps 12] [--configs none 26,26 24,24 28,20 ...] [--precision fp16]

A configuration "a_hi,b_lo[,layout]": backbone stream on CUs [0, a_hi) of every XCD, tail stream on CUs [b_lo, 32)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--configs", nargs="*", default=["none", "32,8", "32,16", "32,24", "32,28", "26,26", "26,26,lin", "16,16"])
    a = ap.parse_args()
    import torch

    sys.argv = [sys.argv[0], "--precision", a.precision]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B = args.batch
    gen = torch.Generator().manual_seed(1000)
    pool = [torch.rand(B, 3, args.size, args.size, generator=gen).to(dev) for _ in range(max(1, args.pool))]
    labels = [torch.rand(B, 20, 2, generator=gen).to(dev) for _ in range(len(pool))]
    fe, model, trainer = bench.make_pipeline(args, dev, a.precision, "upstream")
    rows = []
    for cfg in a.configs + a.configs[:1]:   # the first configuration again at the end: drift of the box
        pipe = bench.TwoStreamPipeline(fe, trainer, args, dev, streams=cfg)
        for i in range(a.warmup):
            pipe.step(pool[i % len(pool)], labels[i % len(pool)], next_img=None if i == a.warmup - 1 else pool[(i + 1) % len(pool)])
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            pipe.step(pool[i % len(pool)], labels[i % len(pool)], next_img=None if i == a.steps - 1 else pool[(i + 1) % len(pool)])
        pipe.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fps = B * a.steps / dt
        rows.append((cfg, fps, dt / a.steps * 1e3))
        print(f"cu split {cfg:12s}: {fps:8.1f} frames/s  {dt / a.steps * 1e3:7.2f} ms/step", flush=True)
        # (the masked streams are left to the process exit: destroying a stream torch still holds events of aborts the process)
    print("| backbone CUs / XCD | tail CUs / XCD | layout | frames/s | ms/step |\n|---|---|---|---|---|")
    for cfg, fps, ms in rows:
        if cfg == "none":
            print(f"| all (shared, tail stream at high priority) | all | - | {fps:.1f} | {ms:.2f} |")
        else:
            p = cfg.split(",")
            print(f"| 0..{int(p[0]) - 1} | {p[1]}..31 | {p[2] if len(p) > 2 else 'rr'} | {fps:.1f} | {ms:.2f} |")


if __name__ == "__main__":
    main()
