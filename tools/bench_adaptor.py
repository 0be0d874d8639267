#!/usr/bin/env python
"""MultiLatentEpipolarAdaptor at the shipped size (16 x 1024 latents of width 512, depth 12, conditioning frame + 2
context frames, b = 1) on the HIP kernels: device time with HIP events, algorithmic FLOPs, and the oracle timed on the
host cores on ONE layer as the CPU baseline.
    python tools/bench_adaptor.py [--iters 5] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd.adaptor import MultiLatentEpipolarAdaptor  # noqa: E402
from oracle import adaptor_oracle as ao  # noqa: E402  (cpu_baseline leg only)
from oracle.unet_oracle import seeded_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--context-frames", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    cfg = dict(ao.FULL_CFG)
    man = {k: list(v.shape) for k, v in MultiLatentEpipolarAdaptor(**cfg).state_dict().items()}
    sd = seeded_state_dict(man, 5, std=0.05)
    m = MultiLatentEpipolarAdaptor(**cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(6)
    N, Lq, C = args.context_frames, 16 * 1024, 512
    Lk = N * 1024
    x = torch.randn(1, Lk, 4, generator=g)
    mask = torch.rand(1, Lq, Lk, generator=g) < 0.05
    xd, md = x.to(dev), mask.to(dev)
    y = m(xd, md)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = m(xd, md)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    per_layer = 2 * Lq * C * 512 + 2 * (Lk + 2) * C * 1024 + 4 * Lq * (Lk + 2) * 512 + 2 * Lq * 512 * C + 2 * 2 * Lq * C * 4 * C
    fl = 12 * per_layer + 2 * Lk * 4 * C + 2 * Lq * C * 4
    line = {"what": f"MultiLatentEpipolarAdaptor, b=1, {N} context frames, 16x1024 latents, depth 12 (SURVEY.md 8 f1)", "ms": ms,
            "algorithmic_tflop": fl / 1e12, "achieved_tflops": fl / ms / 1e9, "frac_of_bf16_peak": fl / ms / 1e9 / 2500.0,
            "note": "mask packing (once per clip) included; attention counted dense"}
    if not args.no_cpu:
        cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        torch.set_num_threads(cores)
        one = dict(cfg, depth=1)
        t0 = time.perf_counter()
        ao.adaptor_forward(sd, one, x, mask)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"kind": "port", "cores": cores, "seconds_per_layer": dt, "seconds_extrapolated": 12 * dt,
                                "sample": "1 of 12 layers through the fp32 oracle"}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
