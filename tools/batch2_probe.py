#!/usr/bin/env python
"""Throughput with TWO clips per forward (UNet batch 4 under CFG) against the benchmark's one clip per forward.
python tools/batch2_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
model = bench.build_model(dev)
for b in (1, 2):
    sets = [bench.synthetic_inputs(model, dev, b=b, clip=i) for i in range(4)]
    for i in range(4):
        if i == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = bench.sample_clip(model, *sets[i], True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    assert torch.isfinite(out).all()
    print(f"clips per forward {b}: {dt * 1e3:.1f} ms per sampling call, {16 * b / dt:.2f} frames/s", flush=True)
