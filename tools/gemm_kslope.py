#!/usr/bin/env python
"""Time per K-slab of the GEMM kernels: fixed M x N, growing K, forced kernel configuration.
    python tools/gemm_kslope.py
Slope = time per 64 of K (per workgroup round), intercept = launch + ramp + epilogue."""
import os
import sys

os.environ["CCV_GEMM_TUNE"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
ops.TRACK_GEMM_PLAN = True


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M, N in ((2048, 1280), (128, 1280), (8192, 640), (32768, 320)):
    for ring in (-1, 2, 3, 4):
        os.environ["CCV_GEMM_RING"] = str(ring)
        os.environ["CCV_GEMM_SPLIT"] = "1"
        row = []
        for K in (64, 320, 640, 1280, 2560, 5120):
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            us = timeit(lambda: ops.gemm(a, w))
            row.append(us)
        plan = ops.LAST_GEMM_PLAN
        slope = (row[-1] - row[-2]) / (2560 / 64)
        print(f"M={M:6d} N={N:5d} plan={plan}: " + " ".join(f"K={k}:{t:6.1f}" for k, t in zip((64, 320, 640, 1280, 2560, 5120), row)) + f"  | {slope * 1000:.0f} ns per 64-slab", flush=True)
