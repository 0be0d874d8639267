#!/usr/bin/env python
"""GroupNorm statistics from the producing convolution's epilogue (ccv_gemm gn_partial -> ccv_groupnorm_apply_parts) against the
separate statistics pass: per-launch times of conv, conv + statistics, norm, norm on handed-over statistics (hipGraph of 20 reps).
    python tools/gn_epilogue_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
REPS = 20


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


CASES = [("conv", 32, 32, 320, 320), ("conv", 32, 32, 960, 320), ("conv", 32, 16, 320, 640), ("tconv", 2, 32, 320, 320), ("tconv", 2, 16, 640, 640),
         ("tconv", 2, 8, 1280, 1280), ("tconv", 2, 4, 1280, 1280)]
for kind, inst, side, cin, cout in CASES:
    if kind == "conv":
        rows, rpi = inst * side * side, side * side
        w = pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
        kw = dict(k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(side, side, side, side, 1, 0))
    else:
        rows, rpi = inst * 16 * side * side, 16 * side * side
        w = pack.pack_tconv3(torch.randn(cout, cin, 3, 1, 1, device=dev) * 0.02)
        kw = dict(k=cin, taps=3, gather=ops.GATHER_TCONV3, tconv=(16, side * side))
    x = torch.randn(rows, cin, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    ops.TRACK_GEMM_PLAN = True
    h, st = ops.gemm(x, w, gn_rows=rpi, **kw)
    plan = ops.LAST_GEMM_PLAN
    ops.TRACK_GEMM_PLAN = False
    if st is None:
        print(f"{kind} inst={inst} side={side} {cin}->{cout}: plan {plan}, no epilogue statistics")
        continue
    t_conv = timed(lambda: ops.gemm(x, w, **kw))
    t_conv_s = timed(lambda: ops.gemm(x, w, gn_rows=rpi, **kw))
    t_gn = timed(lambda: ops.groupnorm(h, gamma, beta, instances=inst, eps=1e-5, silu=True))
    t_gn_s = timed(lambda: ops.groupnorm(h, gamma, beta, instances=inst, eps=1e-5, silu=True, stats=st))
    print(f"{kind} inst={inst} side={side} {cin}->{cout}: plan {plan} slots {st[0].shape[1]}: conv {t_conv:6.1f} -> {t_conv_s:6.1f} us, "
          f"norm {t_gn:6.1f} -> {t_gn_s:6.1f} us, sum {t_conv + t_gn:6.1f} -> {t_conv_s + t_gn_s:6.1f} us", flush=True)
