#!/usr/bin/env python
"""3x3 / temporal convolutions of the UNet as implicit GEMMs, hipGraph-timed (activations rotate over enough copies to be cold in L2,
as inside the model).  A/B aid for the operand walk (CCV_GEMM_TAPINNER=0|1); run under `rocprofv3 --pmc FETCH_SIZE` for the bytes.
    python tools/conv_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
CASES = [("conv", 32, 320, 320), ("conv", 32, 640, 320), ("conv", 32, 960, 320), ("conv", 16, 640, 640), ("conv", 16, 1280, 640),
         ("conv", 8, 1280, 1280), ("tconv", 32, 320, 320), ("tconv", 16, 640, 640), ("tconv", 8, 1280, 1280)]
REPS = 16
for kind, h, cin, cout in CASES:
    M = 32 * h * h
    if kind == "conv":
        w = pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
        kw = dict(k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(h, h, h, h, 1, 0))
    else:
        w = pack.pack_tconv3(torch.randn(cout, cin, 3, 1, 1, device=dev) * 0.02)
        kw = dict(k=cin, taps=3, gather=ops.GATHER_TCONV3, tconv=(16, h * h))
    acts = [torch.randn(M, cin, device=dev).to(torch.bfloat16) for _ in range(8)]
    out = torch.empty(M, cout, device=dev, dtype=torch.bfloat16)
    plan = ops.gemm_plan(acts[0], w, **kw) if hasattr(ops, "gemm_plan") else None
    for i in range(8):
        ops.gemm(acts[i], w, out=out, **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            ops.gemm(acts[i % 8], w, out=out, **kw)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    taps = 9 if kind == "conv" else 3
    print(f"{kind:5s} h={h:2d} {cin:5d}->{cout:5d}: {best:7.1f} us  {2.0 * M * cout * cin * taps / best / 1e6:6.0f} TF/s  TAPINNER={os.environ.get('CCV_GEMM_TAPINNER', '1')}", flush=True)
