#!/usr/bin/env python
"""Phase breakdown of the A-stationary GEMM kernel from s_memtime stamps of wave 0 of every workgroup
(diagnostic: the stamps cost a few percent).   python tools/astat_stamps.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
M, K = 32768, 320
for name, N, kw in (("qkv bf16", 960, {}), ("out res f32", 320, "res"), ("geglu", 2560, "geglu")):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    ns = N // 64
    ws = torch.zeros(M // 128, 2 + 3 * ns, dtype=torch.int64, device=dev)
    args = dict(debug_ws=ws)
    if kw == "res":
        stream = torch.zeros(M, N, device=dev)
        args.update(bias=bias, residual=stream, out_f32=True, out=stream)
    elif kw == "geglu":
        args.update(bias=bias, geglu=True)
    for _ in range(3):
        ops.gemm(a, w, **args)
    torch.cuda.synchronize()
    t = ws.cpu().numpy().astype(np.float64)
    t0 = t[:, 0].min()
    start, aload = t[:, 0] - t0, t[:, 1] - t[:, 0]
    top, mf, ep = t[:, 2::3], t[:, 3::3], t[:, 4::3]
    wait = np.concatenate([(top[:, :1] - t[:, 1:2]), top[:, 1:] - ep[:, :-1]], 1)
    print(f"{name}: N={N} strips={ns}; cycles (median over {M // 128} workgroups): start skew {np.median(start):.0f} (max {start.max():.0f}), "
          f"A load {np.median(aload):.0f}, per strip: wait+barrier {np.median(wait):.0f}, issue+MFMA {np.median(mf - top):.0f}, "
          f"epilogue {np.median(ep - mf):.0f}; total {np.median(ep[:, -1] - t[:, 0]):.0f} (max end {(ep[:, -1] - t0).max():.0f})")
    print("   per-strip wait (median over WGs):", " ".join(f"{x:.0f}" for x in np.median(wait, 0)[:16]))
