#!/usr/bin/env python
"""One-launch transformer sub-blocks (csrc/ccv_fused.hip) against the launches they replace, per call, on cold operands: every call
of a hipGraph of REPS calls works on another copy of the fp16 stream (the copies together exceed the 256 MB Infinity Cache, as a
layer's input does inside the model: written by the layer before, long evicted from L2), weights stay the same (hot in L2 for both).
    python tools/fused_probe.py [ff] [tchain]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
REPS = 32


def timed_graph(fn_of_i, reps=REPS):
    for i in range(reps):
        fn_of_i(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn_of_i(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def probe_ff(M=32768, C=320):
    ncopies = min(REPS, (400 << 20) // (M * C * 2) + 1)
    xs = [(torch.randn(M, C, device=dev) * 1.5).to(torch.float16) for _ in range(ncopies)]
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1 = torch.randn(8 * C, C, device=dev) * 0.05
    b1 = torch.randn(8 * C, device=dev) * 0.1
    w2 = torch.randn(C, 4 * C, device=dev) * 0.03
    b2 = torch.randn(C, device=dev) * 0.1
    w1p, b1p = pack.interleave_geglu(w1, b1)
    w2l, w2p = pack.pack_linear(w2), pack.permute_k16_for_acc_operand(w2)

    def three(i):
        x = xs[i % ncopies]
        n = ops.layernorm(x, gamma, beta, eps=1e-5)
        hid = ops.gemm(n, w1p, bias=b1p, geglu=True)
        ops.gemm(hid, w2l, bias=b2, residual=x, out_dtype=torch.float16, out=x)

    def two(i):      # LayerNorm in the up-projection's prologue (what one clip at a time runs today)
        x = xs[i % ncopies]
        hid = ops.gemm(ops.LazyLN(x, gamma, beta, 1e-5), w1p, bias=b1p, geglu=True)
        ops.gemm(hid, w2l, bias=b2, residual=x, out_dtype=torch.float16, out=x)

    def fused(i):
        x = xs[i % ncopies]
        ops.ff_fused(x, gamma, beta, 1e-5, w1p, b1p, w2p, b2, out=x)

    fl = 2.0 * M * C * 12 * C
    t3, t1 = timed_graph(three), timed_graph(fused)
    old = ops._FUSE_LN_MODE
    ops._FUSE_LN_MODE = "1"
    t2 = timed_graph(two)
    ops._FUSE_LN_MODE = old
    print(f"feed-forward M={M} C={C} ({ncopies} stream copies): LayerNorm + 2 GEMMs {t3:6.1f} us | LN-prologue GEMM + GEMM {t2:6.1f} us | "
          f"ONE launch {t1:6.1f} us ({fl / t1 / 1e6:5.0f} TFLOP/s, {100 * fl / t1 / 1e6 / 2500:.0f} % of the bf16 peak)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ff"]
    if "ff" in which:
        for M in (32768, 16384):
            probe_ff(M)
