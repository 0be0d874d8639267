#!/usr/bin/env python
"""Why do the GEMMs run 25-45 % slower inside the UNet than in tools/gemm_tune.py?  Times a few layer shapes (hipGraph of
`reps` launches each) with
    warm : the same weight / activation / output buffers on every launch (what a micro-benchmark does)
    coldW: a different weight copy per launch, the set of copies larger than the 256 MB Infinity Cache
    coldA: a different activation copy per launch
    cold : both (what a layer sees inside the model: weights last touched one step ago, activations just written)
    python tools/cold_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
SHAPES = [  # M, N, K, residual fp32 out
    (32768, 320, 1280, False),
    (8192, 1920, 640, False),
    (8192, 640, 640, True),
    (2048, 3840, 1280, False),
    (2048, 1280, 1280, True),
    (2048, 1280, 5120, False),
    (512, 1280, 1280, True),
]
REPS = 72


def timed_graph(fn_of_i):
    for i in range(REPS):
        fn_of_i(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            fn_of_i(i)
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


for M, N, K, res in SHAPES:
    per_set = (N * K + M * K) * 2
    ncopies = max(2, min(REPS, (600 << 20) // per_set + 1))
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(ncopies)]
    As = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(ncopies)]
    stream = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else torch.bfloat16)

    def run(iw, ia):
        if res:
            ops.gemm(As[ia], Ws[iw], residual=stream, out_f32=True, out=out)
        else:
            ops.gemm(As[ia], Ws[iw], out=out)

    t_warm = timed_graph(lambda i: run(0, 0))
    t_cw = timed_graph(lambda i: run(i % ncopies, 0))
    t_ca = timed_graph(lambda i: run(0, i % ncopies))
    t_c = timed_graph(lambda i: run(i % ncopies, i % ncopies))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d} {'res32' if res else 'plain'} copies={ncopies:3d}: warm {t_warm:6.1f} us ({fl / t_warm / 1e6:5.0f} TF/s) | "
          f"coldW {t_cw:6.1f} | coldA {t_ca:6.1f} | cold {t_c:6.1f} us ({fl / t_c / 1e6:5.0f} TF/s)", flush=True)
    del Ws, As

# weight-heavy small-M layers: only the weights rotate (activations of these layers are a few MB, just written)
CONVS = [  # kind, h (latent side), cin, cout
    ("conv", 4, 1280, 1280), ("conv", 4, 2560, 1280), ("conv", 8, 1280, 1280), ("conv", 8, 2560, 1280), ("conv", 16, 640, 640),
    ("tconv", 4, 1280, 1280), ("tconv", 8, 1280, 1280), ("tconv", 16, 640, 640),
    ("geglu", 8, 1280, 10240), ("geglu", 4, 1280, 10240), ("lin", 8, 5120, 1280), ("lin", 4, 5120, 1280), ("lin", 8, 1280, 3840), ("lin", 4, 1280, 3840),
]
for kind, h, cin, cout in CONVS:
    M = 32 * h * h
    taps = {"conv": 9, "tconv": 3}.get(kind, 1)
    wbytes = cout * cin * taps * 2
    ncopies = max(2, min(REPS, (600 << 20) // wbytes + 1))
    if kind == "conv":
        Ws = [pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02) for _ in range(ncopies)]
        kw = dict(k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(h, h, h, h, 1, 0))
    elif kind == "tconv":
        Ws = [pack.pack_tconv3(torch.randn(cout, cin, 3, 1, 1, device=dev) * 0.02) for _ in range(ncopies)]
        kw = dict(k=cin, taps=3, gather=ops.GATHER_TCONV3, tconv=(16, h * h))
    else:
        Ws = [(torch.randn(cout, cin, device=dev) * 0.03).to(torch.bfloat16) for _ in range(ncopies)]
        kw = dict(geglu=True) if kind == "geglu" else {}
    a = torch.randn(M, cin, device=dev).to(torch.bfloat16)
    out = torch.empty(M, cout // 2 if kind == "geglu" else cout, device=dev, dtype=torch.bfloat16)
    t_warm = timed_graph(lambda i: ops.gemm(a, Ws[0], out=out, **kw))
    t_cw = timed_graph(lambda i: ops.gemm(a, Ws[i % ncopies], out=out, **kw))
    print(f"{kind:5s} h={h:2d} {cin:5d}->{cout:5d} (M={M}, weights {wbytes / 1e6:5.1f} MB, {ncopies} copies): warm {t_warm:6.1f} us | cold weights {t_cw:6.1f} us "
          f"(+{100 * (t_cw / t_warm - 1):.0f} %)", flush=True)
    del Ws
