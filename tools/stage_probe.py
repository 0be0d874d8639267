#!/usr/bin/env python
"""LDS stages of the family GEMM kernel (csrc/ccv_gemm.hip gemm_dma_kernel<..., ST>): the two-stage loop against the 3- / 4-stage
ring with counted waits, per layer shape of the 32x32 ... 4x4-latent levels, with weights AND activations rotating through > 256 MB
of copies (cold, as in the model).  hipGraph of 72 launches per arm; arms whose plan is not a family tile print the same time.
    CCV_GEMM_TUNE=1 python tools/stage_probe.py
"""
import os
import sys

os.environ["CCV_GEMM_TUNE"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ops.TRACK_GEMM_PLAN = True
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


LAYERS = [  # kind, latent side, cin, cout
    ("conv", 4, 1280, 1280), ("conv", 4, 2560, 1280), ("conv", 8, 1280, 1280), ("conv", 8, 2560, 1280), ("conv", 8, 1920, 1280), ("conv", 8, 640, 1280),
    ("conv", 16, 640, 640), ("conv", 16, 1280, 640), ("conv", 16, 1920, 640),
    ("tconv", 4, 1280, 1280), ("tconv", 8, 1280, 1280), ("tconv", 16, 640, 640),
    ("geglu", 8, 1280, 10240), ("geglu", 4, 1280, 10240), ("lin", 8, 5120, 1280), ("lin", 4, 5120, 1280), ("lin", 8, 1280, 3840), ("lin", 4, 1280, 3840),
    ("res", 8, 1280, 1280), ("res", 4, 1280, 1280), ("res", 16, 640, 640), ("lin", 16, 640, 1920), ("geglu", 16, 640, 5120), ("lin", 16, 2560, 640),
    ("res", 8, 3840, 1280), ("res", 4, 3840, 1280),
    ("res", 32, 320, 320), ("res", 32, 1280, 320), ("lin", 32, 320, 960), ("tconv", 32, 320, 320), ("conv", 32, 320, 320), ("conv", 32, 640, 320),
]
for kind, h, cin, cout in LAYERS:
    M = 32 * h * h
    taps = {"conv": 9, "tconv": 3}.get(kind, 1)
    wbytes = cout * cin * taps * 2
    ncopies = max(2, min(REPS, (600 << 20) // wbytes + 1))
    kw = {}
    if kind == "conv":
        Ws = [pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02) for _ in range(ncopies)]
        kw = dict(k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(h, h, h, h, 1, 0))
    elif kind == "tconv":
        Ws = [pack.pack_tconv3(torch.randn(cout, cin, 3, 1, 1, device=dev) * 0.02) for _ in range(ncopies)]
        kw = dict(k=cin, taps=3, gather=ops.GATHER_TCONV3, tconv=(16, h * h))
    else:
        Ws = [(torch.randn(cout, cin, device=dev) * 0.03).to(torch.bfloat16) for _ in range(ncopies)]
        if kind == "geglu":
            kw = dict(geglu=True)
    na = max(2, min(REPS, (300 << 20) // (M * cin * 2) + 1))
    As = [torch.randn(M, cin, device=dev).to(torch.bfloat16) for _ in range(na)]
    a = As[0]
    if kind == "res":
        stream = torch.randn(M, cout, device=dev)
        out = torch.empty_like(stream)
        kw = dict(residual=stream, out_f32=True)
    else:
        out = torch.empty(M, cout // 2 if kind == "geglu" else cout, device=dev, dtype=torch.bfloat16)
    res = {}
    for stg in (2, 3, 4):
        os.environ["CCV_GEMM_ST"] = str(stg)
        ref = ops.gemm(a, Ws[0], out=out, **kw).float().clone()
        if stg == 2:
            ref0 = ref
        else:
            assert torch.equal(ref, ref0), "the stage count changed the result"
        plan = ops.LAST_GEMM_PLAN
        res[stg] = timed_graph(lambda i: ops.gemm(As[i % na], Ws[i % ncopies], out=out, **kw))
    os.environ.pop("CCV_GEMM_ST")
    fl = 2.0 * M * cout * cin * taps
    print(f"{kind:5s} h={h:2d} {cin:5d}->{cout:5d} (M={M:5d}, W {wbytes / 1e6:5.1f} MB x{ncopies:2d}) plan tile {plan[0]:2d} split {plan[1]:2d}: "
          f"2 stages {res[2]:6.1f} us ({fl / res[2] / 1e6:5.0f} TF/s) | 3: {res[3]:6.1f} us ({100 * (res[3] / res[2] - 1):+.0f} %) | 4: {res[4]:6.1f} us "
          f"({100 * (res[4] / res[2] - 1):+.0f} %)", flush=True)
    del Ws, As
