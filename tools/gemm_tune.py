#!/usr/bin/env python
"""Sweep the GEMM kernel configurations (ring tile x split-K) over the GEMM signatures of the 256x256 model at
the CFG-pair batch and print the time of each next to what the built-in planner picks.
    CCV_GEMM_TUNE=1 python tools/gemm_tune.py [lin] [conv] [tconv]
The planner's cost model in csrc/ccv_gemm.hip (make_plan) is fitted to this table.
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
RINGS = {-1: "128fam", 0: "128x320", 1: "64x320", 2: "128x160", 3: "64x160/4", 4: "64x160/8", 5: "128x320/2", 6: "128x160/2", 7: "64x320/2"}


def timeit(fn, iters=12, warm=2):
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


def rnd(*s, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dtype)


def cases(which):
    out = []
    if "lin" in which:
        for M, C in ((32768, 320), (8192, 640), (2048, 1280), (512, 1280)):
            out += [("lin", M, 3 * C, C, "plain"), ("lin", M, C, C, "res"), ("lin", M, 8 * C, C, "geglu"), ("lin", M, C, 4 * C, "resb")]
    if "conv" in which:
        for h, cin, cout in ((32, 320, 320), (32, 640, 320), (32, 960, 320), (16, 320, 640), (16, 640, 640), (16, 1280, 640),
                             (16, 960, 640), (8, 640, 1280), (8, 1280, 1280), (8, 2560, 1280), (8, 1920, 1280),
                             (4, 1280, 1280), (4, 2560, 1280)):
            out.append(("conv", h, cin, cout, "res"))
    if "tconv" in which:
        for h, c in ((32, 320), (16, 640), (8, 1280), (4, 1280)):
            out.append(("tconv", h, c, c, ""))
    return out


def build(case):
    kind = case[0]
    if kind == "lin":
        _, M, N, K, flavour = case
        a, w = rnd(M, K), rnd(N, K, scale=0.05)
        bias = torch.zeros(N, device=dev)
        if flavour == "geglu":
            return (lambda: ops.gemm(a, w, bias=bias, geglu=True)), 2.0 * M * N * K, f"lin   M={M:6d} N={N:6d} K={K:5d} geglu"
        if flavour == "res":
            res = torch.randn(M, N, device=dev)
            return (lambda: ops.gemm(a, w, bias=bias, residual=res, out_f32=True)), 2.0 * M * N * K, f"lin   M={M:6d} N={N:6d} K={K:5d} res  "
        if flavour == "resb":
            res = torch.randn(M, N, device=dev)
            return (lambda: ops.gemm(a, w, bias=bias, residual=res)), 2.0 * M * N * K, f"lin   M={M:6d} N={N:6d} K={K:5d} resb "
        return (lambda: ops.gemm(a, w)), 2.0 * M * N * K, f"lin   M={M:6d} N={N:6d} K={K:5d} plain"
    if kind == "conv":
        _, h, cin, cout, _ = case
        M = 32 * h * h
        a = rnd(M, cin)
        w = pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
        bias = torch.zeros(cout, device=dev)
        res = torch.randn(M, cout, device=dev)
        fn = lambda: ops.gemm(a, w, k=cin, taps=9, bias=bias, residual=res, out_f32=True, gather=ops.GATHER_CONV3X3,
                              conv=(h, h, h, h, 1, 0))
        return fn, 2.0 * M * cout * 9 * cin, f"conv  h={h:2d} {cin:5d}->{cout:5d}            "
    _, h, c, _, _ = case
    M = 2 * 16 * h * h
    a = rnd(M, c)
    w = pack.pack_tconv3(torch.randn(c, c, 3, 1, 1, device=dev) * 0.02)
    bias = torch.zeros(c, device=dev)
    fn = lambda: ops.gemm(a, w, k=c, taps=3, bias=bias, gather=ops.GATHER_TCONV3, tconv=(16, h * h))
    return fn, 2.0 * M * c * 3 * c, f"tconv h={h:2d} C={c:5d}                 "


def main():
    which = sys.argv[1:] or ["lin", "conv", "tconv"]
    for case in cases(which):
        fn, flops, label = build(case)
        os.environ.pop("CCV_GEMM_RING", None)
        os.environ.pop("CCV_GEMM_SPLIT", None)
        ref = fn().float()
        picked = ops.LAST_GEMM_PLAN
        auto = timeit(fn)
        res = []
        for ring in RINGS:
            for split in (1, 2, 4, 8):
                os.environ["CCV_GEMM_RING"] = str(ring)
                os.environ["CCV_GEMM_SPLIT"] = str(split)
                out = fn().float()
                if ops.LAST_GEMM_PLAN != (ring, split):  # configuration does not apply to this shape
                    continue
                err = ((out - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
                us = timeit(fn)
                res.append((us, ring, split, err))
        res.sort()
        best = res[0]
        txt = "  ".join(f"{RINGS[r]}/s{s}:{us:6.1f}{'!' if err > 2e-2 else ''}" for us, r, s, err in res[:6])
        print(f"{label} auto[{RINGS[picked[0]]}/s{picked[1]}] {auto:7.1f} us {flops / auto / 1e6:6.1f} TF/s | best {best[0]:7.1f} us {flops / best[0] / 1e6:6.1f} TF/s | {txt}", flush=True)


if __name__ == "__main__":
    main()
