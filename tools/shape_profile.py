#!/usr/bin/env python
"""Per-call-signature device time of one CFG-pair UNet forward at the benchmark size (eager, HIP events around
every op wrapper).  Shows which GEMM / attention / norm shapes the clip time is made of.
    python tools/shape_profile.py [--reps 3]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camc2v_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)

records = []  # (key, start_event, end_event, flops)
recording = False


def timed(name, fn, keyfn):
    def wrapper(*a, **kw):
        if not recording:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        key, flops = keyfn(*a, **kw)
        records.append((f"{name} {key}", e0, e1, flops))
        return out
    return wrapper


def gemm_key(a, w, *, k=None, taps=1, m=None, bias=None, bias2=None, residual=None, act=0, geglu=False, out_f32=False,
             out=None, gather=0, conv=None, **kw):
    N = w.shape[0]
    K = k if k is not None else w.shape[1] // taps
    M = m if m is not None else a.shape[0]
    tag = ("f32A " if a.dtype == torch.float32 else "") + ("geglu " if geglu else "") + ("res " if residual is not None else "") + \
          ("of32 " if out_f32 else "") + (f"g{gather} " if gather else "")
    if conv is not None and (conv[4] != 1 or conv[5] != 0):
        tag += f"s{conv[4]}u{conv[5]} "
    return f"M={M} N={N} K={K} taps={taps} {tag}", 2.0 * M * N * K * taps


def attn_key(q, k, v, *, B, inner, H, Lq, Lk, k2=None, Lk2=0, mask_bits=None, wave_bits=None, variant=0, **kw):
    tag = ("masked " if mask_bits is not None else "") + ("sparse " if wave_bits is not None else "") + (f"+{Lk2} " if k2 is not None else "")
    return f"B={B} inner={inner} H={H} Lq={Lq} Lk={Lk} v{variant} {tag}", 4.0 * B * inner * H * Lq * (Lk + Lk2) * 64


def norm_key(x, *a, **kw):
    return f"{tuple(x.shape)} {str(x.dtype)[6:]}", 0.0


ops.gemm = timed("gemm", ops.gemm, gemm_key)
ops.attention = timed("attn", ops.attention, attn_key)
ops.groupnorm = timed("gn", ops.groupnorm, norm_key)
ops.layernorm = timed("ln", ops.layernorm, norm_key)

model = bench.build_model(dev)
cond, uncond, fs, x_T, noises = bench.synthetic_inputs(model, dev)
t = torch.full((1,), 439, dtype=torch.long, device=dev)
uncond = dict(uncond)
uncond["camera_condition"] = dict(cond["camera_condition"], is_uc=True)   # what p_sample_ddim does (reference ddim.py:259-260)


def forward():
    return model.apply_model_pair(x_T, t, cond, uncond, fs=fs)


for _ in range(2):
    forward()
torch.cuda.synchronize()
w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
w0.record()
forward()
w1.record()
torch.cuda.synchronize()
print(f"untimed-wrapper forward: {w0.elapsed_time(w1):.2f} ms")

agg = collections.OrderedDict()
for _ in range(args.reps):
    records.clear()
    recording = True
    forward()
    recording = False
    torch.cuda.synchronize()
    for key, e0, e1, flops in records:
        a = agg.setdefault(key, [0, 0.0, flops])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e3
total = sum(v[1] for v in agg.values()) / args.reps
print(f"sum of wrapped ops: {total / 1e3:.2f} ms per forward ({len(agg)} signatures)")
print(f"{'calls':>5} {'avg_us':>8} {'tot_us':>9} {'pct':>5} {'TF/s':>7}  signature")
for key, (n, us, flops) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    calls = n // args.reps
    tot = us / args.reps
    avg = us / n
    tf = flops / avg / 1e6 if flops else 0.0
    print(f"{calls:5d} {avg:8.1f} {tot:9.1f} {100 * tot / total:5.1f} {tf:7.1f}  {key}")
