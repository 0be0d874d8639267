#!/usr/bin/env python
"""Group a tools/shape_profile.py log by op family: python tools/shape_summary.py gpurun_out/shapes.log"""
import collections
import re
import sys

tot = collections.Counter()
fl = collections.Counter()
for line in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\w+) (.*)", line)
    if not m:
        continue
    calls, avg, t, pct, tf, kind, sig = m.groups()
    t = float(t)
    if kind == "gemm":
        taps = int(re.search(r"taps=(\d+)", sig).group(1))
        M = int(re.search(r"M=(\d+)", sig).group(1))
        if M <= 768 and M != 512:
            kind = "gemm ctx/emb"
        elif "f32A" in sig:
            kind = "gemm f32A"
        elif taps == 9:
            kind = "conv3x3"
        elif taps == 3:
            kind = "tconv"
        elif "geglu" in sig:
            kind = "lin geglu"
        elif "res" in sig:
            kind = "lin res"
        else:
            kind = "lin plain"
    if kind == "attn":
        kind = "attn sparse" if "sparse" in sig else ("attn temporal" if "Lq=16 " in sig else ("attn cross" if "+" in sig else "attn self"))
    tot[kind] += t
    fl[kind] += float(tf) * t  # TF/s * us
s = sum(tot.values())
for k, v in tot.most_common():
    print(f"{k:16s} {v / 1e3:7.2f} ms {100 * v / s:5.1f}%  {fl[k] / v:7.1f} TF/s avg")
print(f"total {s / 1e3:.2f} ms")
