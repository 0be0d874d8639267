#!/usr/bin/env python
"""Sum kernel time by family from a summarize_rocprof.py summary: total, GEMM (gemm_* incl. split-K reduce), rest."""
import sys

for path in sys.argv[1:]:
    tot = gemm = 0.0
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel "):
            if line.startswith("# per-grid") or line.startswith("# counter"):
                break
            continue
        parts = line.rsplit(None, 6)
        if len(parts) != 7:
            continue
        ms = float(parts[2])
        tot += ms
        if parts[0].startswith("gemm_"):
            gemm += ms
    print(f"{path}: total {tot:8.1f} ms  gemm {gemm:8.1f} ms  other {tot - gemm:8.1f} ms")
