#!/usr/bin/env python
"""A/B probe (one box, graph-timed): planner's choice vs the 160-column family tiles (CCV_GEMM_FAMTILE=45/25: 2-stage,
64-deep slabs = full 128-byte rows) on the model's GEMM signatures.
    python tools/ring_probe.py [lin] [conv] [tconv]
"""
import os
import sys

os.environ["CCV_GEMM_TUNE"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402
from tools.gemm_tune import build, cases  # noqa: E402

ARMS = [("auto", {}), ("fam128x160", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "45"}), ("fam64x160", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "25"}),
        ("fam128x160/s2", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "45", "CCV_GEMM_SPLIT": "2"}),
        ("fam128x160/s4", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "45", "CCV_GEMM_SPLIT": "4"})]


def main():
    which = sys.argv[1:] or ["lin", "conv", "tconv"]
    for case in cases(which):
        if case[0] == "lin" and case[4] == "geglu":
            continue
        fn, flops, label = build(case)
        line, ref = label, None
        for name, env in ARMS:
            for k in ("CCV_GEMM_RING", "CCV_GEMM_FAMTILE", "CCV_GEMM_SPLIT"):
                os.environ.pop(k, None)
            os.environ.update(env)
            out = fn().float()
            if ref is None:
                ref = out
            err = ((out - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
            us = min(timeit(fn, iters=40), timeit(fn, iters=40))
            line += f" | {name} {us:6.1f}{'!' if err > 2e-2 else ''}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
