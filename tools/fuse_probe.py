#!/usr/bin/env python
"""Stacked-segment GEMM (K = 3C, one stream update) against three separate linear GEMMs with residual, at the four
resolutions of the model (b = 2), under the planner's choice and forced tiles.   python tools/fuse_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CCV_GEMM_TUNE", "1")
from camc2v_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
for M, C in ((32768, 320), (8192, 640), (2048, 1280), (512, 1280)):
    a = torch.randn(3 * M, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(C, 3 * C, device=dev) * 0.03).to(torch.bfloat16)
    ws = [w[:, i * C:(i + 1) * C].contiguous() for i in range(3)]
    bias = torch.zeros(C, device=dev)
    stream = torch.zeros(M, C, device=dev)

    def sep():
        for i in range(3):
            ops.gemm(a[i * M:(i + 1) * M], ws[i], bias=bias, residual=stream, out_f32=True, out=stream)

    def fused():
        ops.gemm(a, w, k=C, taps=3, m=M, gather=ops.GATHER_SEGMENTS, seg_rows=M, bias=bias, residual=stream, out_f32=True, out=stream)

    line = f"M={M:6d} C={C:5d}: 3 separate {timeit(sep):7.1f} us | fused auto {timeit(fused):7.1f}"
    ops.TRACK_GEMM_PLAN = True
    fused()
    line += f" plan={ops.LAST_GEMM_PLAN}"
    ops.TRACK_GEMM_PLAN = False
    for name, env in (("fam(auto tile)", {"CCV_GEMM_RING": "-1", "CCV_GEMM_F160": "0"}), ("fam45", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "45"}),
                      ("fam42", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "42"}), ("fam44", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "44"}),
                      ("fam24", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "24"}), ("fam22", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "22"}),
                      ("ring2", {"CCV_GEMM_RING": "2"}), ("ring6", {"CCV_GEMM_RING": "6"}), ("ring5", {"CCV_GEMM_RING": "5"}),
                      ("ring2 s2", {"CCV_GEMM_RING": "2", "CCV_GEMM_SPLIT": "2"}), ("fam24 s2", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "24", "CCV_GEMM_SPLIT": "2"}),
                      ("fam22 s3", {"CCV_GEMM_RING": "-1", "CCV_GEMM_FAMTILE": "22", "CCV_GEMM_SPLIT": "3"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            line += f" | {name} {timeit(fused):6.1f}"
        except Exception as e:   # a forced tile that does not fit the shape
            line += f" | {name} n/a"
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print(line, flush=True)
