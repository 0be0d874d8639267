#!/usr/bin/env python
"""Full-size determinism / soak check: sample the benchmark clip N times (hipGraph replay) and compare the latents bit for bit.
    python tools/determinism_check.py [N]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    model = bench.build_model(dev)
    cond, uncond, fs, x_T, noises = bench.synthetic_inputs(model, dev, rank=0)
    ref = None
    for i in range(n):
        out = bench.sample_clip(model, cond, uncond, fs, x_T, noises, True).clone()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        if ref is None:
            ref = out
        same = torch.equal(out, ref)
        print(f"clip {i}: absmax {out.abs().max().item():.4f}  identical to clip 0: {same}", flush=True)
        if not same:
            d = (out - ref).abs().max().item()
            print(f"  max abs difference {d:.3e}")
            raise SystemExit(1)


if __name__ == "__main__":
    main()
