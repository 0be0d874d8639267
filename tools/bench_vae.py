#!/usr/bin/env python
"""First-stage decode of one clip (16 frames, 32x32 latents -> 256x256 pixels) on the HIP kernels: device time with
HIP events, algorithmic FLOPs (2 M N K taps per conv / GEMM, 4 L^2 C for the single-head attention), and the oracle
timed on the host cores on ONE frame as the CPU baseline.
    python tools/bench_vae.py [--frames 16] [--iters 5] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd.vae import AutoencoderKL  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402  (cpu_baseline leg only)
from oracle.unet_oracle import seeded_state_dict  # noqa: E402


def decoder_flops(cfg, hl):
    """Algorithmic FLOPs of AutoencoderKL.decode for one frame with an hl x hl latent."""
    ch, mult, nb = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"] + 1
    c = ch * mult[-1]
    px = hl * hl
    fl = 2 * px * 4 * 4 + 2 * px * 9 * 4 * c                    # post_quant_conv, conv_in
    res = lambda cin, cout, p: 2 * p * 9 * cin * cout + 2 * p * 9 * cout * cout + (2 * p * cin * cout if cin != cout else 0)
    fl += 2 * res(c, c, px) + 4 * 2 * px * c * c + 4 * px * px * c   # mid: 2 ResnetBlocks + q,k,v,proj + attention
    for lvl in reversed(range(len(mult))):
        cout = ch * mult[lvl]
        for _ in range(nb):
            fl += res(c, cout, px)
            c = cout
        if lvl != 0:
            px *= 4
            fl += 2 * px * 9 * c * c
    return fl + 2 * px * 9 * c * 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    man = {k: list(v.shape) for k, v in AutoencoderKL(ddconfig=dict(vo.FULL_DDCONFIG), embed_dim=4).state_dict().items()}
    sd = seeded_state_dict(man, 7, std=0.03)
    vae = AutoencoderKL(ddconfig=dict(vo.FULL_DDCONFIG), embed_dim=4)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(dev).eval()
    g = torch.Generator().manual_seed(9)
    z = torch.randn(args.frames, 4, 32, 32, generator=g)
    zd = z.to(dev)
    y = vae.decode(zd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = vae.decode(zd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fl = decoder_flops(vo.FULL_DDCONFIG, 32) * args.frames
    # encode side: the conditioning frame + 2 context frames of a clip (3 images 256x256 -> posterior parameters)
    img = torch.randn(3, 3, 256, 256, generator=g).to(dev)
    vae.encode(img)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        post = vae.encode(img)
    e1.record()
    torch.cuda.synchronize()
    ms_enc = e0.elapsed_time(e1) / args.iters
    assert torch.isfinite(post.parameters).all()
    line = {"what": f"AutoencoderKL.decode, {args.frames} frames 32x32 -> 256x256 (SURVEY.md 8 f2)", "ms": ms,
            "frames_per_s": args.frames / ms * 1e3, "algorithmic_tflop": fl / 1e12, "achieved_tflops": fl / ms / 1e9,
            "frac_of_bf16_peak": fl / ms / 1e9 / 2500.0, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "encode_3_images_ms": ms_enc}
    if not args.no_cpu:
        cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        ref = vo.decode(sd, vo.FULL_DDCONFIG, z[:1])
        dt = time.perf_counter() - t0
        err = ((y[:1].float().cpu() - ref).norm() / ref.norm()).item()
        line["cpu_baseline"] = {"kind": "port", "cores": cores, "seconds_per_frame": dt, "frames_per_s": 1.0 / dt,
                                "sample": "1 frame through the fp32 oracle"}
        line["parity_rel_l2_frame0"] = err
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
