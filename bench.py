#!/usr/bin/env python
"""Benchmark of the hot path: 25-step DDIM sampling with classifier-free guidance of CamContextI2V-256
clips (1 x 16 x 256 x 256, CFG 7.5, guidance rescale 0.7, eta 1) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic input = one full 25-step DDIM sampling of
one clip per GPU (50 UNet forwards' worth of work, run as 25 batched cond+uncond forwards).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line (see the driver contract).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20230211          # reference default seed (main/trainer.py:21)
N_CONTEXT = 2            # extra context frames -> cond context 77 + 256*(1+N) tokens
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, MI355X_MICROARCH.md
# (2 * 7.954e8 + 4.559e8) KB per clip: profiles/r01_rocprofv3_pmc_{FETCH,WRITE}_SIZE_bench_eager.txt
# (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; separate --pmc passes, one clip each)
TRAFFIC_BYTES_PER_CLIP = (2 * 7.954437e8 + 4.558582e8) * 1024


def build_model(device, unet_params=None):
    from camc2v_amd import configs
    from utils.utils import instantiate_from_config
    torch.manual_seed(SEED)
    with torch.device(device):
        model = instantiate_from_config(configs.camcontexti2v_256(unet_params))
    g = torch.Generator(device=device).manual_seed(SEED)
    with torch.no_grad():  # seeded N(0, 0.02) weights incl. the zero-initialised tensors; norm gains ~ 1
        for name, p in model.model.diffusion_model.named_parameters():
            p.normal_(0.0, 0.02, generator=g)
            if p.dim() == 1 and name.endswith(".weight"):
                p.add_(1.0)
    model.eval()
    model.model.diffusion_model.prepare()
    return model


def synthetic_inputs(model, device, b=1, t=16, hl=32, rank=0):
    """SURVEY.md section 8(d) synthetic clip."""
    from camc2v_amd import camera
    g = torch.Generator(device=device).manual_seed(SEED + 17 * rank)
    rn = lambda *s: torch.randn(*s, device=device, generator=g)
    ctx_dim = 1024
    img = torch.nn.functional.layer_norm(rn(b, 256 * (1 + N_CONTEXT), ctx_dim), (ctx_dim,))
    img_u = torch.nn.functional.layer_norm(rn(b, 16 * t, ctx_dim), (ctx_dim,))
    cond_ctx = torch.cat([rn(b, 77, ctx_dim), img], 1).contiguous()
    uncond_ctx = torch.cat([rn(b, 77, ctx_dim), img_u], 1).contiguous()
    c_concat = (rn(b, 4, t, hl, hl) * 0.18215).contiguous()
    chans = [320, 640, 1280, 1280]
    feats = [(rn(b, chans[i], t, hl >> i, hl >> i) * 0.1).contiguous() for i in range(4)]
    px = 8 * hl
    K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]], device=device).repeat(b, t, 1, 1)
    w2c = camera.synthetic_trajectory(b, t, device)
    cam = model.camera_condition(K, w2c, torch.zeros(b, dtype=torch.long, device=device), px, px,
                                 pluker_features=feats, generator=g)
    cond = dict(c_concat=[c_concat], c_crossattn=[cond_ctx], camera_condition=cam)
    uncond = dict(c_concat=[c_concat], c_crossattn=[uncond_ctx])
    fs = torch.full((b,), 8, dtype=torch.long, device=device)
    x_T = rn(b, 4, t, hl, hl)
    noises = [rn(b, 4, t, hl, hl) for _ in range(25)]
    return cond, uncond, fs, x_T, noises


def sample_clip(model, cond, uncond, fs, x_T, noises, use_graph):
    from camc2v_amd import configs
    kw = dict(configs.GENERATION_KWARGS)
    steps = kw.pop("ddim_steps")
    samples, _ = model.sample_log(cond, x_T.shape[0], True, steps, x_T=x_T, unconditional_conditioning=uncond,
                                  fs=fs, injected_noise=noises, use_graph=use_graph, **kw)
    return samples


def cpu_baseline(model, device):
    """Oracle (CPU restatement, fp32) timed on the host cores on ONE UNet forward of the reference's
    CPU-runnable case (no camera, ctx 77+16t): bounded sample, ~10-30 s.  Also reports the full-size
    eps parity of the HIP path on exactly that forward."""
    from oracle import unet_oracle
    from camc2v_amd import configs
    unet = model.model.diffusion_model
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(1, 8, 16, 32, 32, generator=g)
    ctx = torch.randn(1, 77 + 256, 1024, generator=g)
    ctx_cond = torch.randn(1, 77 + 256 * (1 + N_CONTEXT), 1024, generator=g)    # the conditional pass's context length
    t = torch.tensor([439])
    fs = torch.tensor([8])
    # the GPU box gives a 1-GPU job a share of 16 host cores; more threads than that only oversubscribe
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = unet_oracle.unet_forward(sd, configs.UNET_256, x, t, ctx, fs, None)
    dt_u = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
        unet_oracle.unet_forward(sd, configs.UNET_256, x, t, ctx_cond, fs, None)
    dt_c = time.perf_counter() - t0
    with torch.no_grad():
        got = unet(x.to(device), t.to(device), context=ctx.to(device), fs=fs.to(device)).float().cpu()
    rel_l2 = ((got - ref).norm() / ref.norm()).item()
    max_rel = ((got - ref).abs().max() / ref.abs().max()).item()
    # the metric's clip needs 25 x (cond + uncond) forwards; the camera-conditioned forwards of the metric cost more than
    # these two (Pluecker projections + 2 TF of epipolar attention each), so this is an upper bound for the CPU
    value = 16.0 / (25.0 * (dt_u + dt_c))
    return dict(value=value, unit="frames/s", cores=cores, kind="port",
                sample=f"2 UNet forwards (no camera, b=1, fp32 oracle): ctx 333 = {dt_u:.2f} s, ctx {ctx_cond.shape[1]} = {dt_c:.2f} s; "
                       f"value = 16 frames / (25 x their sum), an upper bound for the CPU on the CFG+camera clip",
                seconds_per_forward=0.5 * (dt_u + dt_c)), dict(rel_l2=rel_l2, max_rel=max_rel, case="full-size UNet forward, no camera, vs fp32 oracle")


def timed_clips(sample_fn, steps, warmup, dist=None, sync=None):
    """W untimed + exactly K timed calls of `sample_fn`, bracketed by sync + barrier + sync on both sides;
    returns (max-over-ranks wall seconds, this rank's wall seconds, last output).  Device independent so the
    multi-process logic is testable on CPU with gloo."""
    sync = sync or (lambda: None)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        sample_fn()
    fence()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = sample_fn()
    fence()
    mine = time.perf_counter() - t0
    worst = mine
    if dist is not None:
        tt = torch.tensor([mine], dtype=torch.float64)
        if sync is not None and torch.cuda.is_available() and dist.get_backend() == "nccl":
            tt = tt.cuda()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        worst = float(tt.item())
    return worst, mine, out


def dominant_kernel(device, launches=50):
    """The GEMM instantiation with the largest share of the clip (profiles/r01_rocprofv3_kernel_stats_bench_graph.txt):
    gemm_dma_kernel<4, 2, 0> (128x64 tiles: the N = 320 / 960 projections at 32x32 latents), timed live on the fused QKV
    projection,
    M = 2 clips x 16 frames x 1024 tokens, N = 3 x 320, K = 320 -- with HIP events around a hipGraph of `launches`
    back-to-back launches on the current stream (algorithmic FLOPs 2 M N K per launch)."""
    from camc2v_amd import ops
    M, N, K = 32768, 960, 320
    g = torch.Generator(device=device).manual_seed(SEED)
    a = torch.randn(M, K, device=device, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=device, generator=g) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=device, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(a, w, out=out)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(launches):
            fn()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / launches
    tf = 2.0 * M * N * K / us / 1e6
    return {"kernel": "gemm_dma_kernel<4, 2, 0>", "problem": f"QKV projection M={M} N={N} K={K} (bf16 in/out)",
            "flops_per_launch": 2.0 * M * N * K, "us_per_launch": us, "achieved": tf, "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS}


def result_line(elapsed, steps, warmup, world, use_graph, dev_ms=None):
    from camc2v_amd import configs
    clips = steps * world
    tf_per_clip = 25 * (configs.TFLOP_COND_N2 + configs.TFLOP_UNCOND_CAM)
    per_clip_s = (dev_ms / 1e3 / steps) if dev_ms is not None else elapsed / steps
    achieved = tf_per_clip / per_clip_s  # one GPU's rate: algorithmic TFLOP of a clip / its device time
    return {
        "metric": "denoised video frames/sec at 16x256x256, 25 DDIM steps, CFG=7.5",
        "value": 16.0 * clips / elapsed, "unit": "frames/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "CamContextI2V-256 (camera + 2 context frames), 1 clip x 16 frames x 256x256 per GPU, "
                               "25 DDIM steps, CFG 7.5, guidance_rescale 0.7, eta 1.0; seeded N(0,0.02) weights",
                   "clips_per_gpu": 1, "parallelism": f"clip-dp{world}", "launch": "hipGraph" if use_graph else "eager"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": TRAFFIC_BYTES_PER_CLIP,
                     "kernel": "whole DDIM path (all launches of 25 CFG steps); algorithmic 375 TFLOP/clip, masks counted "
                               "dense; traffic = L2<->fabric bytes per clip from rocprofv3 PMC passes (profiles/r01_*pmc*): "
                               "2*FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included",
                     "device_ms_per_clip": 1e3 * per_clip_s},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    torch.set_grad_enabled(False)
    model = build_model(device)
    cond, uncond, fs, x_T, noises = synthetic_inputs(model, device, rank=rank)
    use_graph = not args.no_graph

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    state = {"n": 0}

    def one_clip():
        if state["n"] == args.warmup:          # first timed clip: HIP event on the launch stream
            ev0.record()
        state["n"] += 1
        return sample_clip(model, cond, uncond, fs, x_T, noises, use_graph)

    elapsed, _, out = timed_clips(one_clip, args.steps, args.warmup, dist, torch.cuda.synchronize)
    ev1.record()
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1)
    assert torch.isfinite(out).all()

    if rank == 0:
        line = result_line(elapsed, args.steps, args.warmup, world, use_graph, dev_ms)
        line["roofline"]["dominant_kernel"] = dominant_kernel(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity_full_size"] = cpu_baseline(model, device)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
