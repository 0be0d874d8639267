#!/usr/bin/env python
"""Benchmark of the hot path: 25-step DDIM sampling with classifier-free guidance of CamContextI2V-256
clips (1 x 16 x 256 x 256, CFG 7.5, guidance rescale 0.7, eta 1) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input = one full 25-step DDIM sampling of
one clip per GPU (50 UNet forwards' worth of work, run as 25 batched cond+uncond forwards).  EVERY clip has its
own conditioning tensors (text / image tokens, context latents, Pluecker features, fundamental matrices and masks,
start noise), all resident in HBM before the timed region; the once-per-clip work the sampler does with them (copy
into the hipGraph's static buffers, context K/V projections of the 16 cross-attention layers, Pluecker rows) is
inside the timed region.  Rank 0 prints ONE JSON line (see the driver contract).

N > 1: one process per GPU.  Started by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) it joins the job; started plainly with `--gpus N` it launches that command itself as a CHILD process
before touching the GPU and relays the child's JSON line.  Clips shard across ranks with no collective on the data
path (`scaling: weak`); one RCCL all_gather of the final latents closes the timed region so that the line can
report how many ranks took part.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20230211          # reference default seed (main/trainer.py:21)
N_CONTEXT = 2            # extra context frames -> cond context 77 + 256*(1+N) tokens
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, MI355X_MICROARCH.md
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")   # PMC-derived bytes of this round's profile run


# BASELINE.json configs -> concrete runs (SURVEY.md section 8d "Configs -> concrete runs").  c1 is the metric's configuration and the
# default; the others are supplementary lines (`python bench.py --config c3`), committed under profiles/.
WORKLOADS = {
    "c1": dict(target="model.camcontexti2v.CamContextI2V", T=16, cfg=7.5, camera=True, cond_ctx=77 + 256 * (1 + N_CONTEXT), uncond_ctx=77 + 256,
               metric="denoised video frames/sec at 16x256x256, 25 DDIM steps, CFG=7.5",
               label="CamContextI2V-256 (camera + 2 context frames), 1 clip x 16 frames x 256x256 per GPU, 25 DDIM steps, CFG 7.5, "
                     "guidance_rescale 0.7, eta 1.0"),
    "c0": dict(target="model.dynamicrafter.DynamiCrafter", T=16, cfg=1.0, camera=False, cond_ctx=77 + 256, uncond_ctx=None,
               metric="denoised video frames/sec at 16x256x256, 25 DDIM steps, CFG off (DynamiCrafter 256 base, BASELINE.json configs[0])",
               label="DynamiCrafter-256 base (no camera), 1 clip x 16 frames x 256x256, 25 DDIM steps, CFG off (25 UNet forwards, b = 1), "
                     "context 77 + 16 x 16 tokens per frame, eta 1.0"),
    "c3": dict(target="baseline.cami2v.CamI2V", T=16, cfg=7.5, camera=True, cond_ctx=77 + 256, uncond_ctx=77 + 256,
               metric="denoised video frames/sec at 16x256x256, 25 DDIM steps, CFG=7.5 (CamI2V 256 baseline, BASELINE.json configs[3])",
               label="CamI2V-256 baseline (configs/baseline/cami2v_256.yaml: Pluecker features + epipolar attention, context 77 + 16 x 16 "
                     "per frame on both CFG passes, no context-frame adaptor), 1 clip x 16 frames x 256x256, 25 DDIM steps, CFG 7.5, "
                     "guidance_rescale 0.7, eta 1.0"),
    "c4": dict(target="model.camcontexti2v.CamContextI2V", T=32, cfg=3.5, camera=True, cond_ctx=77 + 16 * 32, uncond_ctx=77 + 16 * 32,
               metric="denoised video frames/sec at 32x256x256, 25 DDIM steps, CFG=3.5 (BASELINE.json configs[4], UNet level)",
               label="CamContextI2V-256 UNet at 32 frames (SURVEY.md section 8d C5: the reference cannot run t = 32 end to end; synthetic pose "
                     "features and epipolar masks of a 32-frame trajectory, context 77 + 16 x 32 on both CFG passes), 1 clip x 32 frames x "
                     "256x256, 25 DDIM steps, CFG 3.5, guidance_rescale 0.7, eta 1.0; bf16 epipolar attention (the e4m3 variant was "
                     "measured slower and retired, profiles/r04_fp8_retired.txt)"),
}


def workload_tflop_per_clip(w):
    """Algorithmic TFLOP of one clip by SURVEY.md section 8(d)'s dense convention."""
    from camc2v_amd import configs
    if w["T"] == 32:
        # per forward at t = 32: everything per-frame doubles (2 x (4.904 no-camera + 0.256 camera linears)), the epipolar attention over
        # T*h*w tokens quadruples (4 x 1.963); both CFG passes carry the camera and the per-frame context
        per_fwd = 2 * (configs.TFLOP_NOCAM + 0.256) + 4 * 1.963
        return 25 * 2 * per_fwd
    if not w["camera"]:
        return 25 * configs.TFLOP_NOCAM
    if w["cond_ctx"] == 77 + 256:
        return 25 * 2 * configs.TFLOP_UNCOND_CAM
    return 25 * (configs.TFLOP_COND_N2 + configs.TFLOP_UNCOND_CAM)


WORKLOAD = WORKLOADS["c1"]       # set by main() from --config
WORKLOAD_KEY = "c1"


def build_model(device, unet_params=None, workload=None):
    from camc2v_amd import configs
    from utils.utils import instantiate_from_config
    w = workload or WORKLOAD
    torch.manual_seed(SEED)
    cfg = configs.camcontexti2v_256(unet_params if unet_params is not None else dict(configs.UNET_256, temporal_length=w["T"]))
    cfg["target"] = w["target"]
    cfg["params"]["temporal_length"] = w["T"]
    if w["target"] != "model.camcontexti2v.CamContextI2V":      # the baselines take no context-frame arguments
        for k in ("multi_cond_strategy", "use_zero_conv_latent_input"):
            cfg["params"].pop(k, None)
    if not w["camera"]:
        for k in ("add_type", "pose_encoder_config", "epipolar_config"):
            cfg["params"].pop(k, None)
    with torch.device(device):
        model = instantiate_from_config(cfg)
    g = torch.Generator(device=device).manual_seed(SEED)
    with torch.no_grad():  # seeded N(0, 0.02) weights incl. the zero-initialised tensors; norm gains ~ 1
        for name, p in model.model.diffusion_model.named_parameters():
            p.normal_(0.0, 0.02, generator=g)
            if p.dim() == 1 and name.endswith(".weight"):
                p.add_(1.0)
    model.eval()
    model.model.diffusion_model.prepare()
    return model


def synthetic_inputs(model, device, b=1, t=None, hl=32, rank=0, clip=0, workload=None):
    """SURVEY.md section 8(d) synthetic clip; (rank, clip) select the random draws, the camera trajectory is the fixed one."""
    from camc2v_amd import camera
    w = workload or WORKLOAD
    t = w["T"] if t is None else t
    g = torch.Generator(device=device).manual_seed(SEED + 17 * rank + 1009 * clip)
    rn = lambda *s: torch.randn(*s, device=device, generator=g)
    ctx_dim = 1024

    def context(L):
        img = torch.nn.functional.layer_norm(rn(b, L - 77, ctx_dim), (ctx_dim,))
        return torch.cat([rn(b, 77, ctx_dim), img], 1).contiguous()
    if w is WORKLOADS["c1"]:      # (draw order of rounds 1-3 kept: the committed parity numbers refer to these tensors)
        img = torch.nn.functional.layer_norm(rn(b, 256 * (1 + N_CONTEXT), ctx_dim), (ctx_dim,))
        img_u = torch.nn.functional.layer_norm(rn(b, 16 * t, ctx_dim), (ctx_dim,))
        cond_ctx = torch.cat([rn(b, 77, ctx_dim), img], 1).contiguous()
        uncond_ctx = torch.cat([rn(b, 77, ctx_dim), img_u], 1).contiguous()
    else:
        cond_ctx = context(w["cond_ctx"])
        uncond_ctx = context(w["uncond_ctx"]) if w["uncond_ctx"] else None
    c_concat = (rn(b, 4, t, hl, hl) * 0.18215).contiguous()
    cond = dict(c_concat=[c_concat], c_crossattn=[cond_ctx])
    if w["camera"]:
        chans = [320, 640, 1280, 1280]
        feats = [(rn(b, chans[i], t, hl >> i, hl >> i) * 0.1).contiguous() for i in range(4)]
        px = 8 * hl
        K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]], device=device).repeat(b, t, 1, 1)
        w2c = camera.synthetic_trajectory(b, t, device)
        cond["camera_condition"] = model.camera_condition(K, w2c, torch.zeros(b, dtype=torch.long, device=device), px, px,
                                                          pluker_features=feats, generator=g)
    uncond = dict(c_concat=[c_concat], c_crossattn=[uncond_ctx]) if uncond_ctx is not None else None
    fs = torch.full((b,), 8, dtype=torch.long, device=device)
    x_T = rn(b, 4, t, hl, hl)
    noises = [rn(b, 4, t, hl, hl) for _ in range(25)]
    return cond, uncond, fs, x_T, noises


def sample_clip(model, cond, uncond, fs, x_T, noises, use_graph):
    from camc2v_amd import configs
    kw = dict(configs.GENERATION_KWARGS)
    steps = kw.pop("ddim_steps")
    kw["unconditional_guidance_scale"] = WORKLOAD["cfg"]
    kw["enable_camera_condition"] = WORKLOAD["camera"]
    samples, _ = model.sample_log(cond, x_T.shape[0], True, steps, x_T=x_T, unconditional_conditioning=uncond,
                                  fs=fs, injected_noise=noises, use_graph=use_graph, **kw)
    return samples


def cfg_step(model, device, inputs, t_value=439):
    """One sampler step's UNet work on the clip `inputs`: the cond + uncond pair the sampler runs under CFG, one forward without."""
    cond, uncond, fs, x_T, _ = inputs
    t = torch.full((x_T.shape[0],), t_value, dtype=torch.long, device=device)
    if uncond is None:
        return (model.apply_model(x_T, t, cond, fs=fs),)
    uc = dict(uncond, camera_condition=dict(cond["camera_condition"], is_uc=True)) if WORKLOAD["camera"] else uncond
    return model.apply_model_pair(x_T, t, cond, uc, fs=fs, enable_camera_condition=WORKLOAD["camera"])


def host_cores():
    # the GPU box gives a 1-GPU job a share of 16 host cores; more threads than that only oversubscribe
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


def cpu_baseline(model, device, inputs):
    """The oracle (fp32 CPU restatement of the reference's UNet) timed on the host cores on the workload's own step: ONE sampler
    step of the benchmark clip -- under CFG the conditional + the unconditional forward (c1: camera-conditioned, context 77 + 768 and
    77 + 256 tokens; c3: 77 + 256 both), without CFG (c0) the one forward -- same weights, inputs and epipolar masks as the GPU run
    (bounded sample: 1 of the clip's 25 steps).  The same outputs give the full-size parity of the HIP path (what the sampler runs)."""
    from oracle import geometry_oracle, unet_oracle
    from camc2v_amd import configs
    w = WORKLOAD
    if w["T"] != 16:
        return dict(value=None, unit="frames/s", cores=host_cores(), kind="port",
                    sample="not run: the oracle's dense fp32 epipolar attention over 32768 x 32768 scores (4.3 GB per head) does not fit a "
                           "bounded CPU sample; configs[4] is held to the oracle at reduced width in tests/test_unet_gpu.py"), None
    cond, uncond, fs, x_T, _ = inputs
    unet = model.model.diffusion_model
    t = torch.full((1,), 439, dtype=torch.long, device=device)
    with torch.no_grad():
        got = [e.float().cpu() for e in cfg_step(model, device, inputs)]
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    cam_cpu = None
    if w["camera"]:
        cam = cond["camera_condition"]
        F = cam["fundamental"].float().cpu()
        masks = {d: geometry_oracle.epipolar_mask(F, 256 // d, 256 // d, d) for d in (8, 16, 32, 64)}
        cam_cpu = dict(pluker_embedding_features=[f.float().cpu() for f in cam["pluker_embedding_features"]],
                       sample_locs_dict=masks, add_type=cam["add_type"])
    x = torch.cat([x_T, cond["c_concat"][0]], 1).float().cpu()
    cores = host_cores()
    torch.set_num_threads(cores)
    secs, parity = [], {}
    passes = [(got[0], cond["c_crossattn"][0], "cond")] + ([(got[1], uncond["c_crossattn"][0], "uncond")] if uncond is not None else [])
    for g_, ctx, what in passes:
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = unet_oracle.unet_forward(sd, dict(configs.UNET_256), x, t.cpu(), ctx.float().cpu(), fs.cpu(), cam_cpu, origin_h=256)
        secs.append(time.perf_counter() - t0)
        parity[f"{what}_rel_l2"] = ((g_ - ref).norm() / ref.norm()).item()
        parity[f"{what}_max_rel"] = ((g_ - ref).abs().max() / ref.abs().max()).item()
    ctxs = " + ".join(f"{what} ctx {c.shape[1]}" for _, c, what in passes)
    parity["case"] = f"full-size sampler step ({ctxs}, b=1, 32x32 latents{', native packed masks' if w['camera'] else ''}) vs the fp32 oracle"
    value = 16.0 / (25.0 * sum(secs))
    return dict(value=value, unit="frames/s", cores=cores, kind="port",
                sample=f"1 of the clip's 25 sampler steps = {len(passes)} UNet forward(s) (fp32 oracle, b=1): "
                       + ", ".join(f"{what} ctx {c.shape[1]} = {sec:.2f} s" for (_, c, what), sec in zip(passes, secs))
                       + "; value = 16 frames / (25 x their sum)",
                seconds_per_cfg_step=sum(secs)), parity


def timed_clips(sample_fn, steps, warmup, dist=None, sync=None, after=None, lanes=1, lane_ctx=None):
    """W untimed + exactly K timed calls of `sample_fn(i)`, bracketed by sync + barrier + sync on both sides;
    `after(last_output)` (the final all_gather) runs inside the timed region.  Returns (max-over-ranks wall seconds,
    this rank's wall seconds, last output, after's result).  Device independent so the multi-process logic is testable
    on CPU with gloo.

    lanes > 1: that many clips are IN FLIGHT on this rank -- one host thread per lane, each inside `lane_ctx(l)` (on the
    GPU: its own HIP stream, hence its own hipGraphs and static buffers); the lanes pull the K timed clip indices from one
    shared counter, so exactly K clips are sampled whatever K is.  Every lane first runs the W warm-up clips itself (its
    graphs are captured there), one lane after the other."""
    sync = sync or (lambda: None)
    lane_ctx = lane_ctx or (lambda l: contextlib.nullcontext())

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    outs = {}
    if lanes <= 1:
        for i in range(warmup):
            sample_fn(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            outs[i] = sample_fn(warmup + i)
    else:
        for l in range(lanes):
            with lane_ctx(l):
                for i in range(warmup):
                    sample_fn(i)
        fence()
        lock, nxt, errors = threading.Lock(), [0], []

        def lane(l):
            try:
                with lane_ctx(l):
                    while True:
                        with lock:
                            i = nxt[0]
                            nxt[0] += 1
                        if i >= steps:
                            break
                        outs[i] = sample_fn(warmup + i)
                    sync()
            except BaseException as e:   # re-raised on the main thread
                errors.append(e)

        threads = [threading.Thread(target=lane, args=(l,)) for l in range(lanes)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    out = outs.get(steps - 1)
    sync()
    own = time.perf_counter() - t0       # this rank's own clips, before any rank waits for another
    extra = after(out) if after is not None else None
    fence()
    mine = time.perf_counter() - t0
    worst = mine
    if dist is not None:
        tt = torch.tensor([mine, own], dtype=torch.float64)
        if torch.cuda.is_available() and dist.get_backend() == "nccl":
            tt = tt.cuda()
        every = [torch.empty_like(tt) for _ in range(dist.get_world_size())]
        dist.all_gather(every, tt)                       # every rank's own wall time (the line reports them all)
        PER_RANK_S[:] = [float(t[1].item()) for t in every]
        worst = max(float(t[0].item()) for t in every)
    else:
        PER_RANK_S[:] = [own]
    return worst, mine, out, extra


PER_RANK_S = []      # wall seconds every rank spent on its OWN clips of the timed region, before the closing all_gather / barrier


def gather_latents(dist, world):
    """after-hook of the timed region: all_gather of the final latents (128 KB per rank); returns the number of ranks whose
    latents arrived finite and distinct."""
    def run(out):
        if dist is None:
            return 1
        flat = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)     # one buffer, one collective
        parts = flat.view((world,) + tuple(out.shape))
        if dist.get_backend() == "nccl":                 # the form is decided from the backend, never by catching a failed collective
            dist.all_gather_into_tensor(flat, out.contiguous())
        else:
            dist.all_gather(list(parts.unbind(0)), out.contiguous())
        sums = [float(p.double().abs().sum()) for p in parts]
        return len({round(s, 3) for s in sums if s == s and s != float("inf")})
    return run


# ---- the dominant kernel, timed live ---------------------------------------------------------------------------------
def _block_fractions(bits, wbits, L):
    """(fraction of (64-query group, 32-key block) pairs with a visible key = what the per-wave kernel visits, both halves each;
    fraction of (32-query patch, 32-key block) pairs with a visible key = the half-blocks the workgroup-shared kernel multiplies)."""
    nblk = (L + 31) // 32
    words = wbits.reshape(-1).to(torch.int64) & 0xFFFFFFFF
    pop = sum(int(((words >> i) & 1).sum()) for i in range(32))
    visited64 = pop / (wbits.shape[0] * wbits.shape[1] * nblk)
    nb = bits.shape[0]
    half = (bits.reshape(nb, L // 32, 32, bits.shape[-1]) != 0).any(dim=2)[..., :nblk]
    return visited64, float(half.float().mean())


def dominant_kernel(model, device, inputs, reps=20):
    """The masked epipolar attention (largest share of kernel time in the rocprofv3 trace of this command; round 4: attn_shared_kernel<4,4>,
    K / V blocks shared by a workgroup): the 5 + 5 temporal blocks at 32x32 and 16x16 latents of one CFG step (b = 2: cond + uncond), on
    the benchmark clip's own masks, timed with HIP events on the launch stream: once in isolation (a hipGraph of those 10
    launches, warm caches: `us_per_launch_isolated`) and once inside eager CFG steps of the model (`us_per_launch`, the
    figure the roofline uses and the one the rocprofv3 trace of this command agrees with).
    Algorithmic FLOPs per launch by SURVEY.md section 8(d)'s dense convention (4 Lq Lk 64 H b, mask ignored), and the
    executed share ((32-query, 32-key) half-blocks the kernel multiplies / all) beside it."""
    from camc2v_amd import ops
    if not WORKLOAD["camera"]:
        return None
    T = WORKLOAD["T"]
    cam = inputs[0]["camera_condition"]["sample_locs_packed"]
    g = torch.Generator(device=device).manual_seed(SEED)
    shared = os.environ.get("CCV_ATTN_SHARED", "4") in ("4", "8")
    calls, flops, visited, executed = [], [], [], []
    for d, hl, H in ((8, 32, 5), (16, 16, 10)):
        bits, flags, perm, wbits, order = cam[d]
        L = T * hl * hl
        qkv = torch.randn(2 * L, 3 * H * 64, device=device, generator=g).to(torch.bfloat16)
        kreg = torch.randn(4, H * 64, device=device, generator=g).to(torch.bfloat16)
        s = (L * 3 * H * 64, 0, 3 * H * 64)
        out = torch.empty(2 * L, H * 64, device=device, dtype=torch.bfloat16)
        kw = dict(B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s, mask_bits=bits, tile_flags=flags, mask_nb=bits.shape[0],
                  wave_bits=wbits, group_order=order, perm=perm, kreg=kreg, vreg=kreg, out=out, o_str=(L * H * 64, 0, H * 64))
        calls.append((qkv, H, kw))
        flops.append(4.0 * L * L * 64 * H * 2)
        v64, h32 = _block_fractions(bits, wbits, L)
        visited.append(v64)
        executed.append(h32 if shared and (L + 31) // 32 <= 1037 else v64)

    def run():
        for qkv, H, kw in calls:
            for _ in range(5):
                ops.attention(qkv, qkv[:, H * 64:], qkv[:, 2 * H * 64:], **kw)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us_isolated = e0.elapsed_time(e1) * 1e3 / (reps * 10)
    # ... and where the trace sees them: inside the UNet.  Eager CFG steps of the benchmark clip (the forward pair the sampler
    # runs), HIP events on the launch stream around every sparse-attention launch (ops.SPARSE_PROBE); the first step is warm-up.
    # This is the figure the roofline uses -- K / V come fresh from the QKV projection and the other layers' traffic has passed
    # through the caches in between, as in the profiled run.
    ops.SPARSE_PROBE = probe = []
    try:
        with torch.no_grad():
            for _ in range(3):
                cfg_step(model, device, inputs)
        torch.cuda.synchronize()
    finally:
        ops.SPARSE_PROBE = None
    per_step = len(probe) // 3
    timed = probe[per_step:]
    us = sum(a.elapsed_time(b) for a, b, *_ in timed) * 1e3 / max(1, len(timed)) if timed else us_isolated
    fl = 0.5 * (flops[0] + flops[1])                  # mean over the 10 launches
    fl_exec = 0.5 * (flops[0] * executed[0] + flops[1] * executed[1])
    tf = fl / us / 1e6
    kernel = "attn_shared_kernel<4,4>" if shared else "attn_sparse_kernel"
    return dict(kernel=kernel, us_per_launch=us, us_per_launch_isolated=us_isolated, launches_timed_in_model=len(timed),
                flops_per_launch=fl, achieved_dense=tf, frac_dense=tf / PEAK_BF16_TFLOPS,
                executed_fraction=fl_exec / fl, effective_tflops=fl_exec / us / 1e6,
                problem=f"masked epipolar attention, b=2 (cond+uncond), L={T * 1024} H=5 (x5) and L={T * 256} H=10 (x5) per CFG step, 4 register "
                        "tokens, benchmark masks; dense FLOP convention 4 Lq Lk 64 H b",
                visited_block_fraction={"32x32": visited[0], "16x16": visited[1]},
                executed_halfblock_fraction={"32x32": executed[0], "16x16": executed[1]})


def gemm_family(model, device, inputs, steps=2):
    """The GEMM family (ccv_gemm: every linear layer, 3x3 / temporal convolution and stacked projection; ~64 % of the kernel time of
    a clip) timed live: HIP events on the launch stream around every ccv_gemm call (split-K reduce launches included) of eager CFG
    steps of the benchmark clip (`ops.GEMM_PROBE`; one warm-up step, `steps` timed).  FLOPs = sum of 2 M N K taps of the calls,
    i.e. executed work (context K/V projections are cached per clip and do not appear)."""
    from camc2v_amd import ops
    with torch.no_grad():
        cfg_step(model, device, inputs)      # warm-up (caches, allocator)
        torch.cuda.synchronize()
        ops.GEMM_PROBE = probe = []
        try:
            for _ in range(steps):
                cfg_step(model, device, inputs)
            torch.cuda.synchronize()
        finally:
            ops.GEMM_PROBE = None
    ms = sum(a.elapsed_time(b) for a, b, _ in probe) / steps
    tf = sum(f for _, _, f in probe) / steps / 1e12
    return dict(calls_per_cfg_step=len(probe) // steps, ms_per_cfg_step=ms, executed_tflop_per_cfg_step=tf, achieved=tf / ms * 1e3,
                frac=tf / ms * 1e3 / PEAK_BF16_TFLOPS, unit="TFLOP/s",
                note="every ccv_gemm call of one eager sampler step (b = 2 under CFG), HIP events on the launch stream; the events' own cost "
                     "(~1 us per call) is inside the figure")


def two_clips_per_forward(model, device, use_graph, rank=0, calls=2):
    """Supplementary figure (NOT the metric's value): throughput when one sampling call carries TWO clips (UNet batch 4 under CFG),
    the reference's own default eval batch size (configs/models/camcontexti2v_256.yaml data.params.batch_size: 2).  The 8x8 / 4x4
    latent layers (M = 2048 / 512 rows per clip pair) are latency-bound at one clip; a second clip rides along almost free there."""
    sets = [synthetic_inputs(model, device, b=2, rank=rank, clip=100 + i) for i in range(calls + 1)]
    sample_clip(model, *sets[0], use_graph)          # warm-up: captures the graphs of this signature
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls):
        out = sample_clip(model, *sets[1 + i], use_graph)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
    assert torch.isfinite(out).all()
    return {"frames_per_s": 2.0 * WORKLOAD["T"] / dt, "ms_per_call": 1e3 * dt, "clips_per_call": 2,
            "note": "two independent clips per sampling call (own conditioning each); not the metric's configuration"}


def skipped_flops(inputs):
    """TFLOP per clip that the reference's dense count contains and this path does not execute: the masked-out (32-query, 32-key)
    half-blocks of the epipolar attention (counted from the clip's own masks) and the per-step, per-frame context K/V projections
    (done once per clip here)."""
    w = WORKLOAD
    T = w["T"]
    epi = 0.0
    if w["camera"]:
        cam = inputs[0]["camera_condition"]["sample_locs_packed"]
        shared = os.environ.get("CCV_ATTN_SHARED", "4") in ("4", "8")
        for d, hl, H, nblocks in ((8, 32, 5, 5), (16, 16, 10, 5), (32, 8, 20, 5), (64, 4, 20, 1)):
            L = T * hl * hl
            v64, h32 = _block_fractions(cam[d][0], cam[d][3], L)
            frac = h32 if shared and (L + 31) // 32 <= 1037 else v64
            if hl <= 8:
                frac = 1.0            # small maps run the tiled masked kernel (128x64 tile skipping only): counted as dense
            epi += nblocks * 4.0 * L * L * 64 * H * (1.0 - frac)
    passes = 2 if w["uncond_ctx"] else 1
    epi_clip = epi * passes * 25 / 1e12
    sum_c = 5 * 320 + 5 * 640 + 6 * 1280                # the 16 cross-attention layers' widths
    ctxs = [w["cond_ctx"]] + ([w["uncond_ctx"]] if w["uncond_ctx"] else [])
    per_frame_tokens = lambda L: (77 + 16) if L == 77 + 16 * T else L      # reference context rule (openaimodel3d.py:575)
    kv_ref = 25 * sum(4.0 * T * per_frame_tokens(L) * 1024 * sum_c for L in ctxs) / 1e12
    kv_here = sum(4.0 * L * 1024 * sum_c for L in ctxs) / 1e12
    return epi_clip, kv_ref - kv_here


def result_line(elapsed, steps, warmup, world, use_graph, dev_ms=None, dom=None, skipped=None, ranks_seen=None, lanes=1, sharded=False, gemm=None):
    w = WORKLOAD
    clips = steps if sharded else steps * world          # frame-sharded: the ranks sample each clip TOGETHER
    tf_per_clip = workload_tflop_per_clip(w)
    per_clip_s = (dev_ms / 1e3 / steps) if dev_ms is not None else elapsed / steps
    achieved = tf_per_clip / per_clip_s  # one GPU's rate: algorithmic TFLOP of a clip / its device time
    whole = {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
             "algorithmic_tflop_per_clip": tf_per_clip, "device_ms_per_clip": 1e3 * per_clip_s,
             "note": "all launches of 25 CFG steps; reference's dense FLOP count (masks ignored, context K/V per step and frame)"}
    if skipped is not None:
        epi, kv = skipped
        eff = (tf_per_clip - epi - kv) / per_clip_s
        whole.update(executed_tflop_per_clip=tf_per_clip - epi - kv, effective_achieved=eff, effective_frac=eff / PEAK_BF16_TFLOPS,
                     skipped={"epipolar_masked_blocks_tflop": epi, "context_kv_reprojection_tflop": kv,
                              "fraction_of_dense": (epi + kv) / tf_per_clip})
    roof = {"bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "achieved": None, "frac": None, "traffic": None}
    if dom is not None:
        # achieved / frac: what the hardware does -- FLOPs of the 32-key blocks the kernel actually multiplies / time; the reference's
        # dense count (mask ignored) beside it as achieved_dense / frac_dense
        roof.update(achieved=dom["effective_tflops"], frac=dom["effective_tflops"] / PEAK_BF16_TFLOPS,
                    achieved_dense=dom["achieved_dense"], frac_dense=dom["frac_dense"], kernel=dom["kernel"], us_per_launch=dom["us_per_launch"],
                    flops_per_launch=dom["flops_per_launch"], executed_fraction=dom["executed_fraction"], problem=dom["problem"],
                    visited_block_fraction=dom["visited_block_fraction"],
                    us_per_launch_isolated=dom.get("us_per_launch_isolated"), launches_timed_in_model=dom.get("launches_timed_in_model"))
    # HBM-side bytes per launch of the dominant kernel from this round's PMC passes -- collected on the metric's configuration (c1) and for the
    # masked attention kernel: the other configurations report null (no counter run of theirs is committed)
    if os.path.exists(TRAFFIC_FILE) and WORKLOAD_KEY == "c1" and dom is not None:
        try:
            tr = json.load(open(TRAFFIC_FILE))
            roof["traffic"] = tr.get("dominant_kernel_bytes_per_launch")
            roof["traffic_source"] = tr.get("source")
            whole["traffic_bytes_per_clip"] = tr.get("bytes_per_clip")
        except (OSError, ValueError):
            pass
    if gemm is not None:
        roof["gemm_family"] = gemm
        if dom is None:      # no masked attention in this workload: the GEMM family (linear layers + implicit-GEMM convolutions) dominates
            roof.update(achieved=gemm["achieved"], frac=gemm["frac"], kernel="ccv_gemm family (gemm_dma / gemm_ring / gemm_astat kernels)",
                        problem="every ccv_gemm launch of one sampler step, executed FLOPs / their summed device time")
    roof["whole_path"] = whole
    line = {
        "metric": w["metric"],
        "value": float(w["T"]) * clips / elapsed, "unit": "frames/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": w["label"] + "; seeded N(0,0.02) weights; every clip "
                               "has its own conditioning tensors (per-clip prologue inside the timed region); every sampling call "
                               "is ONE clip (UNet batch 2 under CFG) -- clips_in_flight_per_gpu independent calls run on "
                               "their own HIP streams at a time",
                   "baseline_config": WORKLOAD_KEY,
                   "clips_per_gpu": 1, "clips_in_flight_per_gpu": lanes, "parallelism": (sharded if isinstance(sharded, str) else f"frame-shard{world}") if sharded else f"clip-dp{world}",
                   "launch": "hipGraph" if use_graph else "eager"},
        "roofline": roof,
    }
    if ranks_seen is not None:
        line["config"]["ranks_in_final_all_gather"] = ranks_seen
    if PER_RANK_S:
        line["config"]["per_rank_ms_per_step"] = [1e3 * t / steps for t in PER_RANK_S]
    return line


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: run the N-rank job as a child process (this process has not touched
    the GPU and never will) and relay its output and exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main(argv=None, hooks=None):
    """hooks (tests only): dict(device=, backend=, build_model=, synthetic_inputs=, sample_clip=, sync=, extras=False) lets the
    multi-process logic run on CPU with gloo and a stand-in clip."""
    argv = sys.argv[1:] if argv is None else list(argv)
    if os.environ.get("CCV_BENCH_WATCHDOG"):      # diagnosis: dump every thread's stack and exit after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CCV_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--same-conditioning", action="store_true", help="every clip reuses clip 0's tensors (round-1 behaviour)")
    ap.add_argument("--lanes", type=int, default=2, help="independent clips in flight per GPU (one host thread + HIP stream + hipGraph set each)")
    ap.add_argument("--frame-shard", action="store_true", help="single-clip latency mode: the N ranks sample EVERY clip together, its 16 frames "
                    "sharded over them (camc2v_amd/parallel.py; eager, strong scaling); default is one independent clip stream per rank")
    ap.add_argument("--cfg-split", action="store_true", help="single-clip latency mode for 2 ranks: rank 0 runs the conditional, rank 1 the "
                    "unconditional forward of every step (camc2v_amd/parallel.py: CfgSplit; hipGraph per rank, strong scaling)")
    ap.add_argument("--shard-graph", action="store_true", help="with --frame-shard on RCCL: capture the sharded step, collectives included, "
                    "into a hipGraph (the exchanges are issued on the compute stream); EXPERIMENTAL: has never run (no multi-GPU node was reachable)")
    ap.add_argument("--clips-only", action="store_true", help="profiling runs: no live kernel timing, no CPU baseline -- the trace then holds the clips' launches only")
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="c1", help="BASELINE.json configs: c1 (default, the metric's configuration: "
                    "CamContextI2V CFG 7.5), c0 (DynamiCrafter, CFG off), c3 (CamI2V baseline), c4 (32 frames, CFG 3.5)")
    args = ap.parse_args(argv)
    global WORKLOAD, WORKLOAD_KEY
    WORKLOAD, WORKLOAD_KEY = WORKLOADS[args.config], args.config
    hooks = hooks or {}

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return spawn_ranks(args, argv)      # before any GPU call in this process
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    on_gpu = "device" not in hooks
    # CCV_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo -- rehearses the N-rank code path on a one-GPU box (RCCL refuses two ranks
    # on one device); the numbers of such a run mean nothing
    rehearsal = on_gpu and os.environ.get("CCV_BENCH_REHEARSAL") == "1"
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
        if rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    else:
        device, sync = hooks["device"], hooks.get("sync", lambda: None)
    dist = None
    # CCV_BENCH_RCCL_1RANK=1 (launched by torch.distributed.run with one rank): the N-rank code path -- RCCL communicator on the rank's
    # device, identity check, barriers, per-rank times, the final all_gather_into_tensor -- on a one-GPU box
    if world > 1 or (on_gpu and "WORLD_SIZE" in os.environ and os.environ.get("CCV_BENCH_RCCL_1RANK") == "1"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            if on_gpu and rehearsal:
                dist.init_process_group("gloo")
            elif on_gpu:
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(hooks.get("backend", "gloo"), rank=rank, world_size=world)

    if dist is not None and on_gpu and not rehearsal:
        # every rank on a device of its own, and RCCL sees all of them, before anything is timed
        ids = [None] * world
        props = torch.cuda.get_device_properties(device)
        dist.all_gather_object(ids, (socket.gethostname(), torch.cuda.current_device(), str(getattr(props, "uuid", "")), getattr(props, "pci_bus_id", -1)))
        if len(set(ids)) != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: {world} ranks but {len(set(ids))} distinct devices: {ids}")
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        if int(probe.item()) != world:
            raise SystemExit(f"bench.py: RCCL all_reduce saw {int(probe.item())} of {world} ranks")

    torch.set_grad_enabled(False)
    model = hooks.get("build_model", build_model)(device)
    make_inputs = hooks.get("synthetic_inputs", synthetic_inputs)
    run_clip = hooks.get("sample_clip", sample_clip)
    n_sets = 1 if args.same_conditioning else args.steps + args.warmup
    if args.cfg_split and world != 2:
        raise SystemExit("bench.py: --cfg-split needs exactly 2 ranks")
    sharded = bool((args.frame_shard or args.cfg_split) and world > 1)     # the ranks sample every clip TOGETHER
    sets = [make_inputs(model, device, rank=0 if sharded else rank, clip=i) for i in range(n_sets)]     # sharded: all ranks work on the same clips
    use_graph = not args.no_graph and not (sharded and args.frame_shard and not (args.shard_graph and on_gpu and not rehearsal))
    if sharded:
        args.lanes = 1
        if on_gpu and args.frame_shard:
            model.model.diffusion_model.enable_frame_sharding()
        elif on_gpu:
            from camc2v_amd.parallel import CfgSplit
            model.cfg_split = CfgSplit()

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if on_gpu else None
    first_ms = {}

    def one_clip(i):
        if on_gpu and i == args.warmup and args.lanes <= 1:          # first timed clip: HIP event on the launch stream
            ev[0].record()
        t0 = time.perf_counter()
        out = run_clip(model, *sets[i % n_sets], use_graph)
        if i == 0 and "v" not in first_ms:
            sync()
            first_ms["v"] = 1e3 * (time.perf_counter() - t0)
        return out

    lanes = max(1, min(args.lanes, args.steps))
    lane_ctx = None
    if on_gpu and lanes > 1:
        streams = [torch.cuda.Stream(device) for _ in range(lanes)]

        @contextlib.contextmanager
        def lane_ctx(l):                       # device, grad mode and current stream are per host thread
            torch.cuda.set_device(device)
            with torch.no_grad(), torch.cuda.stream(streams[l]):
                yield
                streams[l].synchronize()
    if on_gpu:
        from camc2v_amd import ops as _ops
        _ops.set_streams_in_flight(lanes)      # planner hint, read when the graphs are captured
    elapsed, _, out, ranks_seen = timed_clips(one_clip, args.steps, args.warmup, dist, sync, gather_latents(dist, world), lanes, lane_ctx)
    dev_ms = None
    if on_gpu and lanes == 1:
        ev[1].record()
        torch.cuda.synchronize()
        dev_ms = ev[0].elapsed_time(ev[1])
    assert torch.isfinite(out).all()

    if rank == 0:
        extras = on_gpu and hooks.get("extras", True) and not args.clips_only
        dom = dominant_kernel(model, device, sets[0]) if extras else None
        skipped = skipped_flops(sets[0]) if extras else None
        gemm = gemm_family(model, device, sets[0]) if extras and not sharded else None
        line = result_line(elapsed, args.steps, args.warmup, world, use_graph, dev_ms, dom, skipped, ranks_seen, lanes,
                           ("cfg-split2" if args.cfg_split else sharded) if sharded else False, gemm)
        if first_ms:
            line["config"]["first_clip_ms"] = first_ms["v"]    # includes packing, graph capture (a new signature) and caches
        if world == 1 and extras and lanes > 1:      # the same clips one at a time (one stream), for comparison
            _ops.set_streams_in_flight(1)
            run_clip(model, *sets[0], use_graph)         # this (the default) stream has no graphs yet: capture them untimed
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(2):
                run_clip(model, *sets[(args.warmup + i) % n_sets], use_graph)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            line["config"]["one_clip_at_a_time"] = {"frames_per_s": float(WORKLOAD["T"]) / dt, "ms_per_clip": 1e3 * dt}
            line["value_one_clip_at_a_time"] = float(WORKLOAD["T"]) / dt
        if world == 1 and extras and args.config == "c1":
            line["config"]["two_clips_per_forward"] = two_clips_per_forward(model, device, use_graph, rank)
        if world == 1 and extras and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity_full_size"] = cpu_baseline(model, device, sets[0])
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        if "device" not in hooks or hooks.get("destroy", True):
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
