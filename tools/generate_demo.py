#!/usr/bin/env python
"""End-to-end walk through everything built on the HIP kernels, at the shipped sizes with seeded weights and synthetic
inputs (no checkpoints / no network here): context images -> first-stage encode -> context-frame adaptor (c_concat);
image tokens -> Resampler (c_crossattn); poses -> ray embedding -> pose encoder (Pluecker features) + epipolar masks;
25 CFG DDIM steps of the camera-conditioned UNet; first-stage decode to 16 frames of 256x256.  Prints per-stage device
time.  The OpenCLIP encoders are not part of this repository: their outputs are synthetic tensors of the right shape.
    python tools/generate_demo.py [--steps 25]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camc2v_amd import camera, configs  # noqa: E402


def seeded(module, seed, std=0.02):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.normal_(0.0, std, generator=g)
            if p.dim() == 1 and name.endswith(".weight"):
                p.add_(1.0)
    return module


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    from utils.utils import instantiate_from_config
    cfg = configs.camcontexti2v_256()
    cfg["params"].update(configs.FEEDERS_256)
    cfg["params"]["scale_factor"] = 0.18215
    torch.manual_seed(bench.SEED)
    with torch.device(dev):
        model = instantiate_from_config(cfg)
        model.build_feeders()
    seeded(model, bench.SEED)
    model.eval()
    model.model.diffusion_model.prepare()

    torch.linalg.inv(torch.eye(4, device=dev)[None])      # loads the solver library (seconds on a fresh box) outside the stage timers
    g = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    T, px, n_ctx = 16, 256, 2
    stages = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        stages[name] = round((time.perf_counter() - t0) * 1e3, 2)
        return out

    frames = rn(1 + n_ctx, 3, px, px).clamp(-1, 1)                                    # conditioning frame + context frames
    z_ctx = timed("first_stage_encode_3_images_ms", lambda: model.encode_first_stage(frames, noise=None))
    K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]], device=dev).repeat(1, T, 1, 1)
    w2c = camera.synthetic_trajectory(1, T, dev)
    cond_idx = torch.zeros(1, dtype=torch.long, device=dev)
    feats = timed("pose_encoder_ms", lambda: model.pose_features(K, w2c, cond_idx, px, px))
    cam = timed("epipolar_masks_ms", lambda: model.camera_condition(K, w2c, cond_idx, px, px, pluker_features=feats, generator=g))
    # c_concat: adaptor over the latents of the conditioning + context frames with the target x context epipolar mask,
    # Conv3d latent projection, + conditioning latent (context poses: two views further along the trajectory)
    w2c_ctx = camera.synthetic_trajectory(1, 2 * T, dev)[:, [T + 3, 2 * T - 1]]
    z_cond, z_extra = z_ctx[:1], z_ctx[1:].permute(1, 0, 2, 3)[None]
    c_concat = timed("context_concat_adaptor_ms", lambda: model.context_concat(z_cond, z_extra, K, w2c, w2c_ctx, cond_idx))
    clip_tokens = rn(1 + n_ctx, 257, 1280)                                              # stand-in for the OpenCLIP image tokens
    img_ctx = timed("resampler_ms", lambda: model.image_proj_model(clip_tokens)).reshape(1, (1 + n_ctx) * 256, 1024)
    text = rn(1, 77, 1024)
    cond = dict(c_concat=[c_concat], c_crossattn=[torch.cat([text, img_ctx], 1).contiguous()], camera_condition=cam)
    uncond = dict(c_concat=[c_concat], c_crossattn=[torch.cat([rn(1, 77, 1024), rn(1, 16 * T, 1024)], 1).contiguous()])
    fs = torch.full((1,), 8, dtype=torch.long, device=dev)
    x_T = rn(1, 4, T, 32, 32)
    noises = [rn(1, 4, T, 32, 32) for _ in range(args.steps)]
    kw = dict(configs.GENERATION_KWARGS)
    kw.pop("ddim_steps")

    def sample():
        s, _ = model.sample_log(cond, 1, True, args.steps, x_T=x_T, unconditional_conditioning=uncond, fs=fs,
                                injected_noise=noises, use_graph=True, **kw)
        return s
    timed("ddim_first_call_incl_graph_capture_ms", sample)
    z0 = timed(f"ddim_{args.steps}_cfg_steps_ms", sample)
    video = timed("first_stage_decode_16_frames_ms", lambda: model.decode_first_stage(z0))
    assert video.shape == (1, 3, T, px, px) and torch.isfinite(video).all()
    stages["frames"] = list(video.shape)
    print(json.dumps(stages), flush=True)


if __name__ == "__main__":
    main()
