#!/usr/bin/env python
"""Generate tests/golden/traj_medium.npz by RUNNING THE REFERENCE sampler + UNet for full 25-step trajectories
(build container only).  TEST INFRASTRUCTURE.

Two trajectories of the medium-width network (model_channels 128, 16x16 latents, 16 frames; same seeded weights and
inputs as tests/golden/unet_medium.npz), both through the reference's DDIMSampler.sample with the generation kwargs of
02_generate_videos.py:318-327 (25 steps, eta 1, uniform_trailing):

  cam   CamContextI2V-style: camera-patched UNet, conditional context 77+768 tokens, unconditional 77+256 (per frame),
        CFG 7.5, guidance_rescale 0.7                                     (BASELINE.json configs[1] at reduced width)
  dc    DynamiCrafter-style: plain UNetModel.forward (no camera patch, no camera modules in the checkpoint), CFG off
        (scale 1.0 => one forward per step), per-frame context            (BASELINE.json configs[0] at reduced width)

The N(0,1) draws of the sampler are reproduced from the seed by the tests (checksums stored); x after the steps
listed in ``keep_steps`` and the final sample are stored as fp32.  Fixtures hold tensors and scalars only.

Usage:  python oracle/gen_golden_traj.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (shims + reference model builders; imports nothing from the repo root)

KEEP = (0, 1, 2, 4, 9, 14, 19, 24)
NOISE_SEED_CAM, NOISE_SEED_DC = 4343, 4344


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    gg._install_shims()
    torch.set_grad_enabled(False)
    repo_root = os.path.dirname(HERE)
    assert all(os.path.abspath(p or os.getcwd()) != repo_root for p in sys.path), "repo root on sys.path would shadow the reference"
    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.utils_diffusion import make_beta_schedule
    assert sys.modules["lvdm.models.samplers.ddim"].__file__.startswith(gg.REF)

    class CpuSampler(DDIMSampler):
        def register_buffer(self, name, attr):  # reference hard-codes cuda (ddim.py:18-22)
            setattr(self, name, attr)

    betas_np = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas_np, axis=0)
    to32 = lambda a: torch.tensor(a, dtype=torch.float32)

    class DuckModel:
        num_timesteps = 1000
        device = torch.device("cpu")
        use_dynamic_rescale = False
        parameterization = "eps"
        betas = to32(betas_np)
        alphas_cumprod = to32(ac)
        alphas_cumprod_prev = to32(np.append(1.0, ac[:-1]))

        def __init__(self, fn):
            self.fn = fn

        def apply_model(self, x, t, c, **kw):
            return self.fn(x, t, c, **kw)

    seeded_state_dict = gg._load_sibling("unet_oracle").seeded_state_dict
    geo = gg._load_sibling("geometry_oracle")
    inp = gg.medium_inputs()
    T, hl = 16, 16
    shape = (1, 4, T, hl, hl)
    out = dict(keep_steps=np.array(KEEP), noise_seed_cam=np.array(NOISE_SEED_CAM), noise_seed_dc=np.array(NOISE_SEED_DC))

    # ---- camera-conditioned CFG trajectory ---------------------------------------------------------------------------
    unet = gg.build_reference_unet(gg.MEDIUM_CFG, camera=True, origin=128)
    unet.load_state_dict(seeded_state_dict(gg.manifest_of(unet), gg.SEED), strict=True)
    w2c = geo.synthetic_trajectory(1, T)
    g = torch.Generator().manual_seed(gg.SEED)
    pnoise = torch.randn(1, T, T, 3, 1, generator=g)
    K128 = torch.tensor([[64.0, 0, 64], [0, 64.0, 64], [0, 0, 1.0]]).repeat(1, T, 1, 1)
    _, F128, masks = gg.geometry_via_reference(K128, w2c, torch.zeros(1, dtype=torch.long), 128, 128, pnoise)
    fx = np.load(os.path.join(args.out, "unet_medium.npz"))
    assert np.array_equal(F128.numpy(), fx["F128"]), "geometry differs from the committed medium fixture"
    cam = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks,
               cond_frame_index=torch.zeros(1, dtype=torch.long), add_type="add_to_main_branch")

    def apply(x_, t_, c_, **kw):
        return unet(torch.cat([x_, c_["c_concat"][0]], 1), t_, context=c_["c_crossattn"][0], fs=kw.get("fs"),
                    camera_condition=c_.get("camera_condition"))

    s = CpuSampler(DuckModel(apply))
    cond = dict(c_concat=[inp["c_concat"]], c_crossattn=[inp["ctx_rep"]], camera_condition=cam)
    uncond = dict(c_concat=[inp["c_concat"]], c_crossattn=[inp["ctx_pf"]])
    torch.manual_seed(NOISE_SEED_CAM)
    samples, inter = s.sample(25, 1, shape[1:], cond, eta=1.0, x_T=inp["x_T"], verbose=False, log_every_t=1,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uncond,
                              timestep_spacing="uniform_trailing", guidance_rescale=0.7, fs=inp["fs"],
                              enable_camera_condition=True)
    torch.manual_seed(NOISE_SEED_CAM)
    noises = [torch.randn(shape) for _ in range(25)]
    xs = inter["x_inter"][1:]          # [0] is x_T
    assert len(xs) == 25 and torch.equal(xs[-1], samples)
    out["cam_x_steps"] = torch.stack([xs[i] for i in KEEP]).numpy()
    out["cam_noise_checksum"] = np.array([gg.checksum(n) for n in noises])
    print("cam trajectory: final absmax", float(samples.abs().max()), "std", float(samples.std()))
    del unet

    # ---- DynamiCrafter-style trajectory: plain forward, CFG off ----------------------------------------------------------
    unet = gg.build_reference_unet(gg.MEDIUM_CFG, camera=False)
    man = gg.manifest_of(unet)
    unet.load_state_dict(seeded_state_dict(man, gg.SEED), strict=True)
    out["dc_num_keys"] = np.array(len(man))

    def apply_dc(x_, t_, c_, **kw):
        return unet(torch.cat([x_, c_["c_concat"][0]], 1), t_, context=c_["c_crossattn"][0], fs=kw.get("fs"))

    s = CpuSampler(DuckModel(apply_dc))
    cond = dict(c_concat=[inp["c_concat"]], c_crossattn=[inp["ctx_pf"]])
    torch.manual_seed(NOISE_SEED_DC)
    samples, inter = s.sample(25, 1, shape[1:], cond, eta=1.0, x_T=inp["x_T"], verbose=False, log_every_t=1,
                              unconditional_guidance_scale=1.0, unconditional_conditioning=None,
                              timestep_spacing="uniform_trailing", guidance_rescale=0.0, fs=inp["fs"])
    torch.manual_seed(NOISE_SEED_DC)
    noises = [torch.randn(shape) for _ in range(25)]
    xs = inter["x_inter"][1:]
    assert len(xs) == 25 and torch.equal(xs[-1], samples)
    out["dc_x_steps"] = torch.stack([xs[i] for i in KEEP]).numpy()
    out["dc_noise_checksum"] = np.array([gg.checksum(n) for n in noises])
    print("dc trajectory: final absmax", float(samples.abs().max()), "std", float(samples.std()))
    path = os.path.join(args.out, "traj_medium.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
