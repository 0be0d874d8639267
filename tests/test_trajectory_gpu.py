"""Full 25-step DDIM trajectories of the HIP path against trajectories produced by RUNNING THE REFERENCE
(tests/golden/traj_medium.npz, written by oracle/gen_golden_traj.py: the reference's DDIMSampler.sample + UNet on the
medium-width network -- model_channels 128, 16 frames of 16x16 latents -- with the generation kwargs of
02_generate_videos.py:318-327; the sampler's N(0,1) draws are reproduced from the recorded seed and injected).

Stated tolerance over 25 steps (bf16 GEMM/attention operands against the reference's fp32):
    camera-conditioned, CFG 7.5, guidance_rescale 0.7, eta 1:  rel-L2 of x_t <= 4.2e-2 at every recorded step  (measured 2.0e-2 after step 1 ... 3.4e-2 from step 10 on)
    DynamiCrafter (no camera), CFG off, eta 1:                 rel-L2 of x_t <= 8e-3 at every recorded step   (measured 3.4e-3 ... 5.0e-3)
(single forward: 2e-2, tests/test_unet_gpu.py::test_medium_fixture_tight_tolerance).  The per-step errors are printed:
they do not grow with the step count -- the DDIM update contracts the error as the noise level falls.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TOL_CAM, TOL_DC = 4.2e-2, 8e-3       # round 4: tightened to 1.25x / 1.6x the measured 3.36e-2 / 5.0e-3 (were 5e-2 / 1.5e-2); the fp16-operand build is held to 1e-2


def _noises(seed, sums, shape):
    torch.manual_seed(int(seed))
    zs = [torch.randn(shape) for _ in range(25)]
    got = np.array([float(z.double().sum()) for z in zs])
    assert np.allclose(got, sums, rtol=0, atol=1e-6), "torch CPU RNG stream changed: the injected noise is not the reference's"
    return zs


def _report(what, xs, fx_steps, keep, tol):
    worst = 0.0
    for i, ref in zip(keep, fx_steps):
        ref = torch.from_numpy(ref)
        got = xs[int(i)].float().cpu()
        assert torch.isfinite(got).all()
        l2 = ((got - ref).norm() / ref.norm()).item()
        mx = ((got - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, l2)
        print(f"[parity] {what}: after step {int(i) + 1:2d}/25 rel_l2={l2:.3e} max_rel={mx:.3e}")
    assert worst <= tol, f"{what}: worst rel-L2 {worst:.3e} over the trajectory (tolerance {tol})"


def _medium_unet_params():
    from oracle.golden_inputs import MEDIUM_CFG
    return {"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(MEDIUM_CFG)}


def test_25_step_camera_cfg_trajectory_vs_reference(golden_dir):
    """configs[1] at reduced width, end to end through `model.camcontexti2v.CamContextI2V` + DDIMSampler."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import camera
    from oracle.golden_inputs import SEED, medium_inputs
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(golden_dir, "traj_medium.npz"))
    med = np.load(os.path.join(golden_dir, "unet_medium.npz"))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    model = instantiate_from_config({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config=_medium_unet_params(), linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4,
        image_size=[16, 16], temporal_length=16, add_type="add_to_main_branch",
        pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": {}},
        epipolar_config=dict(origin_h=128, origin_w=128, is_3d_full_attn=False, num_register_tokens=4,
                             attention_resolution=[8, 4, 2, 1], compression_factor=1))})
    model.model.diffusion_model.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    model = model.to(dev).eval()
    inp = medium_inputs()
    to = lambda t: t.to(dev)
    F = torch.from_numpy(med["F128"]).to(dev)
    cam = dict(pluker_embedding_features=[to(f) for f in inp["feats"]], sample_locs_dict=None,
               sample_locs_packed=camera.epipolar_masks_packed(F, 16, 128, 128),
               cond_frame_index=torch.zeros(1, dtype=torch.long, device=dev), add_type="add_to_main_branch")
    cc = to(inp["c_concat"])
    cond = dict(c_concat=[cc], c_crossattn=[to(inp["ctx_rep"])], camera_condition=cam)
    zs = _noises(fx["noise_seed_cam"], fx["cam_noise_checksum"], (1, 4, 16, 16, 16))
    kw = dict(eta=1.0, x_T=inp["x_T"], unconditional_guidance_scale=7.5, timestep_spacing="uniform_trailing",
              guidance_rescale=0.7, fs=to(inp["fs"]), enable_camera_condition=True, injected_noise=zs)
    uncond = dict(c_concat=[cc], c_crossattn=[to(inp["ctx_pf"])])
    samples, inter = model.sample_log(cond, 1, True, 25, unconditional_conditioning=uncond, log_every_t=1, **kw)
    xs = inter["x_inter"][1:]
    assert len(xs) == 25 and torch.equal(xs[-1], samples)
    _report("25-step camera CFG 7.5 trajectory vs REFERENCE", xs, fx["cam_x_steps"], fx["keep_steps"], TOL_CAM)
    # the hipGraph replay of the same clip gives the same latents bit for bit
    uncond = dict(c_concat=[cc], c_crossattn=[to(inp["ctx_pf"])])
    graphed, _ = model.sample_log(cond, 1, True, 25, unconditional_conditioning=uncond, use_graph=True, **kw)
    assert torch.equal(graphed, samples)


def test_25_step_dynamicrafter_cfg_off_trajectory_vs_reference(golden_dir):
    """configs[0] at reduced width: `model.dynamicrafter.DynamiCrafter` target (no camera modules in the checkpoint),
    CFG off => one UNet forward per step, per-frame image tokens (context 77 + 16 t)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.golden_inputs import SEED, medium_inputs
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(golden_dir, "traj_medium.npz"))
    model = instantiate_from_config({"target": "model.dynamicrafter.DynamiCrafter", "params": dict(
        unet_config=_medium_unet_params(), linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4,
        image_size=[16, 16], temporal_length=16)})
    unet = model.model.diffusion_model
    man = {k: list(v.shape) for k, v in unet.state_dict().items()}
    assert len(man) == int(fx["dc_num_keys"]), "checkpoint layout differs from the reference's plain UNetModel"
    unet.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    model = model.to(dev).eval()
    inp = medium_inputs()
    to = lambda t: t.to(dev)
    cond = dict(c_concat=[to(inp["c_concat"])], c_crossattn=[to(inp["ctx_pf"])])
    zs = _noises(fx["noise_seed_dc"], fx["dc_noise_checksum"], (1, 4, 16, 16, 16))
    kw = dict(eta=1.0, x_T=inp["x_T"], unconditional_guidance_scale=1.0, unconditional_conditioning=None,
              timestep_spacing="uniform_trailing", guidance_rescale=0.0, fs=to(inp["fs"]), injected_noise=zs)
    samples, inter = model.sample_log(cond, 1, True, 25, log_every_t=1, **kw)
    xs = inter["x_inter"][1:]
    assert len(xs) == 25 and torch.equal(xs[-1], samples)
    _report("25-step DynamiCrafter CFG-off trajectory vs REFERENCE", xs, fx["dc_x_steps"], fx["keep_steps"], TOL_DC)
    graphed, _ = model.sample_log(cond, 1, True, 25, use_graph=True, **kw)
    assert torch.equal(graphed, samples)
