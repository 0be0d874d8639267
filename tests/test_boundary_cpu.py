"""Host side of the drop-in boundary that needs no GPU: the reference's shipped configs instantiate through the plugin
mechanism (fixtures: the `model:` sections of its yaml files, tests/golden/model_configs.json), checkpoints in the three
containers the reference accepts load with the legacy rename and an explicit report, the per-sample writers produce the layout
03_evaluation.py reads, and the harness's dataset stand-in has the reference dataset's batch keys."""
import json
import os

import numpy as np
import pytest
import torch

from utils.utils import instantiate_from_config


@pytest.fixture(scope="module")
def configs(golden_dir):
    return json.load(open(os.path.join(golden_dir, "model_configs.json")))


@pytest.mark.parametrize("name,cls,n_unet,feeders", [
    ("camcontexti2v_256", "CamContextI2V", 1660, {"first_stage_model", "image_proj_model", "pose_encoder", "multi_cond_latent_adaptor"}),
    ("cami2v_256", "CamI2V", 1660, {"first_stage_model", "image_proj_model", "pose_encoder"}),
    ("dynamicrafter_256", "DynamiCrafter", 1532, {"first_stage_model", "image_proj_model"})])
def test_reference_model_configs_instantiate(configs, golden_dir, name, cls, n_unet, feeders):
    with torch.device("meta"):          # 1.5 B parameters: shapes and names only
        model = instantiate_from_config(configs[name]["model"])
        made = set(model.build_feeders())
    assert type(model).__name__ == cls and made == feeders
    sd = model.state_dict()
    unet = {k[len("model.diffusion_model."):]: list(v.shape) for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    assert len(unet) == n_unet
    if n_unet == 1660:                  # the camera-conditioned UNet: key for key the reference's manifest
        assert unet == json.load(open(os.path.join(golden_dir, "unet_full_manifest.json")))
    for attr in ("first_stage_key", "cond_stage_key", "uncond_type", "fps_condition_type"):
        assert hasattr(model, attr)
    assert callable(model.log_images) and callable(model.get_batch_input) and callable(model.apply_model)
    kw = configs[name].get("log_images_kwargs")
    if kw:
        assert kw["ddim_steps"] == 25 and kw["timestep_spacing"] == "uniform_trailing"


def _small_model():
    from oracle.golden_inputs import SMALL_CFG
    torch.manual_seed(0)
    return instantiate_from_config({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(SMALL_CFG)},
        linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4, image_size=[8, 8], temporal_length=16,
        use_zero_conv_latent_input=True, multi_cond_strategy="token_concat_latent_epipolar", first_stage_key="video",
        pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": {}},
        epipolar_config=dict(origin_h=64, origin_w=64, is_3d_full_attn=False, num_register_tokens=4, attention_resolution=[8, 4, 2, 1]))})


@pytest.mark.parametrize("container", ["state_dict", "module", "bare"])
def test_load_checkpoints_three_containers_and_legacy_names(tmp_path, container):
    """main/utils_train.py:165-214: Lightning {'state_dict'}, DeepSpeed {'module'}, bare dict; framestride_embed -> fps_embedding;
    tensors of modules this package does not build (OpenCLIP encoders, logvar) end up in the report, not in an exception."""
    from main.utils_train import load_checkpoints
    from camc2v_amd.checkpoint import extract_state_dict
    src = _small_model()
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v.clone() for k, v in src.state_dict().items()}
    ck = {}
    for k, v in sd.items():
        ck[k.replace("fps_embedding", "framestride_embed")] = v
    assert any("framestride_embed" in k for k in ck)
    ck["cond_stage_model.model.ln_final.weight"] = torch.ones(4)
    ck["embedder.model.visual.proj"] = torch.ones(4, 4)
    ck["logvar"] = torch.zeros(1000)
    obj = {"state_dict": ck, "epoch": 3} if container == "state_dict" else {"module": ck, "dp_world_size": 8} if container == "module" else ck
    path = tmp_path / "ckpt.pt"
    torch.save(obj, path)
    assert extract_state_dict(torch.load(path, weights_only=True))[0] == container
    model, report = _small_model(), []
    out = load_checkpoints(model, {"pretrained_checkpoint": str(path)}, report)
    assert out is model
    rep = report[0]
    assert rep["container"] == container and not rep["strict"] and rep["missing"] == []
    assert len(rep["renamed"]) == sum("fps_embedding" in k for k in sd)
    assert dict(rep["ignored_prefixes"]) == {"cond_stage_model": 1, "embedder": 1, "logvar": 1}
    got = model.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # attribute-style config (OmegaConf in the reference) and the no-checkpoint case
    class Cfg:
        pretrained_checkpoint = str(path)
    load_checkpoints(_small_model(), Cfg())
    assert load_checkpoints(model, {}) is model
    # a checkpoint that lacks tensors of the model is reported by name
    del ck[next(k for k in ck if k.endswith("time_embed.0.weight"))]
    torch.save(obj, path)
    report = []
    load_checkpoints(_small_model(), {"pretrained_checkpoint": str(path)}, report)
    assert report[0]["missing"] == ["model.diffusion_model.time_embed.0.weight"]
    # a UNet-only strict load through the same loader
    unet_sd = {k[len("model.diffusion_model."):]: v for k, v in sd.items() if k.startswith("model.diffusion_model.")}
    from camc2v_amd.checkpoint import load_state_dict_with_report
    assert load_state_dict_with_report(_small_model().model.diffusion_model, unet_sd)["strict"]


def test_video_writers_round_trip_and_sample_layout(tmp_path):
    from utils.save_video import log_evaluation, prepare_to_log
    from camc2v_amd.video_io import read_mjpeg_mp4, write_video
    T, H, W = 5, 48, 64
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    frames = np.stack([np.stack([(255 * (0.5 + 0.5 * np.sin(6 * xx + 0.4 * t + c))).astype(np.uint8) for c in range(3)], -1) for t in range(T)])
    codec = write_video(tmp_path / "a.mp4", torch.from_numpy(frames), fps=7)
    if codec == "mjpeg":
        back, fps = read_mjpeg_mp4(tmp_path / "a.mp4")
        assert back.shape == frames.shape and fps == pytest.approx(7.0)
        assert np.abs(back.astype(int) - frames.astype(int)).mean() < 2.0          # JPEG q=95
        raw = open(tmp_path / "a.mp4", "rb").read()
        assert raw[4:8] == b"ftyp" and b"moov" in raw and b"mp4v" in raw and b"stco" in raw
    b, n = 2, 2
    logs = dict(samples=torch.rand(b, 3, T, H, W) * 2 - 1, gt_video=torch.rand(b, 3, T, H, W) * 2 - 1,
                image_condition=torch.rand(b, 3, 1, H, W) * 2 - 1, cond_frames=torch.rand(b, n, 3, H, W) * 2 - 1,
                camera_data=torch.rand(b, T, 19), condition=["clip a_fs=8", "clip b_fs=8"], video_path=["x/clip_a.mp4", "x/clip_b.mp4"],
                depth_scale=torch.ones(b), reconst=torch.rand(b, 3, T, H, W))
    dirs = log_evaluation(prepare_to_log(logs, -1), tmp_path / "test", save_fps=7, rescale=True)
    assert [os.path.basename(d) for d in dirs] == ["clip_a", "clip_b"]
    for d in dirs:
        assert sorted(os.listdir(d)) == ["camera_data.npy", "captions.txt", "context_0.png", "context_1.png", "generated.mp4", "ground_truth.mp4"]
        assert np.load(os.path.join(d, "camera_data.npy")).shape == (T, 19)
        assert open(os.path.join(d, "captions.txt")).read().splitlines() == ["clip a_fs=8", "clip b_fs=8"]
        from PIL import Image
        assert Image.open(os.path.join(d, "context_0.png")).size == (W, H)


def test_synthetic_dataset_has_the_reference_batch_keys():
    from camc2v_amd.data import SyntheticRealEstate, collate
    ds = SyntheticRealEstate(num_samples=3, resolution=[64, 64], num_additional_cond_frames=[1, 4], frame_stride=8, exclude_samples=["synthetic_00001"])
    assert len(ds) == 2
    b = collate([ds[0], ds[1]])
    # data/realestate10k.py:294-307
    assert {"video", "caption", "video_path", "fps", "frame_stride", "RT", "camera_data", "camera_intrinsics", "cond_frames", "RT_cond"} <= set(b)
    assert b["video"].shape == (2, 3, 16, 64, 64) and b["video"].abs().max() <= 1.0
    assert b["cond_frames"].shape == (2, 4, 3, 64, 64) and b["RT_cond"].shape == (2, 4, 4, 4) and b["camera_data"].shape == (2, 16, 19)
    assert torch.allclose(b["camera_data"][0, :, 7:].reshape(16, 3, 4), b["RT"][0, :, :3])
    assert b["video_path"] == ["synthetic_00000.mp4", "synthetic_00002.mp4"]


def test_get_batch_input_refuses_cpu_and_missing_encoders():
    from camc2v_amd.data import SyntheticRealEstate, collate
    from camc2v_amd.lib import CcvError
    model = _small_model()
    batch = collate([SyntheticRealEstate(num_samples=1, resolution=[64, 64])[0]])
    with pytest.raises((CcvError, RuntimeError)):      # no first-stage model built / CPU tensors: loud, no silent fallback
        model.get_batch_input(batch, random_uncond=False)
    with pytest.raises(NotImplementedError):
        model.get_batch_input(batch, random_uncond=True)
    with pytest.raises(CcvError):
        model.get_learned_conditioning(["a prompt"])


def test_trajectory_extension_as_the_reference_writes_it():
    """main/runtime.py:193-202: a 16-pose trajectory asked for 48 frames is extended by last @ (last^-1 @ c2ws) blocks."""
    from main.runtime import extend_trajectory
    from camc2v_amd.data import SyntheticRealEstate
    c2w = torch.linalg.inv(SyntheticRealEstate(num_samples=1, resolution=[64, 64])[0]["RT"])
    ext = extend_trajectory(c2w, 48)
    assert ext.shape[0] >= 48 and torch.equal(ext[:16], c2w)
    last = c2w[-1]
    assert torch.allclose(ext[16:32], torch.einsum("ik,tkj->tij", last, torch.einsum("ik,tkj->tij", torch.linalg.inv(last), c2w)), atol=1e-5)
    # the second replay starts from the end of the first: relative motion of frame j in replay 2 == in the original
    rel0 = torch.linalg.inv(c2w[0]) @ c2w[5]
    rel2 = torch.linalg.inv(ext[32]) @ ext[37]
    assert torch.allclose(rel0, rel2, atol=1e-4)
    assert extend_trajectory(c2w, 10) is c2w
