"""CPU-side host-logic tests: checkpoint layout, plugin mechanism, schedule, error behaviour."""
import json
import os

import numpy as np
import pytest
import torch

torch.set_grad_enabled(False)

EPI = dict(origin_h=256, origin_w=256, is_3d_full_attn=False, num_register_tokens=4,
           attention_resolution=[8, 4, 2, 1], compression_factor=1)


def test_full_unet_state_dict_matches_reference_manifest(golden_dir):
    """1660 tensors / 1 500 881 876 parameters, same names and shapes as the reference UNet incl. the camera
    modules (built on the meta device: nothing is allocated)."""
    from oracle.golden_inputs import FULL_CFG
    from utils.utils import instantiate_from_config
    man = json.load(open(os.path.join(golden_dir, "unet_full_manifest.json")))
    with torch.device("meta"):
        unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": FULL_CFG})
        unet.enable_camera_conditioning(EPI)
    sd = {k: list(v.shape) for k, v in unet.state_dict().items()}
    assert sd == man
    assert unet.input_ds == [1, 1, 1, 1, 2, 2, 2, 4, 4, 4, 8, 8] and unet.output_ds == [8, 8, 8, 4, 4, 4, 2, 2, 2, 1, 1, 1]
    assert len(list(unet.camera_blocks())) == 16
    # attributes the reference wrappers probe (model/camcontexti2v.py:125-143, :296)
    assert unet.init_attn[0].proj_in.out_channels == 512 and unet.temporal_length == 16
    blk = next(iter(unet.camera_blocks()))[1]
    assert blk.context_dim is None and blk.attn1.to_k.in_features == 320 and blk.attn1.heads == 5
    assert type(unet.input_blocks[1]).__name__ == "TimestepEmbedSequential"


def test_small_unet_loads_reference_layout_strict(golden_dir):
    from oracle.golden_inputs import SEED, SMALL_CFG
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config
    man = json.load(open(os.path.join(golden_dir, "unet_small_manifest.json")))
    unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG})
    unet.enable_camera_conditioning(dict(EPI, origin_h=64, origin_w=64))
    missing, unexpected = unet.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    assert not missing and not unexpected


def test_zero_init_matches_reference_convention():
    from oracle.golden_inputs import SMALL_CFG
    from camc2v_amd.unet import UNetModel
    unet = UNetModel(**SMALL_CFG).enable_camera_conditioning(dict(EPI, origin_h=64, origin_w=64))
    sd = unet.state_dict()
    for key in ("out.2.weight", "input_blocks.1.0.out_layers.3.weight", "input_blocks.1.1.proj_out.weight",
                "input_blocks.1.0.temopral_conv.conv4.3.weight", "fps_embedding.2.weight",
                "input_blocks.1.2.transformer_blocks.0.pluker_projection.weight",
                "input_blocks.1.2.transformer_blocks.0.epipolar.epipolar_attn.to_out.0.weight"):
        assert float(sd[key].abs().max()) == 0.0, key


def test_cpu_forward_fails_loudly():
    from oracle.golden_inputs import SMALL_CFG
    from camc2v_amd.lib import CcvError
    from camc2v_amd.unet import UNetModel
    unet = UNetModel(**SMALL_CFG)
    with pytest.raises(CcvError):
        unet(torch.zeros(1, 8, 16, 8, 8), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 333, 1024))


def test_sampler_schedule_matches_reference_fixture(golden_dir):
    from camc2v_amd.sampler import DDIMSampler, make_beta_schedule, make_ddim_timesteps
    fx = np.load(os.path.join(golden_dir, "ddim.npz"))
    betas = make_beta_schedule("linear", 1000, 0.00085, 0.012)
    ac = np.cumprod(1.0 - betas)

    class Duck:
        num_timesteps = 1000
        device = torch.device("cpu")
        betas = torch.tensor(make_beta_schedule("linear", 1000, 0.00085, 0.012), dtype=torch.float32)
        alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)

    for eta in (0.0, 1.0):
        s = DDIMSampler(Duck())
        s.make_schedule(25, "uniform_trailing", eta, verbose=False)
        tag = f"eta{int(eta)}"
        assert np.array_equal(s.ddim_timesteps, fx[f"timesteps_{tag}"])
        coef = s.ddim_coef.numpy()
        np.testing.assert_allclose(coef[:, 0], fx[f"ddim_alphas_{tag}"], rtol=1e-6)
        np.testing.assert_allclose(coef[:, 1], fx[f"ddim_alphas_prev_{tag}"], rtol=1e-6)
        np.testing.assert_allclose(coef[:, 2], fx[f"ddim_sigmas_{tag}"], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(coef[:, 3], fx[f"ddim_sqrt_one_minus_alphas_{tag}"], rtol=1e-6)
    assert np.array_equal(make_ddim_timesteps("uniform", 50, 1000, verbose=False), fx["timesteps_uniform50"])


def test_instantiate_from_config_contract():
    from utils.utils import instantiate_from_config
    assert instantiate_from_config("__is_first_stage__") is None
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})
    m = instantiate_from_config({"target": "torch.nn.Linear", "params": {"in_features": 3, "out_features": 2}})
    assert isinstance(m, torch.nn.Linear)


def test_first_stage_state_dict_matches_reference_manifest(golden_dir):
    """AutoencoderKL through the reference's import path and yaml-style config: same 248 keys and shapes as the
    reference's module (full-size and fixture-size configs), strict load, no CPU fallback."""
    import json
    import os

    import pytest
    import torch

    from oracle import vae_oracle as vo
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config
    for name, cfg in (("full", vo.FULL_DDCONFIG), ("small", vo.SMALL_DDCONFIG)):
        man = json.load(open(os.path.join(golden_dir, f"vae_{name}_manifest.json")))
        with torch.device("meta" if name == "full" else "cpu"):
            m = instantiate_from_config({"target": "lvdm.models.autoencoder.AutoencoderKL",
                                         "params": dict(embed_dim=4, monitor="val/rec_loss", ddconfig=dict(cfg),
                                                        lossconfig={"target": "torch.nn.Identity"})})
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
        if name == "small":
            m.load_state_dict(seeded_state_dict(man, 1), strict=True)
            from camc2v_amd.lib import CcvError
            with pytest.raises(CcvError):          # no CPU fallback
                m.encode(torch.zeros(1, 3, 64, 64))
            with pytest.raises(CcvError):
                m.decode(torch.zeros(1, 4, 8, 8))


def test_adaptor_state_dict_matches_reference_manifest(golden_dir):
    """MultiLatentEpipolarAdaptor through the reference's import path: same keys and shapes as the reference's module
    (shipped and fixture configurations); options outside the shipped configuration refuse loudly."""
    import json
    import os

    import pytest
    import torch

    from oracle import adaptor_oracle as ao
    from utils.utils import instantiate_from_config
    for name, cfg in (("full", ao.FULL_CFG), ("small", ao.SMALL_CFG)):
        man = json.load(open(os.path.join(golden_dir, f"adaptor_{name}_manifest.json")))
        with torch.device("meta" if name == "full" else "cpu"):
            m = instantiate_from_config({"target": "model.modules.adaptors.MultiLatentEpipolarAdaptor", "params": dict(cfg)})
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
    with pytest.raises(NotImplementedError):
        instantiate_from_config({"target": "model.modules.adaptors.MultiLatentEpipolarAdaptor",
                                 "params": dict(ao.SMALL_CFG, use_plucker_embedding=True)})
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):
        m(torch.zeros(1, 32, 4))


def test_resampler_state_dict_matches_reference_manifest(golden_dir):
    import json
    import os

    import pytest
    import torch

    from oracle import resampler_oracle as ro
    from utils.utils import instantiate_from_config
    for name, cfg in (("full", ro.FULL_CFG), ("small", ro.SMALL_CFG)):
        man = json.load(open(os.path.join(golden_dir, f"resampler_{name}_manifest.json")))
        with torch.device("meta" if name == "full" else "cpu"):
            m = instantiate_from_config({"target": "lvdm.modules.encoders.resampler.Resampler", "params": dict(cfg)})
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
    from camc2v_amd.lib import CcvError
    with pytest.raises(CcvError):
        m(torch.zeros(1, 9, 64))

