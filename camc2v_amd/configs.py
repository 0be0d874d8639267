"""Hyper-parameters of the shipped 256x256 models, restated from the reference's yaml
(configs/models/camcontexti2v_256.yaml, configs/baseline/cami2v_256.yaml) so that the benchmark and the
tests can build the full-size network where /root/reference is absent (the GPU box)."""

UNET_256 = dict(
    in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=1024,
    use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
    use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
    image_cross_attention=True, image_cross_attention_scale_learnable=True, default_fs=3, fs_condition=True,
)

EPIPOLAR_256 = dict(origin_h=256, origin_w=256, is_3d_full_attn=False, num_register_tokens=4,
                    attention_resolution=[8, 4, 2, 1], compression_factor=1, add_small_perturbation_on_zero_T=True)


def camcontexti2v_256(unet_params=None):
    """``model:`` section of configs/models/camcontexti2v_256.yaml restricted to what the sampling path reads."""
    return {
        "target": "model.camcontexti2v.CamContextI2V",
        "params": dict(
            parameterization="eps", linear_start=0.00085, linear_end=0.012, timesteps=1000,
            multi_cond_strategy="token_concat_latent_epipolar", conditioning_key="hybrid", image_size=[32, 32],
            channels=4, scale_factor=0.18215, uncond_type="empty_seq", use_dynamic_rescale=False,
            use_zero_conv_latent_input=True, add_type="add_to_main_branch",
            unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel",
                         "params": dict(unet_params or UNET_256)},
            pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": {}},
            epipolar_config=dict(EPIPOLAR_256),
        ),
    }


# once-per-clip feeders built on the HIP kernels (yaml first_stage_config / image_proj_stage_config /
# pose_encoder_config / multi_latent_adaptor, configs/models/camcontexti2v_256.yaml:74-151); instantiated by
# ``model.build_feeders()``
FEEDERS_256 = dict(
    first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL", "params": dict(
        embed_dim=4, monitor="val/rec_loss", lossconfig={"target": "torch.nn.Identity"},
        ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                      num_res_blocks=2, attn_resolutions=[], dropout=0.0))},
    image_proj_stage_config={"target": "lvdm.modules.encoders.resampler.Resampler", "params": dict(
        dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=1024, ff_mult=4,
        video_length=16, use_timestep_emb=True)},
    pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": dict(
        downscale_factor=8, channels=[320, 640, 1280, 1280], nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
        compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
        temporal_position_encoding=True, temporal_position_encoding_max_len=16)},
    multi_latent_adaptor={"target": "model.modules.adaptors.MultiLatentEpipolarAdaptor", "params": dict(
        query_dim=512, num_queries=1024, video_length=16, embedding_dim=4, output_dim=4, depth=12, checkpoint=True,
        timestep_embedding_type="sinusoidal_embedded", use_plucker_embedding=False)},
)

# generation kwargs wired by the reference launcher (CamContextI2V/02_generate_videos.py:318-327)
GENERATION_KWARGS = dict(ddim_steps=25, eta=1.0, unconditional_guidance_scale=7.5,
                         timestep_spacing="uniform_trailing", guidance_rescale=0.7, enable_camera_condition=True)

# algorithmic TFLOP per UNet forward at b=1, t=16, 32x32 latents
TFLOP_COND_N2 = 7.88
TFLOP_UNCOND_CAM = 7.12
TFLOP_NOCAM = 4.904
