"""Model wrappers behind the reference's ``target:`` paths (``model.camcontexti2v.CamContextI2V``,
``model.dynamicrafter.DynamiCrafter``, ``baseline.cami2v.cami2v.CamI2V``): the sampling-side slice.

They accept the yaml ``params`` of configs/models/camcontexti2v_256.yaml / configs/baseline/*.yaml,
build the MI355X UNet through ``unet_config``, add the camera modules natively (the reference
monkey-patches them in, model/camcontexti2v.py:111-170) and expose ``apply_model`` / ``sample_log`` /
``camera_condition``.  The once-per-clip feeders are not instantiated by the constructor:
their configs are kept on the object; ``build_feeders()`` instantiates the ones built here under the reference's
attribute names (``first_stage_model``, ``image_proj_model``, ``pose_encoder``, ``multi_cond_latent_adaptor``) so that
the matching checkpoint slices load.  The OpenCLIP encoders need third-party code and weights and stay inputs.
"""
import torch

from . import camera
from .config import instantiate_from_config
from .diffusion import LatentDiffusionCore

_FEEDER_KEYS = ("first_stage_config", "cond_stage_config", "img_cond_stage_config", "image_proj_stage_config",
                "pose_encoder_config", "multi_latent_adaptor", "pose_guided_cond_encoder_config")


class DynamiCrafter(LatentDiffusionCore):
    """Image-to-video base model (reference model/dynamicrafter.py, lvdm/models/ddpm3d.py:1030-1248)."""

    def __init__(self, unet_config, *args, **kwargs):
        feeder_configs = {k: kwargs.pop(k) for k in _FEEDER_KEYS if k in kwargs}
        uncond_type = kwargs.pop("uncond_type", "empty_seq")
        kwargs.setdefault("conditioning_key", "hybrid")
        image_size = kwargs.pop("image_size", [32, 32])
        super().__init__(unet_config, *args, image_size=image_size, **kwargs)
        self.feeder_configs, self.uncond_type = feeder_configs, uncond_type

    _FEEDER_ATTRS = {"first_stage_config": "first_stage_model", "image_proj_stage_config": "image_proj_model",
                     "pose_encoder_config": "pose_encoder", "multi_latent_adaptor": "multi_cond_latent_adaptor"}

    def build_feeders(self, which=None):
        """Instantiate the once-per-clip feeders built on the HIP kernels from the kept yaml configs (all of them, or the
        config keys in ``which``); returns the list of attribute names created."""
        made = []
        for key, attr in self._FEEDER_ATTRS.items():
            if key in self.feeder_configs and (which is None or key in which) and not hasattr(self, attr):
                mod = instantiate_from_config(self.feeder_configs[key]).eval()
                for p_ in mod.parameters():
                    p_.requires_grad = False
                setattr(self, attr, mod)
                made.append(attr)
        return made


class CameraControlLVDM(DynamiCrafter):
    """Adds camera conditioning to the UNet (reference model/base.py:20-70)."""

    def __init__(self, *args, add_type="add_into_temporal_attn", epipolar_config=None, normalize_T0=False,
                 camera_embedding="plucker", **kwargs):
        pose_cfg = kwargs.get("pose_encoder_config")
        super().__init__(*args, **kwargs)
        self.add_type = add_type
        self.normalize_T0 = normalize_T0
        self.camera_embedding = camera_embedding
        self.epipolar_config = dict(epipolar_config) if epipolar_config is not None else None
        unet = self.model.diffusion_model
        if self.epipolar_config is not None or pose_cfg is not None:
            epi = None
            if self.epipolar_config is not None:
                epi = {k: v for k, v in self.epipolar_config.items()
                       if k in ("origin_h", "origin_w", "is_3d_full_attn", "num_register_tokens",
                                "compression_factor", "attention_resolution", "only_on_cond_frame")}
                unet.epipolar_origin_h = epi.get("origin_h", 256)
            unet.enable_camera_conditioning(epi, pluker=pose_cfg is not None)

    def ray_condition(self, K, c2w, H, W, device=None, flip_flag=None):
        """Ray / Pluecker embedding of relative poses (reference model/base.py:112-174)."""
        from .pose import ray_condition
        return ray_condition(K, c2w, H, W, device, flip_flag, self.camera_embedding)

    @torch.no_grad()
    def pose_features(self, K, w2c, cond_frame_index, H_px, W_px, trace_scale_factor=1.0):
        """Relative poses -> embedding -> pose encoder -> the UNet's ``pluker_embedding_features`` list of
        [b, C_i, f, h_i, w_i] (reference model/camcontexti2v.py:531-537, 556-561)."""
        enc = getattr(self, "pose_encoder", None)
        if enc is None:
            raise RuntimeError("pose_features: call build_feeders() first (needs pose_encoder_config)")
        rel = camera.relative_c2w(w2c, cond_frame_index, trace_scale_factor)
        emb = self.ray_condition(K.float(), rel, H_px, W_px)
        b = emb.shape[0]
        return [f_.reshape(b, -1, *f_.shape[1:]).permute(0, 2, 1, 3, 4).contiguous() for f_ in enc(emb)]

    def camera_condition(self, K, w2c, cond_frame_index, H_px, W_px, pluker_features=None,
                         trace_scale_factor=1.0, generator=None, noise=None):
        """Geometry half of get_batch_input_camera_condition_process (model/camcontexti2v.py:525-572)."""
        cfg = self.epipolar_config or {}
        if self.epipolar_config is None or cfg.get("is_3d_full_attn", False):
            return {"pluker_embedding_features": pluker_features, "sample_locs_dict": None,
                    "cond_frame_index": cond_frame_index, "add_type": self.add_type}
        return camera.camera_condition(
            K, w2c, cond_frame_index, H_px, W_px, pluker_features, self.add_type,
            attention_resolution=cfg.get("attention_resolution", [8, 4, 2, 1]), trace_scale_factor=trace_scale_factor,
            perturb=cfg.get("add_small_perturbation_on_zero_T", False), generator=generator, noise=noise)


class CamContextI2V(CameraControlLVDM):
    """CamContextI2V (reference model/camcontexti2v.py:30-170): camera-conditioned UNet + context frames.
    The context-frame adaptor only changes the *inputs* (c_concat, c_crossattn length); the UNet is the
    same as CamI2V's."""

    def __init__(self, *args, multi_cond_strategy=None, use_zero_conv_latent_input=False, use_cross_normalization=False,
                 cross_normalization_mode="spatio_temporal", **kwargs):
        for k in ("plucker_proj_trainable", "epipolar_attn_trainable", "pose_guided_cond_trainable",
                  "multi_cond_adaptor_trainable", "first_unet_block_trainable", "first_unet_block_freeze_steps",
                  "use_semantic_branch", "epipolar_mask_freeze_steps",
                  "use_pose_embedding_in_latent_adaptor", "inject_trainable_lora_unet",
                  "lora_config", "diffusion_model_trainable_param_list", "pose_encoder_trainable",
                  "cond_stage_trainable", "image_proj_model_trainable", "weight_decay"):
            kwargs.pop(k, None)
        super().__init__(*args, **kwargs)
        self.multi_cond_strategy = multi_cond_strategy
        self.use_zero_conv_latent_input = use_zero_conv_latent_input
        if cross_normalization_mode not in ("token", "spatio_temporal"):
            raise ValueError(f"cross_normalization_mode {cross_normalization_mode!r}")
        self.cross_normalization_mode = cross_normalization_mode
        if multi_cond_strategy is not None:  # reference camcontexti2v.py:77-80
            from .adaptor import CrossNormalization
            self.use_cross_normalization = use_cross_normalization
            self.cond_cross_norm = CrossNormalization((-3, -2, -1))
        if use_zero_conv_latent_input:      # zero-initialised latent projection (reference camcontexti2v.py:81-84)
            self.multi_cond_in_projection = torch.nn.Conv3d(4, 4, kernel_size=3, stride=1, padding=1)
            torch.nn.init.constant_(self.multi_cond_in_projection.weight, 0.0)
            torch.nn.init.constant_(self.multi_cond_in_projection.bias, 0.0)

    @torch.no_grad()
    def context_concat(self, z_cond, z_context, K, w2c, w2c_context, cond_frame_index):
        """The ``c_concat`` latents of strategy 'token_concat_latent_epipolar' (reference camcontexti2v.py:334-377, with the
        optional cross normalisation and zero-initialised latent projection): z_cond [b, 4, h, w] latent of the conditioning frame,
        z_context [b, 4, n, h, w] latents of the extra context frames, K [b,t,3,3], w2c [b,t,4,4], w2c_context [b,n,4,4].
        Adaptor over [conditioning ; context] tokens with the target x context epipolar mask -> Conv3d(4,4,3) -> +
        conditioning latent on every frame.  Returns fp32 [b, 4, t, h, w]."""
        from . import ops
        adaptor = getattr(self, "multi_cond_latent_adaptor", None)
        if adaptor is None:
            raise RuntimeError("context_concat: call build_feeders() first (needs multi_latent_adaptor)")
        b, c, h, w = z_cond.shape
        t = w2c.shape[1]
        z_inp = torch.cat([z_cond[:, :, None], z_context], 2)                         # b c (1+n) h w
        tokens = z_inp.permute(0, 2, 3, 4, 1).reshape(b, -1, c).contiguous()           # 'B D C H W -> B (C H W) D'
        mask = None
        if adaptor.use_mask:
            F = camera.conditional_fundamental(K, w2c, w2c_context, cond_frame_index)
            mask = ops.epipolar_mask_bits(F, t, h, w, 8)
        lat = adaptor(tokens, mask)                                                    # b (t h w) c
        if getattr(self, "use_cross_normalization", False):
            # statistics over (c, h, w) of every frame ('spatio_temporal') or over the whole clip ('token') against the
            # conditioning latent's (camcontexti2v.py:354-364); the sets of elements are the same in token-major order
            if self.cross_normalization_mode == "spatio_temporal":
                lat = self.cond_cross_norm(lat.reshape(b, t, h * w, 1, c), z_cond.reshape(b, 1, c, h, w))
            else:
                lat = self.cond_cross_norm(lat.reshape(b, 1, t * h * w, c), z_cond)
        x = lat.reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3).contiguous()             # b c t h w
        if not self.use_zero_conv_latent_input:
            return x
        proj = self.multi_cond_in_projection
        return ops.conv3d_small(x, proj.weight, proj.bias, add=z_cond)


class CamI2V(CameraControlLVDM):
    """CamI2V baseline (reference baseline/cami2v/cami2v.py): same camera-conditioned UNet, no context frames."""

    def __init__(self, *args, **kwargs):
        for k in ("diffusion_model_trainable_param_list", "pose_encoder_trainable", "cond_stage_trainable",
                  "image_proj_model_trainable", "weight_decay"):
            kwargs.pop(k, None)
        super().__init__(*args, **kwargs)
