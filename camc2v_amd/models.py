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
import threading

import torch

from . import camera, rng
from .config import instantiate_from_config
from .diffusion import LatentDiffusionCore
from .lib import CcvError

_NULL_PROMPT_LOCK = threading.Lock()

_FEEDER_KEYS = ("first_stage_config", "cond_stage_config", "img_cond_stage_config", "image_proj_stage_config",
                "pose_encoder_config", "multi_latent_adaptor", "pose_guided_cond_encoder_config")


class DynamiCrafter(LatentDiffusionCore):
    """Image-to-video base model (reference model/dynamicrafter.py, lvdm/models/ddpm3d.py:1030-1248)."""

    def __init__(self, unet_config, *args, **kwargs):
        feeder_configs = {k: kwargs.pop(k) for k in _FEEDER_KEYS if k in kwargs}
        uncond_type = kwargs.pop("uncond_type", "empty_seq")
        # keys of the batch dict / sampling-side switches of the reference constructor (lvdm/models/ddpm3d.py:47-62, 1033-1040)
        self_keys = dict(first_stage_key=kwargs.pop("first_stage_key", "video"), cond_stage_key=kwargs.pop("cond_stage_key", "caption"),
                         uncond_prob=kwargs.pop("uncond_prob", 0.05), rand_cond_frame=kwargs.pop("rand_cond_frame", False),
                         fps_condition_type=kwargs.pop("fps_condition_type", "fs"), interp_mode=kwargs.pop("interp_mode", False))
        kwargs.setdefault("conditioning_key", "hybrid")
        image_size = kwargs.pop("image_size", [32, 32])
        super().__init__(unet_config, *args, image_size=image_size, **kwargs)
        self.feeder_configs, self.uncond_type = feeder_configs, uncond_type
        for k, v in self_keys.items():
            setattr(self, k, v)
        self.use_semantic_branch = False
        self.multi_cond_strategy = None

    _FEEDER_ATTRS = {"first_stage_config": "first_stage_model", "image_proj_stage_config": "image_proj_model",
                     "pose_encoder_config": "pose_encoder", "multi_latent_adaptor": "multi_cond_latent_adaptor"}

    _ENCODER_ATTRS = {"cond_stage_config": "cond_stage_model", "img_cond_stage_config": "embedder"}

    def build_feeders(self, which=None, encoders=False):
        """Instantiate the once-per-clip feeders built on the HIP kernels from the kept yaml configs (all of them, or the
        config keys in ``which``); returns the list of attribute names created.  ``encoders=True`` also builds the OpenCLIP
        text / image embedders (``cond_stage_model`` / ``embedder``: ~1 B parameters whose weights come from the checkpoint);
        without them the batch has to carry their outputs (``caption_emb``, ``image_clip_tokens``)."""
        made = []
        attrs = dict(self._FEEDER_ATTRS, **(self._ENCODER_ATTRS if encoders else {}))
        for key, attr in attrs.items():
            if key in self.feeder_configs and (which is None or key in which) and not hasattr(self, attr):
                mod = instantiate_from_config(self.feeder_configs[key]).eval()
                for p_ in mod.parameters():
                    p_.requires_grad = False
                setattr(self, attr, mod)
                made.append(attr)
        return made


    # ---- batch -> conditioning (sampling side of get_batch_input; reference model/base.py:237-345, model/camcontexti2v.py:280-491)
    def get_input(self, batch, k):
        """lvdm/models/ddpm3d.py:370-378, plus the move to the model's device (Lightning does that for the reference)."""
        return batch[k].to(device=self.device, memory_format=torch.contiguous_format).float()

    def get_learned_conditioning(self, c):
        """Text encoder call (ddpm3d.py:600-611).  ``cond_stage_model`` (camc2v_amd/clip.py: ~354 M parameters whose weights come
        from the checkpoint) is built by ``build_feeders(encoders=True)``; without it the embeddings come in the batch (``caption_emb``)."""
        enc = getattr(self, "cond_stage_model", None)
        if enc is None:
            raise CcvError("no text encoder attached: call model.build_feeders(encoders=True) (and load the checkpoint) or pass "
                           "'caption_emb' [b, 77, 1024] / 'null_caption_emb' in the batch")
        return enc.encode(c) if callable(getattr(enc, "encode", None)) else enc(c)

    def _text_embeddings(self, batch, n):
        """(cond_emb [n,77,D], null_prompt [1,77,D]) from the batch's precomputed embeddings or the text encoder."""
        if batch.get("caption_emb") is not None:
            cond_emb = self.get_input(batch, "caption_emb")
        else:
            cond_input = batch[self.cond_stage_key]
            cond_emb = self.get_learned_conditioning(cond_input if isinstance(cond_input, (dict, list)) else cond_input.to(self.device))
        if batch.get("null_caption_emb") is not None:
            null = self.get_input(batch, "null_caption_emb")[:1]
        else:
            if not hasattr(self, "null_prompt"):
                with _NULL_PROMPT_LOCK:      # several lanes (host thread + stream each) may get here together: one encodes, and the
                    if not hasattr(self, "null_prompt"):     # result is published only once its kernels have run
                        null_prompt = self.get_learned_conditioning([""])
                        if torch.is_tensor(null_prompt) and null_prompt.is_cuda:
                            torch.cuda.current_stream(null_prompt.device).synchronize()
                        self.null_prompt = null_prompt
            null = self.null_prompt
        return cond_emb, null

    def _image_tokens(self, batch, key, img, frames_per_sample=1):
        """OpenCLIP image tokens [n, 257, 1280] of the frames `img` [n, 3, H, W]: ``self.embedder`` when one is attached, else
        the precomputed tokens under batch[key] ([b, 257, 1280] or [b, 1 + n_ctx, 257, 1280]: conditioning frame first)."""
        emb = getattr(self, "embedder", None)
        if emb is not None:
            return emb(img)
        if batch.get(key) is None:
            raise CcvError(f"no image encoder attached: call model.build_feeders(encoders=True) (and load the checkpoint) or pass '{key}' in the batch")
        tok = self.get_input(batch, key)
        if tok.dim() == 4:
            if tok.shape[1] < frames_per_sample:
                raise CcvError(f"'{key}' holds tokens of {tok.shape[1]} frames per sample, {frames_per_sample} are needed")
            tok = tok[:, :frames_per_sample]
        return tok.reshape(-1, tok.shape[-2], tok.shape[-1])

    def _project_image_tokens(self, tokens):
        proj = getattr(self, "image_proj_model", None)
        if proj is None:
            raise CcvError("image_proj_model is not built: call model.build_feeders() (needs image_proj_stage_config)")
        return proj(tokens)

    def get_batch_input_camera_condition_process(self, batch, x, cond_frame_index, trace_scale_factor, rand_cond_frame, *a, **k):
        return {}, {}

    def _context_latents(self, batch, x, z_all, cond_frame_index, camera_kwargs):
        """c_concat latents [b, 4, t, h, w]; the base models repeat the conditioning frame's latent (base.py:290-300)."""
        b, t = z_all.shape[0], z_all.shape[2]
        if self.interp_mode:
            cat = torch.zeros_like(z_all)
            cat[:, :, 0], cat[:, :, -1] = z_all[:, :, 0], z_all[:, :, -1]
            return cat
        zc = z_all[torch.arange(b, device=z_all.device), :, cond_frame_index]
        return zc.unsqueeze(2).repeat(1, 1, t, 1, 1)

    @torch.no_grad()
    def get_batch_input(self, batch, random_uncond, return_first_stage_outputs=False, return_original_cond=False, return_fs=False,
                        return_cond_frame_index=False, return_cond_frame=False, return_original_input=False, rand_cond_frame=None,
                        enable_camera_condition=True, return_camera_data=False, return_video_path=False, return_depth_scale=False,
                        trace_scale_factor=1.0, cond_frame_index=None, **kwargs):
        """Sampling-side restatement of get_batch_input: batch dict (the dataset's keys: video, caption, video_path, fps,
        frame_stride, RT, camera_data, camera_intrinsics, cond_frames, RT_cond) -> [z, cond, ...] in the reference's order.
        ``random_uncond`` (training-time condition dropout) must be False here."""
        if random_uncond:
            raise NotImplementedError("random_uncond is the training-time condition dropout; the sampling path passes False")
        x = self.get_input(batch, self.first_stage_key)                     # b c t h w
        tl = getattr(self.model.diffusion_model, "temporal_length", None) or x.shape[2]
        if x.shape[2] > tl:                                                  # base.py:243-246 (correct_batch)
            x = x[:, :, :tl].contiguous()
        b, T, device = x.shape[0], x.shape[2], self.device
        if cond_frame_index is None:
            cond_frame_index = torch.zeros(b, device=device, dtype=torch.long)
            rand = self.rand_cond_frame if rand_cond_frame is None else rand_cond_frame
            if rand:
                cond_frame_index = rng.randint(0, tl, (b,), device=device)
        depth_scale = torch.ones((b,), device=device)
        camera_kwargs = {}
        if enable_camera_condition:
            cam_log, camera_kwargs = self.get_batch_input_camera_condition_process(batch, x, cond_frame_index, trace_scale_factor, rand_cond_frame)
            depth_scale = cam_log.get("depth_scale", depth_scale)

        cond_frames = None
        if batch.get("cond_frames") is not None:
            cond_frames = self.get_input(batch, "cond_frames")              # b n c h w
        x_all = x
        if cond_frames is not None and self.multi_cond_strategy in ("token_concat_latent", "token_concat_latent_epipolar"):
            x_all = torch.cat([x, cond_frames.permute(0, 2, 1, 3, 4)], 2)
        z_all = self.encode_first_stage(x_all, noise=batch.get("first_stage_noise"))
        z = z_all[:, :, :T].contiguous()
        img_cat_cond = self._context_latents(batch, x, z_all, cond_frame_index, camera_kwargs)

        cond_emb, null_prompt = self._text_embeddings(batch, b)
        prompt_imb = cond_emb.detach()                                       # prompt_mask is all False without dropout
        bi = torch.arange(b, device=device)
        img = x[bi, :, cond_frame_index]
        n_img = 1
        if self.use_semantic_branch and cond_frames is not None:
            n_img = 1 + cond_frames.shape[1]
            img = torch.cat([img.unsqueeze(1), cond_frames], 1).reshape(b * n_img, *img.shape[1:])
        img_emb = self._project_image_tokens(self._image_tokens(batch, "image_clip_tokens", img, n_img))
        if n_img > 1:                                                        # multi_cond_func of the token_concat strategies (camcontexti2v.py:629-640)
            img_emb = img_emb.reshape(b, n_img * img_emb.shape[-2], img_emb.shape[-1])
        cond = {}
        if self.model.conditioning_key == "hybrid":
            cond["c_concat"] = [img_cat_cond]
            cond["c_cond_frame_index"] = cond_frame_index
            cond["origin_z_0"] = z.clone()
        cond["c_crossattn"] = [torch.cat([prompt_imb, img_emb], 1).contiguous()]
        cond.update(camera_kwargs)

        out = [z, cond]
        if return_first_stage_outputs:
            out.append(self.decode_first_stage(z))
        if return_original_cond:
            out.append(batch[self.cond_stage_key])
        if return_fs:
            out.append(self.get_input(batch, "frame_stride" if self.fps_condition_type == "fs" else "fps"))
        if return_cond_frame_index:
            out.append(cond_frame_index)
        if return_cond_frame:
            out.append(x[bi, :, cond_frame_index].unsqueeze(2))
        if return_original_input:
            out.append(x)
        if return_camera_data:
            out.append(batch.get("camera_data"))
        if return_video_path:
            out.append(batch["video_path"])
        if return_depth_scale:
            out.append(depth_scale)
        return out

    def log_images_sample_log_pre_process(self, *args, **kwargs):
        return {}, {}

    def log_images_sample_log_post_process(self, *args, **kwargs):
        return {}

    @torch.no_grad()
    def log_images(self, batch, sample=True, ddim_steps=50, ddim_eta=1.0, plot_denoise_rows=False, unconditional_guidance_scale=1.0,
                   mask=None, sampled_img_num=1, enable_camera_condition=True, trace_scale_factor=1.0, cond_frame_index=None,
                   **kwargs):
        """What ImageLogger calls on every test batch (main/callbacks.py:163-181): conditioning -> 25-step CFG DDIM -> decode.
        Same signature, batch handling and log keys as the reference (model/camcontexti2v.py:646-767); kwargs such as
        ``timestep_spacing`` / ``guidance_rescale`` / ``use_graph`` go to the sampler, ``split`` is accepted and ignored."""
        kwargs.pop("split", None)
        for key in list(batch.keys()):
            v = batch[key]
            if v is None or isinstance(v, (str, float, int)):
                continue
            if isinstance(v, list) and len(v) < sampled_img_num:
                continue
            batch[key] = v[:sampled_img_num]
        if plot_denoise_rows:
            raise NotImplementedError("plot_denoise_rows is a training-time visualisation")
        log = dict()
        z, c, xrec, xc, fs, cond_frame_index, cond_x, x, camera_data, video_path, depth_scale = self.get_batch_input(
            batch, random_uncond=False, return_first_stage_outputs=True, return_original_cond=True, return_fs=True,
            return_cond_frame_index=True, return_cond_frame=True, rand_cond_frame=False, enable_camera_condition=enable_camera_condition,
            return_original_input=True, return_camera_data=True, return_video_path=True, return_depth_scale=True,
            trace_scale_factor=trace_scale_factor, cond_frame_index=cond_frame_index)
        N = xrec.shape[0]
        log.update(depth_scale=depth_scale, camera_data=camera_data, video_path=video_path, gt_video=x, image_condition=cond_x, reconst=xrec)
        if batch.get("cond_frames") is not None:
            log["cond_frames"] = batch["cond_frames"]
        log["condition"] = [f"{content}_fs={fs[idx].item()}" for idx, content in enumerate(xc)]
        kwargs.update({"fs": fs.long()})
        if not sample:
            return log
        uc = None
        if unconditional_guidance_scale != 1.0:
            c_emb = c["c_crossattn"][0] if isinstance(c, dict) else c
            if self.uncond_type == "empty_seq":
                _, null = self._text_embeddings(batch, N)
                uc_prompt = null.expand(N, -1, -1)
            elif self.uncond_type == "zero_embed":
                uc_prompt = torch.zeros_like(c_emb[:, :77])
            elif self.uncond_type == "negative_prompt":
                uc_prompt = self.get_learned_conditioning(N * [kwargs["negative_prompt"]])
            else:
                raise ValueError(f"uncond_type {self.uncond_type!r}")
            img0 = torch.zeros_like(xrec[:, :, 0])                                              # the all-zero image
            uc_img = self._project_image_tokens(self._image_tokens(batch, "uncond_image_clip_tokens", img0))
            if uc_img.shape[0] == 1 and N > 1:
                uc_img = uc_img.expand(N, -1, -1)
            uc = torch.cat([uc_prompt.to(uc_img), uc_img], 1).contiguous()
            if isinstance(c, dict):
                uc = {"c_concat": [c["c_concat"][0]], "c_crossattn": [uc]} if "c_concat" in c else {"c_crossattn": [uc]}
        pre_log, pre_kwargs = self.log_images_sample_log_pre_process(batch, z, x, cond_frame_index, trace_scale_factor, **kwargs)
        log.update(pre_log)
        kwargs.update(pre_kwargs)
        kwargs.pop("negative_prompt", None)
        samples, _ = self.sample_log(cond=c, batch_size=N, ddim=ddim_steps is not None, ddim_steps=ddim_steps, eta=ddim_eta,
                                     unconditional_guidance_scale=unconditional_guidance_scale, unconditional_conditioning=uc, x0=z,
                                     enable_camera_condition=enable_camera_condition, **kwargs)
        log["samples"] = self.decode_first_stage(samples)
        log.update(self.log_images_sample_log_post_process(log["samples"], **pre_log))
        return log

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


    def get_batch_input_camera_condition_process(self, batch, x, cond_frame_index, trace_scale_factor, rand_cond_frame, *a, **k):
        """model/camcontexti2v.py:525-572 / model/base.py: intrinsics + world-to-camera poses of the batch -> the UNet's
        ``camera_condition`` dict (epipolar masks in packed form, Pluecker features through the pose encoder when built)."""
        K = self.get_input(batch, "camera_intrinsics")      # b t 3 3
        w2c = self.get_input(batch, "RT")                   # b t 4 4
        T = x.shape[2]
        K, w2c = K[:, :T], w2c[:, :T]
        H, W = x.shape[3], x.shape[4]
        feats = None
        if getattr(self, "pose_encoder", None) is not None:
            feats = self.pose_features(K, w2c, cond_frame_index, H, W, trace_scale_factor)
        cam = self.camera_condition(K, w2c, cond_frame_index, H, W, pluker_features=feats, trace_scale_factor=trace_scale_factor,
                                    noise=batch.get("perturbation_noise"))
        return {}, {"camera_condition": cam}


class CamContextI2V(CameraControlLVDM):
    """CamContextI2V (reference model/camcontexti2v.py:30-170): camera-conditioned UNet + context frames.
    The context-frame adaptor only changes the *inputs* (c_concat, c_crossattn length); the UNet is the
    same as CamI2V's."""

    def __init__(self, *args, multi_cond_strategy=None, use_zero_conv_latent_input=False, use_cross_normalization=False,
                 cross_normalization_mode="spatio_temporal", **kwargs):
        use_semantic_branch = kwargs.get("use_semantic_branch", True)      # reference default (camcontexti2v.py:50)
        for k in ("plucker_proj_trainable", "epipolar_attn_trainable", "pose_guided_cond_trainable",
                  "multi_cond_adaptor_trainable", "first_unet_block_trainable", "first_unet_block_freeze_steps",
                  "use_semantic_branch", "epipolar_mask_freeze_steps",
                  "use_pose_embedding_in_latent_adaptor", "inject_trainable_lora_unet",
                  "lora_config", "diffusion_model_trainable_param_list", "pose_encoder_trainable",
                  "cond_stage_trainable", "image_proj_model_trainable", "weight_decay"):
            kwargs.pop(k, None)
        super().__init__(*args, **kwargs)
        self.multi_cond_strategy = multi_cond_strategy
        self.use_semantic_branch = use_semantic_branch
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


    def _context_latents(self, batch, x, z_all, cond_frame_index, camera_kwargs):
        """c_concat of strategy 'token_concat_latent_epipolar' (camcontexti2v.py:334-377): the adaptor over [conditioning ;
        context] latents with the target x context epipolar mask; without context frames, or with another strategy, the base
        behaviour."""
        if batch.get("cond_frames") is None or self.multi_cond_strategy != "token_concat_latent_epipolar":
            if self.multi_cond_strategy == "token_concat_latent":
                raise NotImplementedError("multi_cond_strategy 'token_concat_latent' is not used by the shipped config")
            return super()._context_latents(batch, x, z_all, cond_frame_index, camera_kwargs)
        if getattr(getattr(self, "multi_cond_latent_adaptor", None), "use_plucker_embedding", False):
            raise NotImplementedError("use_plucker_embedding in the latent adaptor is off in the shipped config")
        b, T = x.shape[0], x.shape[2]
        n = z_all.shape[2] - T
        bi = torch.arange(b, device=z_all.device)
        z_cond = z_all[bi, :, cond_frame_index]                       # b 4 h w
        z_ctx = z_all[:, :, T:T + n]                                   # b 4 n h w
        K = self.get_input(batch, "camera_intrinsics")[:, :T]
        w2c = self.get_input(batch, "RT")[:, :T]
        w2c_ctx = self.get_input(batch, "RT_cond")
        return self.context_concat(z_cond, z_ctx, K, w2c, w2c_ctx, cond_frame_index)


class CamI2V(CameraControlLVDM):
    """CamI2V baseline (reference baseline/cami2v/cami2v.py): same camera-conditioned UNet, no context frames."""

    def __init__(self, *args, **kwargs):
        for k in ("diffusion_model_trainable_param_list", "pose_encoder_trainable", "cond_stage_trainable",
                  "image_proj_model_trainable", "weight_decay"):
            kwargs.pop(k, None)
        super().__init__(*args, **kwargs)
