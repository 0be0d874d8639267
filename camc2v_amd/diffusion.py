"""Diffusion-process boundary of the hot path: DiffusionWrapper + the schedule/apply_model/sample_log
slice of LatentDiffusion (reference lvdm/models/ddpm3d.py:125-188, 724-739, 992-1002, 1251-1320).

Only what the sampler needs is here; training, losses, EMA, VAE and text/image encoders are outside the
hot path.  Checkpoint keys keep the reference prefixes
(``model.diffusion_model.*``, ``betas``, ``alphas_cumprod`` ...).
"""
import numpy as np
import torch
import torch.nn as nn

from . import rng
from .config import instantiate_from_config
from .lib import CcvError
from .sampler import DDIMSampler, make_beta_schedule, rescale_zero_terminal_snr


class DiffusionWrapper(nn.Module):
    """Routes conditioning into the UNet (reference ddpm3d.py:1251-1320): 'hybrid' concatenates ``c_concat``
    on the channel axis and ``c_crossattn`` on the token axis; all other cond keys pass through as kwargs."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat=None, c_crossattn=None, c_adm=None, s=None, mask=None, **kwargs):
        key = self.conditioning_key
        if key is None:
            return self.diffusion_model(x, t)
        if key == "concat":
            return self.diffusion_model(torch.cat([x] + list(c_concat), dim=1), t, **kwargs)
        # a single context tensor is handed over as is: the UNet caches its K/V projections by tensor identity
        ctx = (lambda: c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(list(c_crossattn), 1))
        if key == "crossattn":
            return self.diffusion_model(x, t, context=ctx(), **kwargs)
        if key == "hybrid":
            return self.diffusion_model(torch.cat([x] + list(c_concat), dim=1), t, context=ctx(), **kwargs)
        raise NotImplementedError(f"conditioning_key {key!r} is not used by the shipped configs")


class LatentDiffusionCore(nn.Module):
    """Schedule buffers + ``apply_model`` + ``sample_log``: what DDIMSampler needs from the model."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, parameterization="eps", conditioning_key=None, channels=3, image_size=256,
                 temporal_length=None, use_dynamic_rescale=False, rescale_betas_zero_snr=False, base_scale=0.7, turning_step=400,
                 first_stage_config=None, scale_factor=1.0, perframe_ae=False, encoder_type="2d", **ignored):
        super().__init__()
        # first-stage decoder: optional -- the DDIM path itself never touches it
        self.scale_factor, self.perframe_ae, self.encoder_type = float(scale_factor), bool(perframe_ae), encoder_type
        if first_stage_config is not None:
            self.first_stage_model = instantiate_from_config(first_stage_config).eval()
            for p_ in self.first_stage_model.parameters():
                p_.requires_grad = False
        if parameterization not in ("eps", "v"):
            raise NotImplementedError("x0 parameterisation has no sampling path in the reference's DDIMSampler either")
        # the shipped configs are eps-parameterised with a static scale; "v", use_dynamic_rescale and rescale_betas_zero_snr
        # (lvdm/models/ddpm3d.py:75,130-134,524-529) run the sampler's general (torch) step instead of the fused HIP step
        self.parameterization = parameterization
        self.use_dynamic_rescale = bool(use_dynamic_rescale)
        self.channels = channels
        self.image_size = image_size if isinstance(image_size, (list, tuple)) else [image_size, image_size]
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.temporal_length = temporal_length if temporal_length is not None else getattr(
            self.model.diffusion_model, "temporal_length", None)
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end, cosine_s)
        if rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = int(timesteps)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(np.append(1.0, ac[:-1])))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - ac)))
        if self.use_dynamic_rescale:      # ddpm3d.py:524-529
            scale_arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(self.num_timesteps, base_scale)))
            self.register_buffer("scale_arr", f32(scale_arr))

    @property
    def device(self):
        return self.betas.device

    def q_sample(self, x_start, t, noise=None):
        noise = rng.randn_like(x_start) if noise is None else noise
        shape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
        return (self.sqrt_alphas_cumprod[t].reshape(shape) * x_start
                + self.sqrt_one_minus_alphas_cumprod[t].reshape(shape) * noise)

    def _extract(self, table, t, x):
        return table[t].reshape((x.shape[0],) + (1,) * (x.dim() - 1))

    def predict_start_from_z_and_v(self, x_t, t, v):          # ddpm3d.py:241-247
        return self._extract(self.sqrt_alphas_cumprod, t, x_t) * x_t - self._extract(self.sqrt_one_minus_alphas_cumprod, t, x_t) * v

    def predict_eps_from_z_and_v(self, x_t, t, v):            # ddpm3d.py:249-253
        return self._extract(self.sqrt_alphas_cumprod, t, x_t) * v + self._extract(self.sqrt_one_minus_alphas_cumprod, t, x_t) * x_t

    def _as_dict(self, cond):
        if isinstance(cond, dict):
            return cond
        if not isinstance(cond, list):
            cond = [cond]
        return {("c_concat" if self.model.conditioning_key == "concat" else "c_crossattn"): cond}

    def apply_model(self, x_noisy, t, cond, **kwargs):
        out = self.model(x_noisy, t, **self._as_dict(cond), **kwargs)
        return out[0] if isinstance(out, tuple) else out

    @staticmethod
    def _same_extras(c, uc):
        """True when everything besides c_concat / c_crossattn is shared by the two CFG halves: the same objects, or --
        for the camera dict -- the shallow copy the sampler puts into the unconditional dict (same tensors, plus the
        'is_uc' marker)."""
        keys = (set(c) | set(uc)) - {"c_concat", "c_crossattn"}
        for k in keys:
            if k not in c or k not in uc:
                return False
            a, b = c[k], uc[k]
            if a is b:
                continue
            if k == "camera_condition" and isinstance(a, dict) and isinstance(b, dict):
                ka = set(a) - {"is_uc"}
                plain = lambda v: isinstance(v, (str, int, float, bool, type(None)))
                if ka != set(b) - {"is_uc"} or any(not (a[f] is b[f] or (plain(a[f]) and plain(b[f]) and a[f] == b[f])) for f in ka):
                    return False
                continue
            return False
        return True

    def apply_model_pair(self, x, t, cond, uncond, **kwargs):
        """Conditional and unconditional eps in ONE UNet forward on a 2b batch.  The two contexts may have
        different lengths (cond: 77+256(1+N) tokens, uncond: 77+16t): they are handed over as a list and
        only the cross-attention runs per half.  Everything that is not c_concat / c_crossattn must be
        shared by the two halves (the sampler shares the camera dict; fs and flags come in kwargs)."""
        c, uc = self._as_dict(cond), self._as_dict(uncond)
        if self.model.conditioning_key != "hybrid" or not self._same_extras(c, uc):
            # e.g. a conditional dict with a camera but enable_camera_condition off: the reference's second apply_model
            # then runs WITHOUT the camera (ddim.py:258-263), which one batched forward cannot express
            return self.apply_model(x, t, c, **kwargs), self.apply_model(x, t, uc, **kwargs)
        b = x.shape[0]
        # a single context tensor is handed over as is: the UNet caches its K/V projections by tensor identity
        one = lambda parts: parts[0] if len(parts) == 1 else torch.cat(list(parts), 1)
        ctx = [one(c["c_crossattn"]), one(uc["c_crossattn"])]
        extra = {k: v for k, v in c.items() if k not in ("c_concat", "c_crossattn")}
        kw = dict(kwargs)
        cc, ucc = list(c["c_concat"]), list(uc["c_concat"])
        if len(cc) == len(ucc) and all(p is q for p, q in zip(cc, ucc)):
            # both halves start from the very same tensors: the context-free head of the UNet runs once
            out = self.model.diffusion_model(torch.cat([x] + cc, 1), t, context=ctx, cfg_shared_input=True, **extra, **kw)
            return out[:b], out[b:]
        xc = torch.cat([torch.cat([x] + cc, 1), torch.cat([x] + ucc, 1)], 0)
        if kw.get("fs") is not None:
            kw["fs"] = torch.cat([kw["fs"], kw["fs"]], 0)
        out = self.model.diffusion_model(xc, torch.cat([t, t], 0), context=ctx, **extra, **kw)
        return out[:b], out[b:]

    @torch.no_grad()
    def encode_first_stage(self, x, noise=None):
        """Frames [b, 3, t, H, W] (or [n, 3, H, W]) -> latents, reference ddpm3d.py:621-646: every frame a batch element of
        the 2-D encoder, scale_factor * posterior.sample() (``noise``: the N(0,1) draw to use, shaped like the latents)."""
        fsm = getattr(self, "first_stage_model", None)
        if fsm is None:
            raise CcvError("encode_first_stage: the model was built without first_stage_config")
        five = x.dim() == 5
        if five:
            b, c, t, hh, ww = x.shape
            x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
            if noise is not None:
                noise = noise.permute(0, 2, 1, 3, 4).reshape(b * t, *noise.shape[1:2], *noise.shape[3:])
        if self.perframe_ae:
            zs = [fsm.encode(x[i:i + 1]).sample(None if noise is None else noise[i:i + 1]) for i in range(x.shape[0])]
            z = torch.cat(zs, 0)
        else:
            z = fsm.encode(x).sample(noise)
        z = self.scale_factor * z
        if five:
            z = z.reshape(b, t, z.shape[1], z.shape[2], z.shape[3]).permute(0, 2, 1, 3, 4)
        return z

    @torch.no_grad()
    def decode_first_stage(self, z, **kwargs):
        """Latents [b, c, t, h, w] (or [n, c, h, w]) -> frames, reference ddpm3d.py:648-673: 1/scale_factor, every frame a
        batch element of the 2-D decoder (perframe_ae decodes them one by one: same result, less memory)."""
        fsm = getattr(self, "first_stage_model", None)
        if fsm is None:
            raise CcvError("decode_first_stage: the model was built without first_stage_config")
        five = z.dim() == 5
        if five:
            b, c, t, hh, ww = z.shape
            z = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
        z = z.float() * (1.0 / self.scale_factor)
        if self.perframe_ae:
            out = torch.cat([fsm.decode(z[i:i + 1], **kwargs) for i in range(z.shape[0])], 0)
        else:
            out = fsm.decode(z, **kwargs)
        if five:
            out = out.reshape(b, t, out.shape[1], out.shape[2], out.shape[3]).permute(0, 2, 1, 3, 4)
        return out

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        if not ddim:
            raise NotImplementedError("ancestral DDPM sampling is not on the generation path")
        shape = (self.channels, self.temporal_length, *self.image_size)
        return DDIMSampler(self).sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)


__all__ = ["DiffusionWrapper", "LatentDiffusionCore", "CcvError"]
