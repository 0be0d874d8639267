"""fp32/fp64 CPU restatement of the DDIM sampler used by CamContextI2V generation.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference anchors (relative to /root/reference/CamContextI2V):
  make_beta_schedule               lvdm/models/utils_diffusion.py:31-53
  make_ddim_timesteps              lvdm/models/utils_diffusion.py:56-76
  make_ddim_sampling_parameters    lvdm/models/utils_diffusion.py:79-91
  rescale_noise_cfg                lvdm/models/utils_diffusion.py:147-157
  DDPM.register_schedule           lvdm/models/ddpm3d.py:125-188 (alphas_cumprod)
  DDIMSampler.make_schedule        lvdm/models/samplers/ddim.py:24-57
  DDIMSampler.p_sample_ddim        lvdm/models/samplers/ddim.py:240-346
  DDIMSampler.ddim_sampling        lvdm/models/samplers/ddim.py:133-238
"""
import math

import numpy as np
import torch


def alphas_cumprod(timesteps=1000, linear_start=0.00085, linear_end=0.012):
    """'linear' schedule = linspace in sqrt(beta), float64 (utils_diffusion.py:31-36),
    cumprod in float64, stored as fp32 (ddpm3d.py:141-150)."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0)


def ddim_timesteps(method, num_ddim, num_ddpm=1000):
    """utils_diffusion.py:56-76."""
    if method == "uniform":
        c = num_ddpm // num_ddim
        return np.asarray(list(range(0, num_ddpm, c))) + 1
    if method == "uniform_trailing":
        c = num_ddpm / num_ddim
        return np.flip(np.round(np.arange(num_ddpm, 0, -c))).astype(np.int64) - 1
    if method == "quad":
        return (np.linspace(0, np.sqrt(num_ddpm * 0.8), num_ddim) ** 2).astype(int) + 1
    raise NotImplementedError(method)


def ddim_tables(num_ddim, eta, method="uniform_trailing", num_ddpm=1000,
                linear_start=0.00085, linear_end=0.012):
    """Returns dict(timesteps, alphas, alphas_prev, sigmas, sqrt_one_minus_alphas) as the
    sampler sees them: the model keeps alphas_cumprod in fp32 (register_buffer with
    to_torch = float32) and make_ddim_sampling_parameters indexes that fp32 tensor."""
    ac = torch.tensor(alphas_cumprod(num_ddpm, linear_start, linear_end), dtype=torch.float32)
    ts = ddim_timesteps(method, num_ddim, num_ddpm)
    a = ac[ts.copy()]
    # The reference builds alphas_prev as a float64 numpy array of the fp32 values and mixes
    # it with the fp32 tensor `alphas`: torch promotes that arithmetic to float64
    # (utils_diffusion.py:82-86).  Every table entry is later rounded to fp32 by
    # torch.full(size, table[index]) in p_sample_ddim (ddim.py:305-308).
    a_prev64 = torch.tensor([ac[0].item()] + ac[ts[:-1].copy()].tolist(), dtype=torch.float64)
    a64 = a.double()
    sig = (eta * torch.sqrt((1 - a_prev64) / (1 - a64) * (1 - a64 / a_prev64))).float()
    return dict(timesteps=ts, alphas=a, alphas_prev=a_prev64.float(), sigmas=sig,
                sqrt_one_minus_alphas=torch.sqrt(1.0 - a))


def rescale_noise_cfg(noise_cfg, noise_text, guidance_rescale):
    """utils_diffusion.py:147-157 (unbiased std over all non-batch dims)."""
    dims = list(range(1, noise_text.ndim))
    std_text = noise_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def camera_cfg_weight(t, scheduler="constant"):
    """ddim.py:272-277: weight of the camera-guidance term for timesteps t (long [b]), one per sample.  (The reference
    reshapes it to [b, 1, 1, 1], which broadcasts against [b, c, t, h, w] latents only for b == 1; per-sample is the
    reading that agrees with it there.)"""
    if scheduler == "constant":
        return 1.0
    if scheduler == "cosine":
        return ((1.0 - t / 999) * math.pi / 2).cos().reshape(-1, *([1] * 4))
    raise NotImplementedError(scheduler)


def cfg_ddim_update(x, e_c, e_uc, noise, a_t, a_prev, sigma_t, sqrt_1m_at, scale, guidance_rescale,
                    e_nc=None, camera_cfg=1.0, camera_weight=1.0):
    """One guidance + DDIM update, ddim.py:267-346 (eps parameterisation, no dynamic rescale).

    e_uc None => no guidance (scale 1).  noise: the N(0,1) draw (injected; the
    reference draws it with noise_like).  e_nc: conditional prediction without the camera (third forward), used when
    camera_cfg != 1 (ddim.py:268-280).  Returns (x_prev, pred_x0, e_t)."""
    if e_uc is None:
        e = e_c
    else:
        e = e_uc + scale * (e_c - e_uc)
        if camera_cfg != 1.0:
            e = e + (camera_cfg - 1.0) * camera_weight * (e_c - e_nc)
        if guidance_rescale > 0.0:
            e = rescale_noise_cfg(e, e_c, guidance_rescale)
    pred_x0 = (x - sqrt_1m_at * e) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).clamp(min=0).sqrt() * e
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * noise
    return x_prev, pred_x0, e


def ddim_sample(apply_cond, apply_uncond, x_T, num_steps, eta, scale, guidance_rescale,
                noises=None, method="uniform_trailing", apply_nocam=None, camera_cfg=1.0, camera_cfg_scheduler="constant"):
    """ddim.py:133-238 restricted to the generation kwargs of 02_generate_videos.py:318-327.

    apply_cond(x, t_long[b]) -> eps ; apply_uncond likewise (None => no CFG).
    noises: list of per-step N(0,1) tensors (index = loop iteration) or None (=> zeros,
    exact for eta == 0).  apply_nocam + camera_cfg != 1: the third forward of camera guidance (ddim.py:268-280).
    Returns (x_0, [x after every step])."""
    tab = ddim_tables(num_steps, eta, method)
    ts = tab["timesteps"]
    x = x_T
    trace = []
    b = x.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = len(ts) - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        e_c = apply_cond(x, t)
        e_uc = apply_uncond(x, t) if (apply_uncond is not None and scale != 1.0) else None
        z = noises[i] if noises is not None else torch.zeros_like(x)
        cam = e_uc is not None and apply_nocam is not None and camera_cfg != 1.0
        x, _, _ = cfg_ddim_update(x, e_c, e_uc, z, tab["alphas"][index], tab["alphas_prev"][index],
                                  tab["sigmas"][index], tab["sqrt_one_minus_alphas"][index],
                                  scale, guidance_rescale, e_nc=apply_nocam(x, t) if cam else None,
                                  camera_cfg=camera_cfg if cam else 1.0,
                                  camera_weight=camera_cfg_weight(t, camera_cfg_scheduler) if cam else 1.0)
        trace.append(x)
    return x, trace
