"""Cases for the sampler's side branches (mask / x0 blending, pasted overlap frames, noise shaping, pasted conditioning frame, v
parameterisation, dynamic rescale: lvdm/models/samplers/ddim.py:174-199, 226-235, 285-335) around an ANALYTIC noise model, shared by
oracle/gen_golden_sampler_branches.py (which drives the REFERENCE's DDIMSampler with them) and tests/test_sampler_branches_gpu.py (which
drives camc2v_amd.sampler.DDIMSampler).  TEST INFRASTRUCTURE."""
import numpy as np
import torch

SEED = 20230211
STEPS, SCALE, RESCALE = 5, 3.0, 0.5
SHAPE = (4, 6, 4, 4)          # c, t, h, w
BATCH = 2


def schedule(zero_snr=False):
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    if zero_snr:
        ab_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
        first, last = ab_sqrt[0].copy(), ab_sqrt[-1].copy()
        ab_sqrt = (ab_sqrt - last) * (first / (first - last))
        ab = ab_sqrt ** 2
        betas = 1.0 - np.concatenate([ab[0:1], ab[1:] / ab[:-1]])
    return betas


class DuckModel:
    """What DDIMSampler reads from its model, around eps(x, t, c) = tanh(0.7 x + c.bias) (0.5 + t / 2000)."""

    def __init__(self, device, parameterization="eps", dynamic_rescale=False, q_noises=None):
        ac = np.cumprod(1.0 - schedule(), axis=0)
        f = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        self.num_timesteps = 1000
        self.device = torch.device(device)
        self.betas = f(schedule())
        self.alphas_cumprod = f(ac)
        self.alphas_cumprod_prev = f(np.append(1.0, ac[:-1]))
        self.sqrt_alphas_cumprod = f(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f(np.sqrt(1.0 - ac))
        self.parameterization = parameterization
        self.use_dynamic_rescale = dynamic_rescale
        if dynamic_rescale:
            self.scale_arr = f(np.concatenate((np.linspace(1.0, 0.7, 400), np.full(1000, 0.7))))
        self._q = iter(q_noises or [])

    def _ext(self, tab, t, x):
        return tab[t].reshape((x.shape[0],) + (1,) * (x.dim() - 1))

    def q_sample(self, x_start, t, noise=None):
        noise = next(self._q).to(x_start)[:, :, : x_start.shape[2]] if noise is None else noise
        return self._ext(self.sqrt_alphas_cumprod, t, x_start) * x_start + self._ext(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise

    def predict_start_from_z_and_v(self, x_t, t, v):
        return self._ext(self.sqrt_alphas_cumprod, t, x_t) * x_t - self._ext(self.sqrt_one_minus_alphas_cumprod, t, x_t) * v

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return self._ext(self.sqrt_alphas_cumprod, t, x_t) * v + self._ext(self.sqrt_one_minus_alphas_cumprod, t, x_t) * x_t

    def apply_model(self, x, t, c, **kw):
        scale = (0.5 + t.float() / 2000.0).reshape((x.shape[0],) + (1,) * (x.dim() - 1))
        return torch.tanh(0.7 * x.float() + c["bias"].to(x.device).float()) * scale


def tensors(device="cpu"):
    g = torch.Generator().manual_seed(SEED + 5)
    r = lambda *s: torch.randn(*s, generator=g)
    full = (BATCH,) + SHAPE
    t = dict(x_T=r(*full), x0=r(*full), origin=r(*full), scene=r(*full), q_noises=[r(*full) for _ in range(3 * STEPS)])
    t["mask"] = (torch.rand(full, generator=g) < 0.4).float()
    t["scene_mask"] = (torch.rand((BATCH, 1) + SHAPE[1:], generator=g) < 0.5).float()
    t["cond_frame_index"] = torch.tensor([1, 4])
    mv = lambda v: [u.to(device) for u in v] if isinstance(v, list) else v.to(device)
    return {k: mv(v) for k, v in t.items()}


CASES = {
    # name: (model kwargs, sampler kwargs builder)
    "mask_noised": (dict(), lambda t: dict(mask=t["mask"], x0=t["x0"])),
    "mask_clean": (dict(), lambda t: dict(mask=t["mask"], x0=t["x0"], clean_cond=True)),
    "paste_overlap": (dict(), lambda t: dict(paste_overlap_frames=True, num_overlap=2)),
    "noise_shaping": (dict(), lambda t: dict(noise_shaping=True, noise_shaping_minimum_timesteps=400, scene_mask=t["scene_mask"])),
    "noise_shaping_scene": (dict(), lambda t: dict(noise_shaping=True, noise_shaping_minimum_timesteps=0, scene_mask=t["scene_mask"],
                                                   scene_frames=t["scene"])),
    "paste_cond_frame": (dict(), lambda t: dict(paste_cond_frame=True)),
    "v_param": (dict(parameterization="v"), lambda t: dict()),
    "dynamic_rescale": (dict(dynamic_rescale=True), lambda t: dict()),
    "plain": (dict(), lambda t: dict()),
}


def run_case(name, sampler_cls, device="cpu", **sampler_extra):
    """-> final latents of case `name` through `sampler_cls` (the reference's DDIMSampler subclass or ours)."""
    mkw, skw = CASES[name]
    t = tensors(device)
    model = DuckModel(device, q_noises=t["q_noises"], **mkw)
    dummy = [torch.zeros(BATCH, 1, device=device)]      # (the reference's sample() reads conditioning[first key][0].shape[0])
    cond = dict(c_crossattn=dummy, bias=torch.tensor([0.3], device=device), origin_z_0=t["origin"], c_cond_frame_index=t["cond_frame_index"])
    uncond = dict(c_crossattn=dummy, bias=torch.tensor([-0.2], device=device), origin_z_0=t["origin"], c_cond_frame_index=t["cond_frame_index"])
    sampler = sampler_cls(model)
    out, _ = sampler.sample(STEPS, BATCH, SHAPE, cond, eta=0.0, x_T=t["x_T"], verbose=False, unconditional_guidance_scale=SCALE,
                            unconditional_conditioning=uncond, timestep_spacing="uniform_trailing", guidance_rescale=RESCALE,
                            **skw(t), **sampler_extra)
    return out
