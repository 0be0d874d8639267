#!/usr/bin/env python
"""Generate tests/golden/ddim_camera_cfg.npz by RUNNING THE REFERENCE's ``DDIMSampler.p_sample_ddim`` with camera
guidance (camera_cfg != 1, the optional third forward: lvdm/models/samplers/ddim.py:268-280; build container only).
TEST INFRASTRUCTURE, companion of oracle/gen_golden.py section (4): the model is a duck-typed object whose apply_model
returns supplied tensors (conditional / unconditional / conditional without camera), so the fixture pins the guidance
arithmetic, both weight schedulers, the std rescale and the update.  Fixtures hold tensors and scalars only.
The 'cosine' scheduler cases run at batch 1: the reference reshapes its weight to [b, 1, 1, 1] (ddim.py:275), which only
broadcasts against the 5-D video latents when b == 1 (the interactive demo's case).

Usage:  cd oracle && python gen_golden_camcfg.py [--out ../tests/golden]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/CamContextI2V"


def _load(name):
    spec = importlib.util.spec_from_file_location(f"_ccv_oracle_{name}", os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    repo_root = os.path.dirname(HERE)
    assert all(os.path.abspath(p or os.getcwd()) != repo_root for p in sys.path), \
        "run from oracle/: the repo root on sys.path would shadow the reference's packages"
    gg = _load("gen_golden")
    gg._install_shims()
    do = _load("ddim_oracle")
    torch.set_grad_enabled(False)

    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.utils_diffusion import make_beta_schedule
    assert sys.modules["lvdm.models.samplers.ddim"].__file__.startswith(REF)

    class CpuSampler(DDIMSampler):
        def register_buffer(self, name, attr):  # reference hard-codes cuda (ddim.py:18-22)
            setattr(self, name, attr)

    betas_np = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas_np, axis=0)
    to32 = lambda a: torch.tensor(a, dtype=torch.float32)

    shape = (2, 4, 16, 8, 8)
    g = torch.Generator().manual_seed(gg.SEED + 41)
    x = torch.randn(shape, generator=g)
    e_c = torch.randn(shape, generator=g)
    e_uc = torch.randn(shape, generator=g) * 0.9 + 0.1 * e_c
    e_nc = 0.7 * e_c + 0.5 * torch.randn(shape, generator=g)

    class DuckModel:
        num_timesteps = 1000
        device = torch.device("cpu")
        use_dynamic_rescale = False
        parameterization = "eps"
        betas = to32(betas_np)
        alphas_cumprod = to32(ac)
        alphas_cumprod_prev = to32(np.append(1.0, ac[:-1]))
        calls = []

        def apply_model(self, x_, t_, c_, **kw):
            cam = c_.get("camera_condition")
            which = "nc" if cam is None else ("uc" if cam.get("is_uc") else "c")
            assert (c_["tag"] == "uc") == (which == "uc")
            self.calls.append(which)
            return {"c": e_c, "uc": e_uc, "nc": e_nc}[which][:x_.shape[0]]

    duck = DuckModel()
    s = CpuSampler(duck)
    s.make_schedule(25, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    tab = do.ddim_tables(25, 1.0)
    out = dict(x=x.numpy(), e_c=e_c.numpy(), e_uc=e_uc.numpy(), e_nc=e_nc.numpy(), scale=np.float32(7.5), rescale=np.float32(0.7))
    for scheduler, camera_cfg, index in (("constant", 2.0, 20), ("cosine", 1.5, 20), ("cosine", 3.0, 2)):
        nb = 1 if scheduler == "cosine" else 2
        cond = {"tag": "c", "camera_condition": {"cond_frame_index": torch.zeros(nb, dtype=torch.long)}}
        uncond = {"tag": "uc"}
        ts = torch.full((nb,), int(s.ddim_timesteps[index]), dtype=torch.long)
        duck.calls.clear()
        torch.manual_seed(2000 + index)
        x_prev, pred_x0 = s.p_sample_ddim(x[:nb], cond, ts, index=index, unconditional_guidance_scale=7.5,
                                          unconditional_conditioning=uncond, guidance_rescale=0.7, enable_camera_condition=True,
                                          camera_cfg=camera_cfg, camera_cfg_scheduler=scheduler)
        assert duck.calls == ["c", "uc", "nc"], duck.calls
        torch.manual_seed(2000 + index)
        z = torch.randn((nb, *shape[1:]))
        w = do.camera_cfg_weight(ts, scheduler)
        xo, x0o, _ = do.cfg_ddim_update(x[:nb], e_c[:nb], e_uc[:nb], z, tab["alphas"][index], tab["alphas_prev"][index], tab["sigmas"][index],
                                        tab["sqrt_one_minus_alphas"][index], 7.5, 0.7, e_nc=e_nc[:nb], camera_cfg=camera_cfg, camera_weight=w)
        err = max((xo - x_prev).abs().max().item(), (x0o - pred_x0).abs().max().item())
        assert err < 2e-5 * pred_x0.abs().max().item(), err
        tag = f"{scheduler}_{camera_cfg:g}_{index}"
        out[f"{tag}_x_prev"], out[f"{tag}_pred_x0"], out[f"{tag}_noise"] = x_prev.numpy(), pred_x0.numpy(), z.numpy()
        out[f"{tag}_t"] = ts.numpy()
        print(f"camera cfg {tag}: oracle max abs err {err:.2e}, pred_x0 absmax {pred_x0.abs().max().item():.2f}")
    # camera_cfg is ignored without enable_camera_condition (ddim.py:268)
    duck.calls.clear()
    torch.manual_seed(2100)
    ts = torch.full((2,), int(s.ddim_timesteps[2]), dtype=torch.long)
    x_prev, _ = s.p_sample_ddim(x, {"tag": "c", "camera_condition": {}}, ts, index=2, unconditional_guidance_scale=7.5,
                                unconditional_conditioning={"tag": "uc", "camera_condition": {"is_uc": True}}, guidance_rescale=0.7,
                                camera_cfg=3.0)
    assert duck.calls == ["c", "uc"]
    torch.manual_seed(2100)
    out["disabled_noise"], out["disabled_x_prev"] = torch.randn(shape).numpy(), x_prev.numpy()
    np.savez_compressed(os.path.join(args.out, "ddim_camera_cfg.npz"), **out)


if __name__ == "__main__":
    main()
