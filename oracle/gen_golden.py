#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  This script imports the reference's Python from
/root/reference/CamContextI2V (read-only) with in-process stand-ins for three
modules that the reference imports at file top but never touches on this path
(cv2, pytorch_lightning, torchvision.utils.make_grid), runs its UNet (with the
instance-level camera patch of model/camcontexti2v.py:111-170 replayed on a bare
UNetModel), its DDIMSampler and its epipolar geometry on seeded inputs, and
stores inputs + outputs as small fp32 fixtures.  Fixtures hold tensors and
scalars only; no reference source travels.

Usage:  python oracle/gen_golden.py [--out tests/golden]
The GPU box never runs this (it has no /root/reference).
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference/CamContextI2V"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_sibling(name):
    """Import oracle/<name>.py by file path.  The repo root must NOT be on sys.path here: it holds the
    product's own ``lvdm`` / ``model`` / ``utils`` packages, which would shadow the reference's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(f"_ccv_oracle_{name}", os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_gi = _load_sibling("golden_inputs")
SMALL_CFG, MEDIUM_CFG, FULL_CFG, SEED = _gi.SMALL_CFG, _gi.MEDIUM_CFG, _gi.FULL_CFG, _gi.SEED
small_inputs, medium_inputs, checksum = _gi.small_inputs, _gi.medium_inputs, _gi.checksum



def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mod("cv2")
    plu = mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    mod("pytorch_lightning", LightningModule=nn.Module, utilities=plu)
    tvu = mod("torchvision.utils", make_grid=lambda *a, **k: None)
    mod("torchvision", utils=tvu)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def build_reference_unet(cfg, camera=True, device="cpu", origin=64):
    """Reference UNetModel + the camera patch loop of CamContextI2V.__init__ replayed on it."""
    from lvdm.modules.networks.openaimodel3d import UNetModel
    from model.modules import modified_forwards as mf
    from model.modules.epipolar import Epipolar
    for m_ in (sys.modules["lvdm.modules.networks.openaimodel3d"], mf):
        assert m_.__file__.startswith(REF), f"{m_.__name__} was not imported from the reference: {m_.__file__}"

    with torch.device(device):
        unet = UNetModel(**cfg)
        if camera:
            unet.forward = mf.new_forward_for_unet.__get__(unet, unet.__class__)
            init_inner = unet.init_attn[0].proj_in.out_channels
            for _, m in unet.named_modules():
                cls = m.__class__.__name__
                if cls == "TemporalTransformer":
                    m.forward = mf.new_forward_for_TemporalTransformer.__get__(m, m.__class__)
                elif cls == "TimestepEmbedSequential":
                    m.forward = mf.new_forward_for_TimestepEmbedSequential.__get__(m, m.__class__)
                elif cls == "BasicTransformerBlock":
                    dim = m.attn1.to_k.in_features
                    if m.context_dim is None and dim != init_inner:
                        m.forward = mf.new_forward_for_BasicTransformerBlock_of_TemporalTransformer.__get__(m, m.__class__)
                        m._forward = mf.new__forward_for_BasicTransformerBlock_of_TemporalTransformer.__get__(m, m.__class__)
                        m.add_module("pluker_projection", nn.Linear(dim, dim))
                        m.add_module("epipolar", Epipolar(
                            query_dim=dim, context_dim=dim, heads=m.attn1.heads, origin_h=origin, origin_w=origin,
                            is_3d_full_attn=False, num_register_tokens=4,
                            attention_resolution=[8, 4, 2, 1], compression_factor=1))
    return unet.eval()


def manifest_of(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def geometry_via_reference(K, w2c, cond_idx, H_px, W_px, noise):
    """Call the reference's geometry methods unbound on a stub `self`."""
    import model.camcontexti2v as cc
    from model.base import CameraControlLVDM

    stub = types.SimpleNamespace()
    stub.epipolar_config = types.SimpleNamespace(
        apply_epipolar_soft_mask=False, epipolar_hybrid_attention=False,
        epipolar_hybrid_attention_v2=False, only_self_pixel_on_current_frame=False,
        current_frame_as_register_token=False)
    C = cc.CamContextI2V
    c2w = w2c.inverse()
    rel = CameraControlLVDM.get_relative_pose(stub, c2w, cond_idx, mode="left", normalize_T0=False)
    pairs = C.get_relative_c2w_RT_pairs(stub, rel)
    R, t = pairs[..., :3, :3], pairs[..., :3, 3:4]
    # add_small_perturbation draws torch.randn_like(t): reproduce that draw from `noise`
    zero = (t.abs() < 1e-6).all(dim=-2, keepdim=True)
    t = torch.where(zero, noise * 1e-6, t)
    F = C.get_fundamental_matrix(stub, K.unsqueeze(1), R, t)
    T = w2c.shape[1]
    masks = {d: C.get_epipolar_mask(stub, F, T, H_px // d, W_px // d, d) for d in (8, 16, 32, 64)}
    return rel, F, masks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    _install_shims()
    torch.manual_seed(SEED)
    torch.set_grad_enabled(False)

    seeded_state_dict = _load_sibling("unet_oracle").seeded_state_dict
    _geo = _load_sibling("geometry_oracle")
    synthetic_trajectory = _geo.synthetic_trajectory
    repo_root = os.path.dirname(HERE)
    assert all(os.path.abspath(p or os.getcwd()) != repo_root for p in sys.path), \
        "run as `python oracle/gen_golden.py`: the repo root on sys.path would shadow the reference's packages"

    # ---- (6) full-config key -> shape manifest (meta device, nothing allocated) --------
    full = build_reference_unet(FULL_CFG, camera=True, device="meta")
    with open(os.path.join(args.out, "unet_full_manifest.json"), "w") as f:
        json.dump(manifest_of(full), f, indent=0, sort_keys=True)
    print("full manifest:", len(manifest_of(full)), "tensors,",
          sum(int(np.prod(s)) for s in manifest_of(full).values()), "params")
    del full

    # ---- (2) geometry: 16-frame trajectory, 64x64 px and popcounts at 256x256 ----------
    b, T = 1, 16
    w2c = synthetic_trajectory(b, T)
    g = torch.Generator().manual_seed(SEED)
    pnoise = torch.randn(b, T, T, 3, 1, generator=g)
    cond_idx = torch.zeros(b, dtype=torch.long)
    geo = {}
    for px, fx in ((64, 32.0), (256, 128.0)):
        K = torch.tensor([[fx, 0, px / 2], [0, fx, px / 2], [0, 0, 1.0]]).repeat(b, T, 1, 1)
        rel, F, masks = geometry_via_reference(K, w2c, cond_idx, px, px, pnoise)
        geo[px] = (K, rel, F, masks)
    K64, rel64, F64, masks64 = geo[64]
    out = dict(w2c=w2c.numpy(), perturb_noise=pnoise.numpy(), K64=K64.numpy(), rel64=rel64.numpy(), F64=F64.numpy())
    for d, m in masks64.items():
        out[f"mask64_d{d}_bits"] = np.packbits(m.numpy().astype(np.uint8), axis=-1, bitorder="little")
        out[f"mask64_d{d}_shape"] = np.array(m.shape)
    K256, _, F256, masks256 = geo[256]
    out["K256"] = K256.numpy()
    out["F256"] = F256.numpy()
    for d, m in masks256.items():
        out[f"mask256_d{d}_popcount_rows"] = m.sum(-1).to(torch.int32).numpy()  # [b, L] per-query counts
    np.savez_compressed(os.path.join(args.out, "geometry.npz"), **out)
    print("geometry: densities @256:", {d: float(m.float().mean()) for d, m in masks256.items()})

    # ---- (3) schedule tables + (4) one CFG step, through the reference sampler ---------
    from lvdm.models.samplers.ddim import DDIMSampler
    from lvdm.models.utils_diffusion import make_beta_schedule

    class CpuSampler(DDIMSampler):
        def register_buffer(self, name, attr):  # reference hard-codes cuda (ddim.py:18-22)
            setattr(self, name, attr)

    betas_np = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas_np, axis=0)
    to32 = lambda a: torch.tensor(a, dtype=torch.float32)

    class DuckModel:
        num_timesteps = 1000
        device = torch.device("cpu")
        use_dynamic_rescale = False
        parameterization = "eps"
        betas = to32(betas_np)
        alphas_cumprod = to32(ac)
        alphas_cumprod_prev = to32(np.append(1.0, ac[:-1]))

        def __init__(self, fn):
            self.fn = fn

        def apply_model(self, x, t, c, **kw):
            return self.fn(x, t, c, **kw)

    tabs = {}
    for eta in (0.0, 1.0):
        s = CpuSampler(DuckModel(None))
        s.make_schedule(25, ddim_discretize="uniform_trailing", ddim_eta=eta, verbose=False)
        tag = f"eta{int(eta)}"
        tabs[f"timesteps_{tag}"] = np.asarray(s.ddim_timesteps)
        # what p_sample_ddim actually uses: torch.full(size, table[index]) -> fp32
        for name in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
            tab = getattr(s, name)
            tabs[f"{name}_{tag}"] = np.array([torch.full((1,), tab[i]).item() for i in range(25)], dtype=np.float32)
    s50 = CpuSampler(DuckModel(None))
    s50.make_schedule(50, ddim_discretize="uniform", ddim_eta=0.0, verbose=False)
    tabs["timesteps_uniform50"] = np.asarray(s50.ddim_timesteps)
    tabs["alphas_cumprod"] = ac.astype(np.float32)

    shape = (2, 4, 16, 8, 8)
    g = torch.Generator().manual_seed(SEED + 1)
    x = torch.randn(shape, generator=g)
    e_c = torch.randn(shape, generator=g)
    e_uc = torch.randn(shape, generator=g) * 0.9 + 0.1 * e_c
    cond, uncond = {"tag": "c"}, {"tag": "uc"}
    duck = DuckModel(lambda x_, t_, c_, **kw: e_c if c_["tag"] == "c" else e_uc)
    s = CpuSampler(duck)
    s.make_schedule(25, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    for index in (24, 7, 0):
        torch.manual_seed(1000 + index)
        ts = torch.full((2,), int(s.ddim_timesteps[index]), dtype=torch.long)
        x_prev, pred_x0 = s.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=7.5,
                                          unconditional_conditioning=uncond, guidance_rescale=0.7)
        torch.manual_seed(1000 + index)
        z = torch.randn(shape)
        tabs[f"step{index}_x_prev"] = x_prev.numpy()
        tabs[f"step{index}_pred_x0"] = pred_x0.numpy()
        tabs[f"step{index}_noise"] = z.numpy()
    # no-guidance branch (scale 1.0)
    torch.manual_seed(77)
    x_prev, pred_x0 = s.p_sample_ddim(x, cond, ts, index=3, unconditional_guidance_scale=1.0)
    torch.manual_seed(77)
    tabs["noguid_noise"] = torch.randn(shape).numpy()
    tabs["noguid_x_prev"] = x_prev.numpy()
    tabs["step_x"], tabs["step_e_c"], tabs["step_e_uc"] = x.numpy(), e_c.numpy(), e_uc.numpy()
    np.savez_compressed(os.path.join(args.out, "ddim.npz"), **tabs)
    print("ddim timesteps:", tabs["timesteps_eta1"])

    # ---- (5) reduced-width UNet, same topology: forwards + 3-step DDIM trajectory -------
    unet = build_reference_unet(SMALL_CFG, camera=True)
    man = manifest_of(unet)
    with open(os.path.join(args.out, "unet_small_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    sd = seeded_state_dict(man, SEED)
    unet.load_state_dict(sd, strict=True)

    b, T, hl = 2, 16, 8
    inp = small_inputs(b, T, hl)
    xin, tt, fs, ctx_pf, ctx_rep, feats = inp["x"], inp["t"], inp["fs"], inp["ctx_pf"], inp["ctx_rep"], inp["feats"]
    masks_b = {d: m.expand(b, -1, -1).contiguous() for d, m in masks64.items()}
    # give batch element 1 a different mask: transpose of element 0 (still a valid bool mask)
    for d in masks_b:
        masks_b[d][1] = masks_b[d][0].t()
    cam = dict(pluker_embedding_features=feats, sample_locs_dict=masks_b,
               cond_frame_index=torch.zeros(b, dtype=torch.long), add_type="add_to_main_branch")
    cam_other = dict(cam, add_type="add_into_temporal_attn")
    cam_nomask = dict(cam, sample_locs_dict=None)

    res = dict(x=xin.numpy(), t=tt.numpy(), fs=fs.numpy(), seed=np.array(SEED),
               ctx_pf_checksum=np.array(checksum(ctx_pf)), ctx_rep_checksum=np.array(checksum(ctx_rep)),
               feat_checksum=np.array([checksum(f_) for f_ in feats]))
    for d, m in masks_b.items():
        res[f"mask_d{d}_bits"] = np.packbits(m.numpy().astype(np.uint8), axis=-1, bitorder="little")
    res["y_nocam_pf"] = unet(xin, tt, context=ctx_pf, fs=fs, camera_condition=None).numpy()
    res["y_cam_rep"] = unet(xin, tt, context=ctx_rep, fs=fs, camera_condition=cam).numpy()
    res["y_cam_pf"] = unet(xin, tt, context=ctx_pf, fs=fs, camera_condition=cam).numpy()
    res["y_cam_other_addtype"] = unet(xin[:1], tt[:1], context=ctx_rep[:1], fs=fs[:1], camera_condition={
        **cam_other, "pluker_embedding_features": [f_[:1] for f_ in feats],
        "sample_locs_dict": {d: m[:1] for d, m in masks_b.items()}}).numpy()
    res["y_cam_nomask"] = unet(xin[:1], tt[:1], context=ctx_rep[:1], fs=fs[:1], camera_condition={
        **cam_nomask, "pluker_embedding_features": [f_[:1] for f_ in feats]}).numpy()
    res["y_default_fs"] = unet(xin[:1], tt[:1], context=ctx_pf[:1], fs=None, camera_condition=None).numpy()
    for k in list(res):
        if k.startswith("y_"):
            print(k, "absmax", float(np.abs(res[k]).max()), "std", float(res[k].std()))

    # 3-step DDIM, CFG 7.5, rescale 0.7, eta 1 with the RNG stream recorded
    def apply(x_, t_, c_, **kw):
        return unet(torch.cat([x_, c_["c_concat"][0]], 1), t_, context=c_["c_crossattn"][0], fs=kw.get("fs"),
                    camera_condition=c_.get("camera_condition"))

    duck = DuckModel(apply)
    s = CpuSampler(duck)
    c_concat, x_T = inp["c_concat"], inp["x_T"]
    cond = dict(c_concat=[c_concat], c_crossattn=[ctx_rep], camera_condition=cam)
    uncond = dict(c_concat=[c_concat], c_crossattn=[ctx_pf])
    torch.manual_seed(4242)
    samples, _ = s.sample(3, b, (4, T, hl, hl), cond, eta=1.0, x_T=x_T, verbose=False,
                          unconditional_guidance_scale=7.5, unconditional_conditioning=uncond,
                          timestep_spacing="uniform_trailing", guidance_rescale=0.7, fs=fs,
                          enable_camera_condition=True)
    torch.manual_seed(4242)
    noises = [torch.randn(b, 4, T, hl, hl) for _ in range(3)]
    res["traj_c_concat"], res["traj_x_T"] = c_concat.numpy(), x_T.numpy()
    res["traj_noises"] = torch.stack(noises).numpy()
    res["traj_x0"] = samples.numpy()
    assert "camera_condition" in uncond and uncond["camera_condition"]["is_uc"] is True
    np.savez_compressed(os.path.join(args.out, "unet_small.npz"), **res)
    del unet

    # ---- (5b) medium-width UNet (model_channels 128, 16x16 latents): tight-tolerance parity fixture -------
    epipolar_mask = _geo.epipolar_mask
    unet = build_reference_unet(MEDIUM_CFG, camera=True, origin=128)
    man = manifest_of(unet)
    with open(os.path.join(args.out, "unet_medium_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    unet.load_state_dict(seeded_state_dict(man, SEED), strict=True)
    inp = medium_inputs()
    K128 = torch.tensor([[64.0, 0, 64], [0, 64.0, 64], [0, 0, 1.0]]).repeat(1, 16, 1, 1)
    _, F128, masks128 = geometry_via_reference(K128, w2c, cond_idx, 128, 128, pnoise)
    for d, m in masks128.items():   # the committed fixture carries F only; the oracle rebuilds the masks bit-exactly
        assert torch.equal(m, epipolar_mask(F128, 128 // d, 128 // d, d)), d
    cam = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks128,
               cond_frame_index=torch.zeros(1, dtype=torch.long), add_type="add_to_main_branch")
    med = dict(F128=F128.numpy(), x=inp["x"].numpy(), t=inp["t"].numpy(), fs=inp["fs"].numpy(),
               ctx_pf_checksum=np.array(checksum(inp["ctx_pf"])), ctx_rep_checksum=np.array(checksum(inp["ctx_rep"])),
               mask_popcount=np.array([int(masks128[d].sum()) for d in (8, 16, 32, 64)]))
    med["y_cam_rep"] = unet(inp["x"], inp["t"], context=inp["ctx_rep"], fs=inp["fs"], camera_condition=cam).numpy()
    med["y_cam_pf"] = unet(inp["x"], inp["t"], context=inp["ctx_pf"], fs=inp["fs"], camera_condition=cam).numpy()
    med["y_nocam_pf"] = unet(inp["x"], inp["t"], context=inp["ctx_pf"], fs=inp["fs"], camera_condition=None).numpy()
    for k in ("y_cam_rep", "y_cam_pf", "y_nocam_pf"):
        print("medium", k, "absmax", float(np.abs(med[k]).max()), "std", float(med[k].std()))
    np.savez_compressed(os.path.join(args.out, "unet_medium.npz"), **med)
    print("wrote", args.out, {f: os.path.getsize(os.path.join(args.out, f)) for f in sorted(os.listdir(args.out))})


if __name__ == "__main__":
    main()
