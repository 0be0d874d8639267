"""End-to-end parity of the HIP UNet / sampler on MI355X against (a) fixtures produced by RUNNING THE
REFERENCE (tests/golden/unet_small.npz) and (b) the CPU oracle on the same seeded weights.

Stated tolerance per UNet forward (bf16 GEMM/attention operands; fp32 accumulation, residual stream,
norms and softmax), on the seeded N(0, 0.02) weights of the fixtures:
    relative L2 error <= 7e-2   and   max |err| <= 1.2e-1 * max |ref|.
Why this loose: the randomly initialised network amplifies rounding noise ~40x.  The CPU oracle run with
every matmul operand rounded to bf16 (oracle.unet_oracle.operand_rounding) deviates from the fp32 reference
by rel-L2 4.2e-2 on the same inputs, the HIP path by 4.6e-2; `test_error_is_bf16_rounding_not_implementation`
asserts that the HIP error stays within 1.6x of that emulated floor, and tests/test_ops_gpu.py pins every
kernel separately to ~1e-2 of the output range.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

REL_L2, MAX_REL = 7e-2, 1.2e-1


def _check(got, ref, what, rel_l2=REL_L2, max_rel=MAX_REL):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    l2 = ((got - ref).norm() / ref.norm()).item()
    mx = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"[parity] {what}: rel_l2={l2:.3e} max_rel={mx:.3e}")
    assert l2 <= rel_l2 and mx <= max_rel, f"{what}: rel_l2={l2:.3e} (tol {rel_l2}), max_rel={mx:.3e} (tol {max_rel})"


def _unbits(bits, L):
    return torch.from_numpy(np.unpackbits(bits, axis=-1, bitorder="little")[..., :L].astype(bool))


@pytest.fixture(scope="module")
def small(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.golden_inputs import SEED, SMALL_CFG, small_inputs
    from oracle.unet_oracle import seeded_state_dict
    from utils.utils import instantiate_from_config
    dev = torch.device("cuda:0")
    fx = dict(np.load(os.path.join(golden_dir, "unet_small.npz")))
    man = json.load(open(os.path.join(golden_dir, "unet_small_manifest.json")))
    sd = seeded_state_dict(man, SEED)
    unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG})
    unet.enable_camera_conditioning(dict(origin_h=64, origin_w=64, is_3d_full_attn=False, num_register_tokens=4,
                                         attention_resolution=[8, 4, 2, 1], compression_factor=1))
    unet.load_state_dict(sd, strict=True)   # reference checkpoint layout, strict
    unet = unet.to(dev).eval()
    inp = small_inputs()
    L = {8: 16 * 64, 16: 16 * 16, 32: 16 * 4, 64: 16}
    masks = {d: _unbits(fx[f"mask_d{d}_bits"], L[d]) for d in L}
    to = lambda t: t.to(dev)
    cam = dict(pluker_embedding_features=[to(f) for f in inp["feats"]],
               sample_locs_dict={d: to(m) for d, m in masks.items()},
               cond_frame_index=torch.zeros(2, dtype=torch.long, device=dev), add_type="add_to_main_branch")
    gin = {k: (to(v) if torch.is_tensor(v) else v) for k, v in inp.items()}
    return unet, fx, sd, inp, gin, cam, masks


def test_no_camera_per_frame_context(small):
    unet, fx, _, _, g, _, _ = small
    y = unet(g["x"], g["t"], context=g["ctx_pf"], fs=g["fs"], camera_condition=None)
    _check(y, fx["y_nocam_pf"], "no camera, per-frame ctx vs reference fixture")


def test_error_is_bf16_rounding_not_implementation(small):
    """HIP-vs-fp32 error must be explained by bf16 operand rounding: compare with the oracle run under the same
    arithmetic contract (operands rounded to bf16, fp32 accumulate)."""
    from oracle import unet_oracle as uo
    from oracle.golden_inputs import SMALL_CFG
    unet, fx, sd, inp, g, cam, masks = small
    ref = torch.from_numpy(fx["y_cam_rep"])
    cam_cpu = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks, add_type="add_to_main_branch")
    with uo.operand_rounding(torch.bfloat16):
        emu = uo.unet_forward(sd, SMALL_CFG, inp["x"], inp["t"], inp["ctx_rep"], inp["fs"], cam_cpu)
    y = unet(g["x"], g["t"], context=g["ctx_rep"], fs=g["fs"], camera_condition=cam).cpu()
    e_emu = ((emu - ref).norm() / ref.norm()).item()
    e_hip = ((y - ref).norm() / ref.norm()).item()
    print(f"[parity] bf16-emulated oracle rel_l2={e_emu:.3e}; HIP rel_l2={e_hip:.3e}")
    assert e_hip <= 1.6 * e_emu + 2e-3


def test_camera_repeat_context(small):
    unet, fx, _, _, g, cam, _ = small
    y = unet(g["x"], g["t"], context=g["ctx_rep"], fs=g["fs"], camera_condition=cam, enable_camera_condition=True, split="val")
    _check(y, fx["y_cam_rep"], "camera, repeated ctx vs reference fixture")


def test_camera_per_frame_context(small):
    unet, fx, _, _, g, cam, _ = small
    y = unet(g["x"], g["t"], context=g["ctx_pf"], fs=g["fs"], camera_condition=cam)
    _check(y, fx["y_cam_pf"], "camera, per-frame ctx vs reference fixture")


def test_other_add_type_nomask_default_fs(small):
    unet, fx, _, _, g, cam, _ = small
    one = lambda t: t[:1].contiguous()
    cam1 = dict(cam, pluker_embedding_features=[one(f) for f in cam["pluker_embedding_features"]],
                sample_locs_dict={d: one(m) for d, m in cam["sample_locs_dict"].items()})
    y = unet(one(g["x"]), one(g["t"]), context=one(g["ctx_rep"]), fs=one(g["fs"]),
             camera_condition=dict(cam1, add_type="add_into_temporal_attn"))
    _check(y, fx["y_cam_other_addtype"], "other add_type vs reference fixture")
    y = unet(one(g["x"]), one(g["t"]), context=one(g["ctx_rep"]), fs=one(g["fs"]),
             camera_condition=dict(cam1, sample_locs_dict=None))
    _check(y, fx["y_cam_nomask"], "camera without masks vs reference fixture")
    y = unet(one(g["x"]), one(g["t"]), context=one(g["ctx_pf"]), fs=None, camera_condition=None)
    _check(y, fx["y_default_fs"], "default fs vs reference fixture")


def test_cfg_pair_equals_two_forwards(small):
    """One 2b-batch forward with a list of contexts == two separate forwards (bit exact per half is not
    required: tile selection may differ with M; same tolerance as everything else, in practice ~1e-3)."""
    unet, fx, _, _, g, cam, _ = small
    y_c = unet(g["x"], g["t"], context=g["ctx_rep"], fs=g["fs"], camera_condition=cam)
    y_u = unet(g["x"], g["t"], context=g["ctx_pf"], fs=g["fs"], camera_condition=cam)
    x2 = torch.cat([g["x"], g["x"]], 0)
    y2 = unet(x2, torch.cat([g["t"], g["t"]]), context=[g["ctx_rep"], g["ctx_pf"]], fs=torch.cat([g["fs"], g["fs"]]),
              camera_condition=cam)
    _check(y2[:2], y_c, "pair/cond half")     # this fixture amplifies any rounding difference ~1e4x (see header)
    _check(y2[2:], y_u, "pair/uncond half")
    _check(y2[:2], fx["y_cam_rep"], "pair/cond half vs reference fixture")
    # shared-input form (what apply_model_pair uses when both halves start from the same tensors): the context-free
    # head of the UNet runs once on b samples and is duplicated in front of the first transformer
    y3 = unet(g["x"], g["t"], context=[g["ctx_rep"], g["ctx_pf"]], fs=g["fs"], camera_condition=cam, cfg_shared_input=True)
    assert y3.shape == y2.shape
    _check(y3[:2], y_c, "shared-input pair/cond half")
    _check(y3[2:], y_u, "shared-input pair/uncond half")
    _check(y3[:2], fx["y_cam_rep"], "shared-input pair/cond half vs reference fixture")


def test_native_packed_masks_equal_bool_masks(small):
    """camera_condition['sample_locs_packed'] (bit-packed masks built on the GPU) == the bool-mask path."""
    from camc2v_amd import ops
    unet, fx, _, _, g, cam, _ = small
    packed = {d: ops.pack_mask(m) for d, m in cam["sample_locs_dict"].items()}
    cam_p = dict(cam, sample_locs_dict=None, sample_locs_packed=packed)
    y = unet(g["x"], g["t"], context=g["ctx_rep"], fs=g["fs"], camera_condition=cam_p)
    _check(y, fx["y_cam_rep"], "packed masks vs reference fixture")


def test_ddim_three_steps_vs_reference_fixture(small):
    """3 DDIM steps, CFG 7.5, rescale 0.7, eta 1 with the reference's noise draws injected."""
    from camc2v_amd.diffusion import LatentDiffusionCore
    from oracle.golden_inputs import SMALL_CFG
    unet, fx, sd, _, g, cam, _ = small
    core = LatentDiffusionCore({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG},
                               linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4,
                               image_size=[8, 8], temporal_length=16)
    core.model.diffusion_model = unet          # reuse the loaded network
    core = core.to("cuda:0")
    cond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_rep"]], camera_condition=cam)
    uncond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_pf"]])
    noises = [torch.from_numpy(n) for n in fx["traj_noises"]]
    samples, _ = core.sample_log(cond, 2, True, 3, eta=1.0, x_T=torch.from_numpy(fx["traj_x_T"]),
                                 unconditional_guidance_scale=7.5, unconditional_conditioning=uncond,
                                 timestep_spacing="uniform_trailing", guidance_rescale=0.7, fs=g["fs"],
                                 enable_camera_condition=True, injected_noise=noises)
    assert uncond["camera_condition"]["is_uc"] is True
    _check(samples, fx["traj_x0"], "3-step DDIM trajectory vs reference fixture", 1e-1, 2e-1)


def test_graph_replay_equals_eager(small):
    """hipGraph-captured DDIM step == eager launches, bit for bit (same kernels, same order)."""
    from camc2v_amd.diffusion import LatentDiffusionCore
    from oracle.golden_inputs import SMALL_CFG
    unet, fx, sd, _, g, cam, _ = small
    core = LatentDiffusionCore({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG},
                               linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4,
                               image_size=[8, 8], temporal_length=16)
    core.model.diffusion_model = unet
    core = core.to("cuda:0")
    cond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_rep"]], camera_condition=cam)
    noises = [torch.from_numpy(n).cuda() for n in fx["traj_noises"]] + [torch.from_numpy(fx["traj_noises"][0]).cuda()]
    outs = []
    for use_graph in (False, True, True):
        uncond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_pf"]])
        if use_graph and outs[1:]:
            uncond = outs_uncond          # same tensors -> cached graph is replayed
        outs_uncond = uncond
        s, _ = core.sample_log(cond, 2, True, 4, eta=1.0, x_T=torch.from_numpy(fx["traj_x_T"]).cuda(),
                               unconditional_guidance_scale=7.5, unconditional_conditioning=uncond,
                               timestep_spacing="uniform_trailing", guidance_rescale=0.7, fs=g["fs"],
                               enable_camera_condition=True, injected_noise=noises, use_graph=use_graph)
        outs.append(s.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert len(core.__dict__["_ccv_graph_cache"]) == 1


def test_medium_fixture_tight_tolerance(golden_dir):
    """Well-conditioned fixture (model_channels 128, 16x16 latents, camera + masks from the reference's F):
    HIP vs the reference's fp32 output within rel-L2 2.5e-2 / max 5e-2, within 1.6x of the bf16-operand
    emulated oracle, and bitwise reproducible run to run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import geometry_oracle, unet_oracle as uo
    from oracle.golden_inputs import MEDIUM_CFG, SEED, medium_inputs
    from utils.utils import instantiate_from_config
    fx = np.load(os.path.join(golden_dir, "unet_medium.npz"))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    sd = uo.seeded_state_dict(man, SEED)
    unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": MEDIUM_CFG})
    unet.enable_camera_conditioning(dict(origin_h=128, origin_w=128, is_3d_full_attn=False, num_register_tokens=4,
                                         attention_resolution=[8, 4, 2, 1], compression_factor=1))
    unet.epipolar_origin_h = 128
    unet.load_state_dict(sd, strict=True)
    unet = unet.cuda().eval()
    inp = medium_inputs()
    F = torch.from_numpy(fx["F128"])
    masks = {d: geometry_oracle.epipolar_mask(F, 128 // d, 128 // d, d) for d in (8, 16, 32, 64)}
    cam = dict(pluker_embedding_features=[f.cuda() for f in inp["feats"]],
               sample_locs_dict={d: m.cuda() for d, m in masks.items()}, add_type="add_to_main_branch")
    args = (inp["x"].cuda(), inp["t"].cuda())
    y = unet(*args, context=inp["ctx_rep"].cuda(), fs=inp["fs"].cuda(), camera_condition=cam)
    y_again = unet(*args, context=inp["ctx_rep"].cuda(), fs=inp["fs"].cuda(), camera_condition=cam)
    assert torch.equal(y, y_again), "forward is not bitwise reproducible"
    _check(y, fx["y_cam_rep"], "medium camera vs reference fixture", 2.5e-2, 5e-2)
    y_nc = unet(*args, context=inp["ctx_pf"].cuda(), fs=inp["fs"].cuda(), camera_condition=None)
    _check(y_nc, fx["y_nocam_pf"], "medium no camera vs reference fixture", 2.5e-2, 5e-2)
    ref = torch.from_numpy(fx["y_cam_rep"])
    cam_cpu = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks, add_type="add_to_main_branch")
    with uo.operand_rounding(torch.bfloat16):
        emu = uo.unet_forward(sd, MEDIUM_CFG, inp["x"], inp["t"], inp["ctx_rep"], inp["fs"], cam_cpu, origin_h=128)
    e_emu = ((emu - ref).norm() / ref.norm()).item()
    e_hip = ((y.cpu() - ref).norm() / ref.norm()).item()
    print(f"[parity] medium: bf16-emulated oracle rel_l2={e_emu:.3e}; HIP rel_l2={e_hip:.3e}")
    assert e_hip <= 1.6 * e_emu + 2e-3
    # native mask path (F -> packed bits on the GPU) gives the same eps
    from camc2v_amd import camera
    packed = camera.epipolar_masks_packed(F.cuda(), 16, 128, 128)
    y_p = unet(*args, context=inp["ctx_rep"].cuda(), fs=inp["fs"].cuda(),
               camera_condition=dict(cam, sample_locs_dict=None, sample_locs_packed=packed))
    _check(y_p, y, "native packed masks vs bool masks", 5e-3, 2e-2)


def test_module_level_api(small):
    """Reference-shaped module entry points (CrossAttention / FeedForward take [b, n, C] like the reference's)."""
    from oracle import unet_oracle
    unet, _, sd, _, _, _, _ = small
    blk = unet.input_blocks[1][1].transformer_blocks[0]
    p = "input_blocks.1.1.transformer_blocks.0"
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 50, 64, generator=g)
    ctx = torch.randn(3, 77 + 16, 1024, generator=g)
    _check(blk.attn1(x.cuda()), unet_oracle.cross_attention(sd, p + ".attn1", x, None, 1, False), "attn1 module")
    _check(blk.attn2(x.cuda(), context=ctx.cuda()),
           unet_oracle.cross_attention(sd, p + ".attn2", x, ctx, 1, True), "attn2 module")
    _check(blk.ff(x.cuda()), unet_oracle.feed_forward(sd, p + ".ff", x), "ff module")


def _oracle_masks(T, px, cond_index=0):
    """Epipolar masks of the synthetic benchmark trajectory from the geometry oracle (bool, per downsample key)."""
    from oracle import geometry_oracle as go
    K = torch.tensor([[px / 2, 0, px / 2], [0, px / 2, px / 2], [0, 0, 1.0]]).repeat(1, T, 1, 1)
    w2c = go.synthetic_trajectory(1, T)
    _, masks = go.camera_masks(K, w2c, torch.full((1,), cond_index, dtype=torch.long), px, px, resolutions=(8, 4, 2, 1))
    return masks


def _cfg_trajectory_vs_oracle(unet, sd, target, T, steps, scale, eta, tol, camera_cfg=1.0, camera_cfg_scheduler="constant",
                              use_graph=False):
    """BASELINE.json configs[3]/[4]-style check: a short CFG DDIM trajectory of the camera-conditioned model, both
    contexts per-frame (77 + 16 T tokens), through the reference-shaped wrapper `target`, against the fp32 oracle
    UNet + oracle sampler run here on the CPU with the same weights, inputs and noise draws."""
    from oracle import ddim_oracle, unet_oracle
    from oracle.golden_inputs import SMALL_CFG, small_inputs
    from utils.utils import instantiate_from_config
    dev = torch.device("cuda:0")
    inp = small_inputs(b=1, T=T, seed=77 + T)
    masks = _oracle_masks(T, 64)
    g = torch.Generator().manual_seed(T)
    ctx_uc = torch.randn(1, 77 + 16 * T, 1024, generator=g)
    noises = [torch.randn(1, 4, T, 8, 8, generator=g) for _ in range(steps)]
    cam_cpu = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks,
                   cond_frame_index=torch.zeros(1, dtype=torch.long), add_type="add_to_main_branch")

    def eps(ctx, camera=cam_cpu):
        return lambda x, t: unet_oracle.unet_forward(sd, SMALL_CFG, torch.cat([x, inp["c_concat"]], 1), t, ctx, inp["fs"],
                                                     camera, origin_h=64)

    with torch.no_grad():
        ref, _ = ddim_oracle.ddim_sample(eps(inp["ctx_pf"]), eps(ctx_uc), inp["x_T"], steps, eta, scale, 0.7, noises,
                                         apply_nocam=eps(inp["ctx_pf"], None), camera_cfg=camera_cfg,
                                         camera_cfg_scheduler=camera_cfg_scheduler)
    core = instantiate_from_config({"target": target, "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG},
        linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4, image_size=[8, 8],
        temporal_length=T)})
    core.model.diffusion_model = unet
    core = core.to(dev)
    to = lambda t: t.to(dev)
    cam = dict(pluker_embedding_features=[to(f) for f in inp["feats"]], sample_locs_dict={d: to(m) for d, m in masks.items()},
               cond_frame_index=torch.zeros(1, dtype=torch.long, device=dev), add_type="add_to_main_branch")
    cc = to(inp["c_concat"])
    cond = dict(c_concat=[cc], c_crossattn=[to(inp["ctx_pf"])], camera_condition=cam)
    uncond = dict(c_concat=[cc], c_crossattn=[to(ctx_uc)])
    samples, _ = core.sample_log(cond, 1, True, steps, eta=eta, x_T=inp["x_T"], unconditional_guidance_scale=scale,
                                 unconditional_conditioning=uncond, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                 fs=to(inp["fs"]), enable_camera_condition=True, injected_noise=noises, use_graph=use_graph,
                                 **({} if camera_cfg == 1.0 else dict(camera_cfg=camera_cfg, camera_cfg_scheduler=camera_cfg_scheduler)))
    _check(samples, ref.numpy(), f"{target} T={T} CFG {scale} camera_cfg {camera_cfg}: {steps}-step trajectory vs oracle", *tol)
    return samples


def test_cami2v_baseline_config_trajectory(small):
    """configs[3]: the CamI2V baseline target (epipolar path, ctx 77+16t on both CFG passes), same checkpoint layout."""
    unet, _, sd, _, _, _, _ = small
    _cfg_trajectory_vs_oracle(unet, sd, "baseline.cami2v.CamI2V", T=16, steps=2, scale=7.5, eta=1.0, tol=(1e-1, 2e-1))


def test_32_frame_clip_cfg_3p5(small):
    """configs[4] at UNet level: a 32-frame clip (temporal attention over 32 frames, epipolar attention over 32*h*w
    tokens, context rule 77 + 16*32) at CFG 3.5 against the oracle."""
    unet, _, sd, _, _, _, _ = small
    _cfg_trajectory_vs_oracle(unet, sd, "model.camcontexti2v.CamContextI2V", T=32, steps=2, scale=3.5, eta=0.0, tol=(1e-1, 2e-1))


def test_camera_guidance_trajectory(small):
    """camera_cfg != 1: every step runs a third UNet forward without the camera (lvdm/models/samplers/ddim.py:268-280);
    cosine weight; eager and hipGraph replay agree bit for bit and match the oracle sampler."""
    unet, _, sd, _, _, _, _ = small
    kw = dict(T=16, steps=2, scale=7.5, eta=1.0, tol=(1e-1, 2e-1), camera_cfg=2.0, camera_cfg_scheduler="cosine")
    eager = _cfg_trajectory_vs_oracle(unet, sd, "model.camcontexti2v.CamContextI2V", **kw)
    graphed = _cfg_trajectory_vs_oracle(unet, sd, "model.camcontexti2v.CamContextI2V", use_graph=True, **kw)
    assert torch.equal(eager, graphed)


def _small_core(unet):
    from camc2v_amd.diffusion import LatentDiffusionCore
    from oracle.golden_inputs import SMALL_CFG
    core = LatentDiffusionCore({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG},
                               linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4,
                               image_size=[8, 8], temporal_length=16)
    core.model.diffusion_model = unet
    return core.to("cuda:0")


def test_one_graph_serves_new_clips_and_notices_new_weights(small):
    """The captured step graph lives on static copies of the conditioning: a clip with NEW tensors of the same shapes
    replays the same graph (its step-invariant inputs are recomputed by the prologue graph), bit-identical to eager;
    after load_state_dict the graph is captured again instead of replaying launches that read the old packed weights."""
    unet, fx, sd, _, g, cam, _ = small
    core = _small_core(unet)
    core.__dict__.pop("_ccv_graph_cache", None)
    gen = torch.Generator(device="cuda").manual_seed(9)
    noises = [torch.randn(2, 4, 16, 8, 8, device="cuda", generator=gen) for _ in range(3)]

    def clip(seed, use_graph):
        gg = torch.Generator(device="cuda").manual_seed(seed)
        rn = lambda like: torch.randn(like.shape, device="cuda", generator=gg)
        cam2 = dict(cam, pluker_embedding_features=[rn(f) * 0.1 for f in cam["pluker_embedding_features"]])
        cc = rn(g["c_concat"])
        cond = dict(c_concat=[cc], c_crossattn=[rn(g["ctx_rep"])], camera_condition=cam2)
        uncond = dict(c_concat=[cc], c_crossattn=[rn(g["ctx_pf"])])
        s, _ = core.sample_log(cond, 2, True, 3, eta=1.0, x_T=rn(g["x"][:, :4]), unconditional_guidance_scale=7.5,
                               unconditional_conditioning=uncond, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                               fs=g["fs"], enable_camera_condition=True, injected_noise=noises, use_graph=use_graph)
        assert uncond["camera_condition"]["is_uc"] is True       # what the eager path (and the reference) leaves behind
        return s.clone()

    a_graph, b_graph = clip(1, True), clip(2, True)
    assert len(core.__dict__["_ccv_graph_cache"]) == 1            # one capture served both clips
    first = next(iter(core.__dict__["_ccv_graph_cache"].values()))
    a_eager, b_eager = clip(1, False), clip(2, False)
    assert torch.equal(a_graph, a_eager) and torch.equal(b_graph, b_eager)
    assert not torch.equal(a_graph, b_graph)
    # new weights: the old capture must not be replayed
    sd2 = {k: (v * 1.01 if v.dim() > 1 else v) for k, v in sd.items()}
    unet.load_state_dict(sd2, strict=True)
    try:
        c_graph = clip(1, True)
        assert next(iter(core.__dict__["_ccv_graph_cache"].values())) is not first
        assert torch.equal(c_graph, clip(1, False))
        assert not torch.equal(c_graph, a_graph)
    finally:
        unet.load_state_dict(sd, strict=True)
        core.__dict__.pop("_ccv_graph_cache", None)


def test_two_clips_in_flight_on_two_streams_equal_one_at_a_time(small):
    """bench.py --lanes: two host threads, each on its own HIP stream, sample different clips concurrently through the SAME
    model object.  Every stream gets its own hipGraphs and static conditioning buffers (graph cache keyed by stream), the
    sparse attention's work-queue counters are caller-owned (CcvAttn.queue_counters), so the overlapping launches share no
    state: every clip must come out bit-identical to sampling it alone."""
    import threading
    unet, fx, sd, _, g, cam, _ = small
    core = _small_core(unet)
    core.__dict__.pop("_ccv_graph_cache", None)
    gen = torch.Generator(device="cuda").manual_seed(9)
    noises = [torch.randn(2, 4, 16, 8, 8, device="cuda", generator=gen) for _ in range(3)]

    def inputs(seed):
        gg = torch.Generator(device="cuda").manual_seed(seed)
        rn = lambda like: torch.randn(like.shape, device="cuda", generator=gg)
        cam2 = dict(cam, pluker_embedding_features=[rn(f) * 0.1 for f in cam["pluker_embedding_features"]])
        cc = rn(g["c_concat"])
        return (dict(c_concat=[cc], c_crossattn=[rn(g["ctx_rep"])], camera_condition=cam2), dict(c_concat=[cc], c_crossattn=[rn(g["ctx_pf"])]),
                rn(g["x"][:, :4]))

    def clip(inp, use_graph=True):
        cond, uncond, x_T = inp
        s, _ = core.sample_log(cond, 2, True, 3, eta=1.0, x_T=x_T, unconditional_guidance_scale=7.5, unconditional_conditioning=dict(uncond),
                               timestep_spacing="uniform_trailing", guidance_rescale=0.7, fs=g["fs"], enable_camera_condition=True,
                               injected_noise=noises, use_graph=use_graph)
        return s.clone()

    sets = [inputs(100 + i) for i in range(6)]
    torch.cuda.synchronize()
    alone = [clip(x) for x in sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got, errors = {}, []

    def lane(l):
        try:
            torch.cuda.set_device(0)
            with torch.no_grad(), torch.cuda.stream(streams[l]):
                for rep in range(2):                 # the first pass captures this lane's graphs, the second replays them
                    for i in range(l, len(sets), 2):
                        got[(rep, i)] = clip(sets[i])
                streams[l].synchronize()
        except BaseException as e:
            errors.append(e)

    try:
        threads = [threading.Thread(target=lane, args=(l,)) for l in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        lanes_cached = {k[4] for k in core.__dict__["_ccv_graph_cache"]}
        assert len(lanes_cached) == 3                # the default stream's set + one per lane
        for (rep, i), s in got.items():
            assert torch.equal(s, alone[i]), f"clip {i} (pass {rep}) differs when another clip is in flight"
    finally:
        core.__dict__.pop("_ccv_graph_cache", None)


def test_cfg_pair_falls_back_when_halves_do_not_share_the_camera(small):
    """A conditional dict with a camera but enable_camera_condition off: the reference's second apply_model runs WITHOUT
    the camera (ddim.py:258-263).  The batched 2b forward cannot express that, so apply_model_pair must take two forwards."""
    unet, _, _, _, g, cam, _ = small
    core = _small_core(unet)
    cond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_rep"]], camera_condition=cam)
    uncond = dict(c_concat=[g["c_concat"]], c_crossattn=[g["ctx_pf"]])
    x = g["x"][:, :4].contiguous()
    e_c, e_uc = core.apply_model_pair(x, g["t"], cond, uncond, fs=g["fs"])
    want_uc = core.apply_model(x, g["t"], uncond, fs=g["fs"])                     # no camera
    with_cam = core.apply_model(x, g["t"], dict(uncond, camera_condition=cam), fs=g["fs"])
    assert torch.equal(e_uc, want_uc) and not torch.equal(e_uc, with_cam)
    assert torch.equal(e_c, core.apply_model(x, g["t"], cond, fs=g["fs"]))
    # the sampler's shared copy (same tensors + the is_uc marker) does take the batched path
    shared = dict(uncond, camera_condition=dict(cam, is_uc=True))
    assert core._same_extras(cond, shared) and not core._same_extras(cond, uncond)


def test_fused_camera_projections_equal_three_gemms(small, monkeypatch):
    """x += attn1.to_out(o1) + pluker_projection(n + P) + epipolar.to_out(o2) as ONE stacked-operand GEMM (K = 3C) against
    the three separate stream updates: same numbers up to the fp32 summation order."""
    from camc2v_amd import unet as unet_mod
    unet, fx, _, _, g, cam, _ = small
    assert unet_mod.FUSE_CAMERA_PROJECTIONS
    y_fused = unet(g["x"], g["t"], context=g["ctx_rep"], fs=g["fs"], camera_condition=cam)
    monkeypatch.setattr(unet_mod, "FUSE_CAMERA_PROJECTIONS", False)
    y_three = unet(g["x"], g["t"], context=g["ctx_rep"], fs=g["fs"], camera_condition=cam)
    _check(y_fused, fx["y_cam_rep"], "fused camera projections vs reference fixture")
    _check(y_three, fx["y_cam_rep"], "three separate projections vs reference fixture")
    _check(y_fused, y_three, "fused vs separate", 5e-2, 8e-2)      # this fixture amplifies any rounding difference (see header)


def test_config4_32_frames_cfg_3p5_medium(golden_dir):
    """BASELINE.json configs[4] -- a 32-frame clip at CFG 3.5 -- at the medium width (model_channels 128, 16x16 latents: epipolar
    attention over L = 32 * 256 = 8192 tokens through the workgroup-shared sparse kernel, context rule 77 + 16 * 32 on both CFG
    passes): a 2-step DDIM trajectory (guidance_rescale 0.7, eta 0) through `model.camcontexti2v.CamContextI2V` + DDIMSampler against
    the fp32 oracle UNet + oracle sampler on the host.  Stated tolerance: rel-L2 of the final latents <= 6e-2 (guidance amplifies
    the per-forward error by ~(1 + 2 * 3.5 * rescale)).  The e4m3 (fp8 MFMA) attention variant that configs[4] names was built in
    round 2, measured slower than the bf16 kernel in rounds 2-4 and retired in round 4 (profiles/r04_fp8_retired.txt)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ddim_oracle, unet_oracle
    from oracle.golden_inputs import MEDIUM_CFG, SEED, small_inputs
    from utils.utils import instantiate_from_config
    dev = torch.device("cuda:0")
    T, steps, scale = 32, 2, 3.5
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, SEED)
    inp = small_inputs(b=1, T=T, hl=16, seed=SEED + 40, chans=(128, 256, 512, 512))
    masks = _oracle_masks(T, 128)
    g = torch.Generator().manual_seed(4 * T)
    ctx_uc = torch.randn(1, 77 + 16 * T, 1024, generator=g)
    cam_cpu = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks,
                   cond_frame_index=torch.zeros(1, dtype=torch.long), add_type="add_to_main_branch")

    def eps(ctx):
        return lambda x, t: unet_oracle.unet_forward(sd, MEDIUM_CFG, torch.cat([x, inp["c_concat"]], 1), t, ctx, inp["fs"], cam_cpu, origin_h=128)

    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref, _ = ddim_oracle.ddim_sample(eps(inp["ctx_pf"]), eps(ctx_uc), inp["x_T"], steps, 0.0, scale, 0.7, None)
    core = instantiate_from_config({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(MEDIUM_CFG)},
        linear_start=0.00085, linear_end=0.012, conditioning_key="hybrid", channels=4, image_size=[16, 16], temporal_length=T,
        add_type="add_to_main_branch",
        pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": {}},
        epipolar_config=dict(origin_h=128, origin_w=128, is_3d_full_attn=False, num_register_tokens=4,
                             attention_resolution=[8, 4, 2, 1], compression_factor=1))})
    core.model.diffusion_model.load_state_dict(sd, strict=True)
    core = core.to(dev).eval()
    to = lambda t: t.to(dev)
    cam = dict(pluker_embedding_features=[to(f) for f in inp["feats"]], sample_locs_dict={d: to(m) for d, m in masks.items()},
               cond_frame_index=torch.zeros(1, dtype=torch.long, device=dev), add_type="add_to_main_branch")
    cc = to(inp["c_concat"])

    def sample():
        cond = dict(c_concat=[cc], c_crossattn=[to(inp["ctx_pf"])], camera_condition=cam)
        uncond = dict(c_concat=[cc], c_crossattn=[to(ctx_uc)])
        out, _ = core.sample_log(cond, 1, True, steps, eta=0.0, x_T=inp["x_T"], unconditional_guidance_scale=scale,
                                 unconditional_conditioning=uncond, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                 fs=to(inp["fs"]), enable_camera_condition=True)
        return out

    x16 = sample()
    _check(x16, ref.numpy(), "configs[4] medium, 32 frames, CFG 3.5: 2-step trajectory vs oracle", 6e-2, 1.5e-1)
